// nk2d_stream.h -- the forward year as a COMMAND STREAM (nk2d_stream.hip).
//
// The host-controlled year (nk2d_radau.hip) is a sequence of launches: attempt set-up, fused Newton iterations, error
// estimate, step boundary -- 24 000 of them per free-running 416 x 416 year, each a launch boundary at which 832 waves wait
// for the slowest of them, and a host round trip per Newton iteration.  Here the same sequence is a stream of COMMANDS -- a
// command is the argument block of the launch it replaces -- executed by ONE resident kernel: a workgroup owns a block of
// ypos columns with all their tracers for as long as the kernel runs, executes command after command with the device
// functions of the per-phase kernels (nk2d_bodies.h: the same bits), and between two commands waits for its two lateral
// neighbours only (a column reads what the columns to its left and right wrote in the command before, and nothing else of
// another workgroup).  The controller stays where it is, on the host, with its decisions, its speculation (the next Newton
// iteration or the error estimate queued before the verdict on this one) and its roll-back by pointer swap: it pushes
// commands into a ring in pinned host memory, a relay wave of the kernel copies them into HBM, and the per-column norm
// partials come back through pinned host memory with a completion stamp per workgroup -- no launch, no event, no
// stream synchronisation inside a year.  Whatever has no command (history samples, the rare several-sweep error
// estimates) ends the kernel, runs as ordinary launches and the next command starts the kernel again.
#pragma once

#include "nk2d_bodies.h"

enum { NK2D_OP_EXIT = 1, NK2D_OP_SETUP = 2, NK2D_OP_NEWTON = 3, NK2D_OP_ERR = 4, NK2D_OP_BOUNDARY = 5,
       // the several-sweep error estimate and the second estimate of a rejected step, launch for launch
       NK2D_OP_SWEEP = 6, NK2D_OP_ERR_RHS = 7, NK2D_OP_ERR_RHS2 = 8, NK2D_OP_ERR_NORM = 9, NK2D_OP_COPY = 10,
       // the launch that ends the last Newton iteration of a frozen step and the step (nk2d_r_newton_final)
       NK2D_OP_NEWTON_FINAL = 11,
       // Jacobian planes from a mixing plane in memory (nk2d_k_jac: modules whose Jacobian reads the state)
       NK2D_OP_JAC = 12 };
#define NK2D_CMD_NOTIFY 1   /* the host waits for this command: completion stamp of every workgroup to pinned memory */
#define NK2D_CMD_FACTOR 2   /* OP_NEWTON: the launch that computes the line factorisation of its column (first after an "LU" event) */

struct StreamSetup {        // nk2d_r_attempt_setup
    VmixArgs V;
    PredictArgs A;
    JacOut J;
};
struct StreamBoundary {     // nk2d_r_step_boundary
    VmixArgs V;
    BoundaryArgs B;
    PredictArgs A;
};
struct StreamColumns {      // nk2d_r_err_rhs / _err_rhs2 / _err_norm, a column-wise copy: operands by position
    const double *a, *b, *c, *d;
    double *out, *part;
    size_t nv;
    double h;
};
struct StreamJac {          // nk2d_k_jac
    const double* kvp;
    double *JL, *JU, *JS, *JN, *JC;
    const double* ylin;
    double* UPR;
};
struct StreamFinal {        // nk2d_r_newton_final
    FusedArgs nf;
    FinalArgs fin;
    VmixArgs V;
    JacOut J;
};
struct StreamCmd {
    int op, flags;
    union {
        StreamFinal fn;
        StreamJac jac;
        SweepArgs sw;       // nk2d_k_sweep
        StreamColumns col;
        FusedArgs nf;       // nk2d_r_newton_fused
        ErrArgs err;        // one launch of nk2d_r_err_fused
        StreamSetup su;
        StreamBoundary bd;
    } u;
};

// A ring slot is NK2D_CMD_DWORDS pairs (stamp << 32 | payload dword), each written and read with ONE 8-byte access: a
// reader that finds the expected stamp in every pair has the whole command, whatever order the pairs arrived in -- no fence,
// no flag to order against, on the PCIe hop (host -> relay wave) as on the device (relay wave -> workgroups).
#define NK2D_CMD_DWORDS 192
#define NK2D_RING_SLOTS 256
static_assert(sizeof(StreamCmd) <= 4 * NK2D_CMD_DWORDS, "a command must fit a ring slot");

struct StreamArgs {
    const unsigned long long* h_ring;   // pinned host memory: what the host pushes
    unsigned long long* d_ring;         // HBM: what the relay wave has forwarded
    unsigned* flags;                    // [nwg][32]: commands completed, one 128-byte line per workgroup
    int* abort_flag;
    unsigned* h_done;                   // pinned: [nwg] completion stamps of the commands flagged NOTIFY
    unsigned* h_status;                 // pinned: [0] != 0 once a workgroup or the relay has given up
    double* out;                        // [8]: status, commands completed by workgroup 0
    unsigned seq0;                      // stamp of the first command of this launch
    int nwg, cpw;                       // workgroups, ypos columns per workgroup
    long long spin_ticks;               // longest wait (ticks of s_memrealtime, 100 MHz)
    int fences;
    int coef_lds;                       // bit 0: the static coefficients of the workgroup's ypos columns live in LDS (dynamic shared
                                        // memory), bit 1: so does W of its columns
    double* W;                          // the context's W (3 nv): loaded into LDS when the kernel starts, stored back when it ends
    unsigned long long* prof;           // [nwg][12]: ticks waiting for a command, executing, waiting for neighbours; commands; per op
};

// host side (nk2d_stream.hip; what the integrator itself calls is declared in nk2d_common.h)
int nk2d_stream_push(nk2d_ctx* c, StreamCmd& cmd, bool notify, unsigned* seq_out = nullptr);
int nk2d_stream_wait(nk2d_ctx* c, unsigned seq);
unsigned nk2d_stream_last_seq(const nk2d_ctx* c);
