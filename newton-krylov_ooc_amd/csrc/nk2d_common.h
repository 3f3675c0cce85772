// nk2d_common.h -- context, device layout and wave-level primitives shared by the
// translation units of libnk2d.so (gfx950 only).
//
// Device layout ("lane-blocked columns").  A state vector of one tracer module is
// [tc][ny] columns; a column holds the nz depth levels of one (tracer, ypos) pair and
// is owned by ONE 64-lane wavefront.  With E = ceil(nz/64) levels per lane, level
// k = lane*E + e is stored at
//
//        ((tr*ny + j)*E + e)*64 + lane
//
// so that (a) every load/store of a column is a fully coalesced 512-byte row per e,
// (b) the vertical neighbours k-1 / k+1 are in the lane's own registers or one DPP
// shuffle away, which is what the vertical-mixing stencil and the per-column
// tridiagonal solves need, and (c) the horizontal neighbours (j-1, j+1) are the same
// (e, lane) slot of the adjacent column.  Levels k >= nz are padding and hold 0.
// Host arrays (C order (tracer, depth, ypos)) are converted at upload / download.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/nk2d.h"

#define NK2D_WAVE 64
#define NK2D_WAVES_PER_BLOCK 4
#define NK2D_BLOCK (NK2D_WAVE * NK2D_WAVES_PER_BLOCK)
#define NK2D_MAX_E 8
#define NK2D_OWN_REC_CAP 65536   /* rows of the context's own schedule record */
#define NK2D_CKPT_EVERY 128      /* steps between two checkpoints of a frozen year */


struct nk2d_ctx {
    nk2d_desc d;
    int nz, ny, tc, E, nzp, ncol, nreg;
    int kind;   // module kind: 0 linear sources (iage, forced), 1 phosphorus, 2 forced with forcing files
    size_t nv;  // doubles per state vector  (tc*ny*nzp)
    size_t np;  // doubles per (depth, ypos) plane (ny*nzp)
    int dev;
    hipStream_t stream_;   // the context's stream: through nk2d_s(c), which first ends a running command-stream kernel
    std::string err;
    // the year as a command stream (nk2d_stream.hip): ONE resident kernel executes the launches of the host-controlled year
    // as commands, workgroups handing over to their lateral neighbours instead of meeting at launch boundaries
    struct nk2d_stream_state* strm;
    int stream_years;      // option "stream_years": 1 = forward years of eligible contexts run as command streams
    int stream_two_waves;  // option "stream_two_waves": the 256-register flavour of the kernel where it lets every column be resident
    int stream_on;         // set by the integrator for the span of a year that may run as a command stream
    int stream_lost;       // years in a row whose kernel gave up (two: the context stops trying)
    int64_t stream_cmds, stream_launches, stream_timeouts, stream_years_run;   // counters (nk2d_get_counter)

    // static, packed planes (device)
    double* VV;      // vvel at ypos faces, (ny+1) columns
    double* KH;      // horizontal mixing coeff at ypos faces, (ny+1) columns, walls 0
    double* WT;      // wvel at the top face of cell k,    ny columns
    double* WB;      // wvel at the bottom face of cell k, ny columns
    double* DZR;     // depth.delta_r           [nzp]
    double* ZM0;     // depth.mid[k]            [nzp]
    double* ZM1;     // depth.mid[k+1]          [nzp]
    double* DM;      // depth.delta_mid[k]      [nzp]
    double* DMR;     // depth.delta_mid_r[k]    [nzp]
    double* DYR;     // ypos.delta_r            [ny]
    double* BLDMAX;  // bldepth_max             [ny]
    // region data
    int32_t* MASK;   // [np]
    double* WN;      // grid_weight / sum over region [np]
    // Jacobian planes at t_jac (tracer independent part), np each
    double *JL, *JU, *JS, *JN, *JC;
    // phosphorus module: light limitation [np], d uptake / d po4 at the linearisation state [np],
    // linearisation state of the stand-alone Jacobian entry points [nv]
    double *LIGHT, *UPR, *YLIN;
    // kind 2: forcing records on the device (packed planes / rows), their times on the host, and the length
    // of a KV buffer: the vertical mixing plane, then (kind 2) the source plane and the restoring targets of
    // the same time -- whatever is a function of time alone travels in one bundle
    double *SMSREC, *RESTREC;
    double *sms_t, *rest_t;
    size_t kv_len;
    int ylin_set;
    // vertical mixing planes: 3 stage times, current t, scratch
    double* KV[5];
    // second sets: the launch that ends a frozen step writes the next attempt's stage planes and Jacobian planes while
    // its own stage and sweep parts still read the current ones (nk2d_r_newton_final)
    double* KVN[3];
    double* JB[5];
    // Radau work vectors (nv each unless noted)
    double *Y, *YOLD, *F, *Z /*3nv*/, *ZP /*3nv*/, *W /*3nv*/;
    // third stage buffer: a simplified-Newton iteration whose solve needs ONE sweep runs as a single launch
    // (stage + sweep + update); its update writes the new stage values here while the neighbouring columns still read
    // the old ones from Z in the same launch, then Z and ZN swap
    double* ZN /*3nv*/;
    int min_sweeps;    // least sweeps per solve where columns couple (option "min_sweeps": 1 default, 2 = round-1 rule)
    std::vector<double> last_sched;     // accepted steps of the most recent free-running year (nk2d_last_schedule)
    std::vector<double> own_rec;        // its record buffer when the caller gave none
    std::vector<double> frozen_sched;   // accepted steps the perturbed years of nk2d_jvp repeat (nk2d_set_frozen_schedule)
    int jac_stage_state;   // 1: option "jac_stage" also for modules whose Jacobian reads the state (plane of the stage time, state of the step start)
    int final_fuse;    // 1: a frozen step ends in the launch of its last Newton iteration (option "final_fuse", for A/B runs)
    int jac_stage;     // >= 0: Jacobian of a step attempt from the vertical mixing plane of this stage time (option "jac_stage"); -1: step start
    int team;          // 1: Newton-iteration launches run as k_newton_team (one workgroup per column); 0: k_newton_fused (option "team")
    int single_swap;   // 1: single-launch iterations write ZN and swap (host-side decisions; see nk2d_radau.hip set_lu)
    int swap_updates;  // 1: ... and so do the update launches of several-sweep iterations (vector norm hook)
    double *BR, *BCR, *BCI, *XR[2], *XCR[2], *XCI[2], *TMP, *TMP2;
    // cached line factorisation of the current (h_lu, t_jac): pivot reciprocals and PCR tables
    double *FR_INV, *FC_INVR, *FC_INVI;   // nv each
    double *FR_TAB, *FC_TABR, *FC_TABI;   // ncol * NK2D_TAB * 64 each
    // single precision copies for the fused Newton launches: the line factorisation is an approximate
    // inverse inside an iteration that re-evaluates the exact residual, its storage precision only
    // touches the contraction rate (nk2d_set_option "factor_fp32")
    float *FR32_INV, *FC32_INVR, *FC32_INVI, *FR32_TAB, *FC32_TABR, *FC32_TABI;
    int factor_fp32;
    hipEvent_t snap_ev[8];  // "the launch that carries these partials is queued" (host-side decisions by launches)
    int snap_ready;
    int sweep_wpb;          // waves per block of the sweep kernel (1, 2 or 4)
    int jac_fresh;          // 1: re-evaluate the Jacobian at every step start (see nk2d_set_option)
    double growth_cap;      // > 0: largest step growth factor after a step whose Newton iteration failed at first
                            // (nk2d_set_option "growth_cap"; 0 = SciPy: no memory of the failure)
    // reductions
    double* PART;    // per-task partials
    double* PART2;   // second buffer [ncol]
    double* STEP_PART;      // frozen year: norm partials of the last two Newton iterations of every step, rows of ncol
    size_t step_part_rows;  // rows allocated
    double* STEP_NORM;  // [3 * NK2D_OWN_REC_CAP] per step of a frozen year: sum((dW/scale)^2) of its last and last-but-one iteration, sum((err/scale)^2)
    int64_t frozen_fallbacks;   // frozen years rejected by the a-posteriori Newton check (nk2d_frozen_fallbacks)
    // the frozen year of a small grid in one launch on a schedule cache (k_frozen_persistent, nk2d_kernels.hip)
    void* frozen_cache;
    int frozen_persistent;          // option "frozen_persistent": 1 = where eligible (default), 0 = never
    int frozen_persistent_max_e;    // ... for grids of at most this many levels per lane (option "frozen_persistent_max_e")
    double frozen_cache_max_gb;     // ... whose schedule cache stays below this size (option "frozen_cache_gb")
    int64_t frozen_cache_builds, frozen_persistent_years;
    int frozen_team;      // option "frozen_team": a four-wave team per column inside the one-launch frozen year (grids of at most two levels per lane)
    int frozen_wpb;       // option "frozen_wpb": columns (waves) per workgroup of the wave-per-column one-launch year with neighbour hand-over
    int frozen_coef_lds;  // option "frozen_coef_lds": the one-launch year keeps the static coefficients of a wave's column in LDS
    int frozen_by_column; // option "frozen_by_column": a workgroup of the one-launch year is one ypos column with all its tracers
    int frozen_alloc_async;   // option "frozen_alloc_async": a schedule cache above 8 GB is allocated by a thread of its own
    int frozen_cache_after;   // option "frozen_cache_after": frozen years of a schedule that run launch by launch before its cache is built (default 0; -1: 0 for a cache below 8 GB, 3 above)
    uint64_t frozen_seen_key; int frozen_seen_years;   // the schedule last seen by nk2d_frozen_persistent and its years so far
    int64_t frozen_team_years;
    int64_t frozen_launch_us;   // device time of the one-launch frozen years so far (HIP events around the launch)
    double barrier_timeout_ms;  // longest wait at a grid barrier of the persistent year (option "barrier_timeout_ms")
    int year_fences;            // 1: release / acquire fences around its grid barriers (option "year_fences", validation)
    int64_t frozen_resumes;     // ... and resumed from a checkpoint with one more Newton iteration (nk2d_frozen_resumes)
    int frozen_err_check;       // k > 0: SciPy's error estimate on every k-th step of a frozen year (option "frozen_err_check")
    uint64_t grid_hash;         // hash of the grid / module description given to nk2d_create (nk2d_fingerprint)
    // state kept every NK2D_CKPT_EVERY steps of a frozen year (Y, YOLD, ZP: 5 nv doubles each) for the resume
    std::vector<double*> ckpt;
    double* RED;     // reduced scalars (device)
    double* hRED;    // pinned host mirror
    double* hPART;   // pinned, device-visible per-column partials [ncol] (host-controlled integrator)
    double* hPART2;  // same, for an error estimate queued behind a Newton iteration not yet judged
    double* hPARTB;  // same, second buffer for Newton iterations queued one ahead of the one being judged
    double* hPARTC;  // ... and two ahead (vector norm hook)
    double* ZS /*3nv*/;   // second spare set of stage values, allocated with a vector norm hook: iterations queued TWO ahead
    int hook_spec_depth;  // option "hook_spec_depth": whole iterations a hooked controller queues ahead of a verdict (1 or 2)
    // what the host-side controller queued ahead of a verdict and had to drop (nk2d_get_counter)
    int64_t cnt_spec_dropped, cnt_front_dropped, cnt_err_void, cnt_err_queued;
    double* part_cur;  // where the next fused launch with the update puts its partials (null: hPART / PART)
    int part_on_host;
    int factor_pending;          // set by the integrator's "LU" event, consumed by the next fused launch
    double lu_cre, lu_ccr, lu_cci;  // shifts of the current line factorisation
    int speculate;   // 1: queue the next Newton iteration's front launches before reading the norm
    double spec_bias;   // option "spec_bias": the iteration in flight is EXPECTED to pass SciPy's convergence test when its predicted test value is below this many tolerances (what is queued behind it: the error estimate or another iteration; never a decision)
    // freed state vectors kept for reuse (nk2d_vec_alloc / nk2d_vec_free)
    std::vector<double*> vec_pool;
    std::mutex pool_mutex;
    // staging for host <-> device layout conversion
    double* STAGE;
    double* hSTAGE;  // pinned host twin of STAGE
    size_t stage_elems;
    // downloads in two halves (nk2d_vec_download_begin / _end): staging pairs of their own, kept for reuse
    struct nk2d_download_pool* dl_pool;
    // region scalars staged on device for the algebra kernels
    double* RCOEF;
    double* hRCOEF;  // pinned host twin of RCOEF
    size_t rcoef_elems;

    // line-relaxation contraction bound rho(c) = max_i s_i / (c + q_i), tabulated at create
    // on a log grid of shifts c (rho is decreasing in c: the entry at the grid point below c
    // is a valid, at most ~4 % pessimistic bound)
    std::vector<double> rho_tab;
    double rho_c0, rho_dlog;   // first grid shift, log10 spacing
    // preconditioner (banded LU), see nk2d_precond.hip
    void* precond;
    int pc_fused;  // 1 (default): a panel step of the Gauss-Jordan inversions is ONE launch (k_pc_gj_step) for blocks of 1024 rows and more; 2: at every size; 0: two launches (k_pc_gj_rows + k_pc_gj_update_mfma) -- the same bits
    int pc_valu;   // 1: the round-1 preconditioner kernels (VALU rank-32 update, 8-byte mat-vec loads), for A/B runs
    int pc_fp32;   // 1: Schur inverses of the linear modules' preconditioner stored in single precision (option "pc_fp32")
    int pc_refine; // ... with this many refinement steps per apply against the exact operator (option "pc_refine", default 1)

    // optional dense-output sampling of the running comp_fcn (history files)
    int hist_n, hist_next;
    const double* hist_t;
    double* hist_host;   // [hist_n][tc][nz][ny]

    // counters of the running comp_fcn
    nk2d_stats st;

    // coupling of the integrator's scalar norms across contexts that share one tracer module (tracers
    // sharded over GPUs, nk2d_set_norm_hook): every sum of squares the controller reads goes through
    // norm_hook (an all-reduce supplied by the caller) and n_total is the module's size, not the shard's
    nk2d_norm_hook_fn norm_hook;
    void* norm_hook_user;
    nk2d_norm_hook_vec_fn norm_hook_vec;   // several sums per call (nk2d_set_norm_hook_vec); norm_hook then wraps it
    void* norm_hook_vec_user;
    double global_n;
    // persistent whole-year kernel (device_ctl 3, nk2d_kernels.hip): norm partials [2][ncol], result block,
    // barrier counter + abort flag, sweeps-per-shift table, schedule record, timing events
    double *YR_PART, *YR_OUT, *hYR_OUT, *YR_REC;
    void* YR_SYNC;
    int* YR_MTAB;
    double yr_lin_tol;
    int64_t yr_rec_cap;
    hipEvent_t yr_ev[2];
    // generic event pair on the context's stream (nk2d_timer_begin / nk2d_timer_end)
    hipEvent_t timer_ev[2];
    int timer_ready;

    // sampled HIP-event timing of the line-relaxation sweep kernel (nk2d_profile_*)
    int prof_every;
    std::vector<hipEvent_t> prof_ev;  // start/stop pairs
    size_t prof_used;                 // events in use (2 per sample)
    double prof_ms_sum;
    double prof_overhead_ms;          // elapsed time of an EMPTY event pair (calibration)
    int64_t prof_cnt;
    int64_t sweep_launches;           // launches of the dominant kernel since the last nk2d_profile_reset
    double fused_bytes_all;           // algorithmic bytes of ALL those launches (timed or not)
    int64_t shape_cnt[4];             // non-factorising launches by shape: stage+update, stage only, update only, neither
    double shape_bytes[4];            // their algorithmic bytes
    double sweep_bytes;               // algorithmic bytes of the launches inside timed windows
    int64_t prof_windows;             // timed windows folded into prof_ms_sum (prof_cnt: their launches)
    std::vector<int> prof_win_launches;
    int win_open, win_launches;
    int64_t win_seq;
    double win_bytes;
};

#define NK2D_CHECK(ctx, call)                                                        \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_) + " at " + \
                         __FILE__ + ":" + std::to_string(__LINE__);                  \
            return -1;                                                               \
        }                                                                            \
    } while (0)

#define NK2D_TRY(expr)            \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != 0) return rc_; \
    } while (0)

// nk2d_stream.hip (the year as a command stream, nk2d_stream.h)
void nk2d_turn_take(int waves);   // resident kernels of a process: one at a time, or side by side where their waves leave room
void nk2d_turn_give(int waves);
#define NK2D_RC_STREAM_LOST 17   /* the command-stream kernel gave up (a wait timed out): the caller reruns the year by launches */
int nk2d_stream_pause(nk2d_ctx* c);
bool nk2d_stream_running(const nk2d_ctx* c);
int nk2d_stream_columns_per_workgroup(const nk2d_ctx* c);
int nk2d_stream_two_waves(const nk2d_ctx* c);
int nk2d_stream_eligible(const nk2d_ctx* c);
int nk2d_stream_ready(nk2d_ctx* c);     // buffers of the command stream in place (first use)
int nk2d_stream_end(nk2d_ctx* c);       // ends the kernel, waits for it; NK2D_RC_STREAM_LOST if it had given up on the way
void nk2d_stream_free(nk2d_ctx* c);
int nk2d_stream_profile(nk2d_ctx* c, double* out12);
double* nk2d_stream_part_take(nk2d_ctx* c, const double* name);
const double* nk2d_stream_part_named(const nk2d_ctx* c, const double* name);
void nk2d_stream_part_forget(nk2d_ctx* c, const double* name);
int nk2d_stream_wait_part(nk2d_ctx* c, const double* part, int n);
// the stream every launch, copy and synchronisation of a context goes to: whatever is queued there must come AFTER the
// commands pushed so far, so a resident command-stream kernel is told to finish first (the caller's launch is then ordered
// behind it by the stream; nothing is waited for here)
static inline hipStream_t nk2d_s(nk2d_ctx* c) {
    if (c->strm && nk2d_stream_running(c)) (void)nk2d_stream_pause(c);
    return c->stream_;
}

static inline int nk2d_fail(nk2d_ctx* c, const std::string& msg, int code = -2) {
    c->err = msg;
    return code;
}

// Launch shape of the Newton-iteration launches by default: 0 one wave per column (k_newton_fused), 1 a team of four waves
// (k_newton_team), 2 a pair of waves (k_newton_pair).  All three give bit-identical results.  A team or pair wave holds
// at most 256 VGPRs, so the chip's 1024 SIMDs take 2048 of them at once; beyond that the workgroups run in rounds and
// the split loses (iage 416^2, 832 columns: 22.5 us per launch with teams, frozen year 0.225 s with pairs, against
// 16.4 us / 0.198 s with one wave per column).  Where everything is resident it wins, the more the smaller the grid --
// frozen year of iage, one wave / team / pair per column: 26^2 18.6 / 15.9 / 16.0 ms, 52^2 30.0 / 25.7 / 26.0 ms,
// 104^2 54.1 / 48.1 / 46.4 ms, 208^2 97.9 / 98.3 / 93.7 ms; one-tracer module at 416^2 (416 columns of 7 levels per
// lane): year 0.66 -> 0.60 s with teams (profiles/r02_team*.log).
static inline int nk2d_team_auto(const nk2d_ctx* c) {
    if (c->ncol <= 128) return 1;
    if (c->ncol <= 512) return (c->E >= 5) ? 1 : 2;
    return 0;
}

static inline int nk2d_grid(int ntasks) { return (ntasks + NK2D_WAVES_PER_BLOCK - 1) / NK2D_WAVES_PER_BLOCK; }

// run `stmt` with a compile-time constant EE equal to the runtime levels-per-lane
#define NK2D_DISPATCH_E(Eval, ...)                                 \
    switch (Eval) {                                                  \
        case 1: { constexpr int EE = 1; __VA_ARGS__; } break;               \
        case 2: { constexpr int EE = 2; __VA_ARGS__; } break;               \
        case 3: { constexpr int EE = 3; __VA_ARGS__; } break;               \
        case 4: { constexpr int EE = 4; __VA_ARGS__; } break;               \
        case 5: { constexpr int EE = 5; __VA_ARGS__; } break;               \
        case 6: { constexpr int EE = 6; __VA_ARGS__; } break;               \
        case 7: { constexpr int EE = 7; __VA_ARGS__; } break;               \
        case 8: { constexpr int EE = 8; __VA_ARGS__; } break;               \
        default: break;                                              \
    }

// the same for the kernels that exist for small grids only (at most 4 levels per lane = 256 depth levels)
#define NK2D_DISPATCH_E4(Eval, ...)                                \
    switch (Eval) {                                                  \
        case 1: { constexpr int EE = 1; __VA_ARGS__; } break;               \
        case 2: { constexpr int EE = 2; __VA_ARGS__; } break;               \
        case 3: { constexpr int EE = 3; __VA_ARGS__; } break;               \
        case 4: { constexpr int EE = 4; __VA_ARGS__; } break;               \
        default: break;                                              \
    }

// as above, plus a compile-time module kind KK (0: linear sources, 1: phosphorus, 2: forcing files)
#define NK2D_DISPATCH_EK(Eval, kind, ...)                                        \
    if ((kind) == 1) {                                                            \
        constexpr int KK = 1;                                                     \
        NK2D_DISPATCH_E(Eval, __VA_ARGS__)                                        \
    } else if ((kind) == 2) {                                                     \
        constexpr int KK = 2;                                                     \
        NK2D_DISPATCH_E(Eval, __VA_ARGS__)                                        \
    } else {                                                                      \
        constexpr int KK = 0;                                                     \
        NK2D_DISPATCH_E(Eval, __VA_ARGS__)                                        \
    }

// ---------------------------------------------------------------------------------
// device primitives
// ---------------------------------------------------------------------------------
#ifdef __HIPCC__

struct cplx {
    double re, im;
};
__device__ __forceinline__ cplx c_make(double r, double i) { cplx z; z.re = r; z.im = i; return z; }

// arithmetic helpers overloaded for double / cplx so the solver template serves both
__device__ __forceinline__ double t_from_real(double a, double) { return a; }
__device__ __forceinline__ cplx t_from_real(double a, cplx) { return c_make(a, 0.0); }
__device__ __forceinline__ double t_zero(double) { return 0.0; }
__device__ __forceinline__ cplx t_zero(cplx) { return c_make(0.0, 0.0); }
__device__ __forceinline__ double t_one(double) { return 1.0; }
__device__ __forceinline__ cplx t_one(cplx) { return c_make(1.0, 0.0); }

__device__ __forceinline__ double t_neg(double a) { return -a; }
__device__ __forceinline__ cplx t_neg(cplx a) { return c_make(-a.re, -a.im); }
__device__ __forceinline__ double t_mul(double a, double b) { return a * b; }
__device__ __forceinline__ cplx t_mul(cplx a, cplx b) {
    return c_make(__builtin_fma(a.re, b.re, -(a.im * b.im)), __builtin_fma(a.re, b.im, a.im * b.re));
}
__device__ __forceinline__ double t_mulr(double a, double s) { return a * s; }
__device__ __forceinline__ cplx t_mulr(cplx a, double s) { return c_make(a.re * s, a.im * s); }
// a - b*c
__device__ __forceinline__ double t_nfma(double a, double b, double c) { return __builtin_fma(-b, c, a); }
__device__ __forceinline__ cplx t_nfma(cplx a, cplx b, cplx c) {
    double re = __builtin_fma(-b.re, c.re, a.re);
    re = __builtin_fma(b.im, c.im, re);
    double im = __builtin_fma(-b.re, c.im, a.im);
    im = __builtin_fma(-b.im, c.re, im);
    return c_make(re, im);
}
// a - b*s with s real
__device__ __forceinline__ double t_nfmar(double a, double b, double s) { return __builtin_fma(-b, s, a); }
__device__ __forceinline__ cplx t_nfmar(cplx a, cplx b, double s) {
    return c_make(__builtin_fma(-b.re, s, a.re), __builtin_fma(-b.im, s, a.im));
}
// reciprocal for the pivots of diagonally dominant systems (finite, far from 0 and inf):
// hardware estimate + two Newton steps, ~1 ulp, a third of the instructions of an IEEE divide
__device__ __forceinline__ double fast_rcp(double a) {
#ifdef NK2D_IEEE_DIV
    return 1.0 / a;
#else
    double r = __builtin_amdgcn_rcp(a);
    r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
    return r;
#endif
}
__device__ __forceinline__ double t_recip(double a) { return fast_rcp(a); }
__device__ __forceinline__ cplx t_recip(cplx a) {
    double n = fast_rcp(__builtin_fma(a.re, a.re, a.im * a.im));
    return c_make(a.re * n, -a.im * n);
}

__device__ __forceinline__ double shfl_up_t(double v, int s) { return __shfl_up(v, s, 64); }
__device__ __forceinline__ cplx shfl_up_t(cplx v, int s) { return c_make(__shfl_up(v.re, s, 64), __shfl_up(v.im, s, 64)); }
__device__ __forceinline__ double shfl_down_t(double v, int s) { return __shfl_down(v, s, 64); }
__device__ __forceinline__ cplx shfl_down_t(cplx v, int s) { return c_make(__shfl_down(v.re, s, 64), __shfl_down(v.im, s, 64)); }

// sum over the 64 lanes in a fixed tree order; the total is valid in lane 0
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// value at level k-1 for every level a lane owns (fill at k = 0)
template <int E>
__device__ __forceinline__ void shift_prev(const double (&a)[E], double (&o)[E], int lane, double fill) {
    double up = __shfl_up(a[E - 1], 1, 64);
    o[0] = (lane == 0) ? fill : up;
#pragma unroll
    for (int e = 1; e < E; ++e) o[e] = a[e - 1];
}
// value at level k+1 for every level a lane owns (fill past the last lane)
template <int E>
__device__ __forceinline__ void shift_next(const double (&a)[E], double (&o)[E], int lane, double fill) {
    double dn = __shfl_down(a[0], 1, 64);
    o[E - 1] = (lane == 63) ? fill : dn;
#pragma unroll
    for (int e = 0; e < E - 1; ++e) o[e] = a[e + 1];
}

// Memory policy MP of the column accessors.  0: plain loads and stores -- one kernel launch per phase, the
// launch boundary orders everything.  1: agent-coherent accesses (relaxed agent-scope atomics = `sc1` loads and
// stores on gfx950: the load bypasses the CU's L1, the store is written through) for data that OTHER workgroups
// of a persistent launch read or write between two grid barriers (the one-launch years and the command stream): an array accessed with
// MP = 1 anywhere in such a launch must be accessed with MP = 1 everywhere in it.
// MP = 1: data other workgroups anywhere on the chip exchange inside a launch -- write-through (sc1) stores, L1-bypassing
// (sc1) loads.  MP = 2: the same between workgroups that all sit on ONE XCD -- plain stores (they reach, and stay in, the
// XCD's L2 once the wave has waited for them) and the same L1-bypassing loads, which that L2 then serves.
template <int MP>
__device__ __forceinline__ double ld_mp(const double* p) {
    if constexpr (MP >= 1) {
        const unsigned long long bits = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT);
        return __longlong_as_double((long long)bits);
    } else {
        return *p;
    }
}
template <int MP>
__device__ __forceinline__ void st_mp(double* p, double v) {
    if constexpr (MP == 1) {
        __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *p = v;
    }
}

template <int E, int MP = 0>
__device__ __forceinline__ void load_col(const double* __restrict__ base, size_t col, int lane, double (&o)[E]) {
    const double* p = base + col * (size_t)(E * 64) + lane;
#pragma unroll
    for (int e = 0; e < E; ++e) o[e] = ld_mp<MP>(p + e * 64);
}
template <int E, int MP = 0>
__device__ __forceinline__ void store_col(double* __restrict__ base, size_t col, int lane, const double (&v)[E]) {
    double* p = base + col * (size_t)(E * 64) + lane;
#pragma unroll
    for (int e = 0; e < E; ++e) st_mp<MP>(p + e * 64, v[e]);
}

// Tridiagonal solve of one column held by one wave (E consecutive rows per lane):
//     a[i] x[i-1] + d[i] x[i] + c[i] x[i+1] = r[i]
// a, c real; d, r real or complex.  Rows past the column end must be identity rows
// (a = c = 0, d = 1, r = 0).  Partition method: each lane eliminates inside its block
// of E rows (two sweeps), the 64 block-end unknowns form a tridiagonal system that is
// solved by parallel cyclic reduction with wave shuffles, then the interior unknowns
// follow by substitution.  No pivoting: the matrices here are strictly diagonally
// dominant (shifted M-matrices).
// ---------------------------------------------------------------------------------
// The solve is split in two.  The matrix-only part (pivot
// reciprocals, the coupling g to the next lane's first row, the parallel-cyclic-reduction
// multipliers of the 64 block-end unknowns and the final pivot) depends only on
// (shift, Jacobian planes), i.e. on SciPy's "LU" event, while a relaxation solve runs
// sweeps x Newton iterations x steps right-hand sides through it.  `tridiag_factor`
// computes that part once; `tridiag_apply` then needs no division and only two shuffles
// of the right-hand side per PCR level (instead of eight values).
// Table layout per lane: K1[6], K2[6], IB, G  (NK2D_TAB = 14 values of T).
// ---------------------------------------------------------------------------------
#define NK2D_TAB 14

template <int E, typename T>
__device__ __forceinline__ void tridiag_factor(const double (&a)[E], const double (&c)[E], const T (&d)[E],
                                               T (&inv)[E], T (&tab)[NK2D_TAB], int lane) {
    T al[E], be[E];
    T dlast = d[0];
    inv[0] = t_recip(d[0]);
    al[0] = t_from_real(a[0], T());
#pragma unroll
    for (int i = 1; i < E; ++i) {
        T m = t_mulr(inv[i - 1], a[i]);
        T dd = t_nfmar(d[i], m, c[i - 1]);
        dlast = dd;
        inv[i] = t_recip(dd);
        al[i] = t_neg(t_mul(m, al[i - 1]));
    }
    T A, B, C, G;
    if constexpr (E >= 2) {
        be[E - 1] = t_zero(T());
        be[E - 2] = t_from_real(c[E - 2], T());
#pragma unroll
        for (int i = E - 3; i >= 0; --i) {
            T m = t_mulr(inv[i + 1], c[i]);
            al[i] = t_nfma(al[i], m, al[i + 1]);
            be[i] = t_neg(t_mul(m, be[i + 1]));
        }
        T n_al = shfl_down_t(al[0], 1), n_inv = shfl_down_t(inv[0], 1), n_be = shfl_down_t(be[0], 1);
        G = t_mulr(n_inv, c[E - 1]);
        A = al[E - 1];
        B = t_nfma(dlast, G, n_al);
        C = t_neg(t_mul(G, n_be));
    } else {
        G = t_zero(T());
        A = t_from_real(a[0], T());
        B = d[0];
        C = t_from_real(c[0], T());
    }
    int lv = 0;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1, ++lv) {
        T iB = t_recip(B);
        T Am = shfl_up_t(A, s), iBm = shfl_up_t(iB, s), Cm = shfl_up_t(C, s);
        T Ap = shfl_down_t(A, s), iBp = shfl_down_t(iB, s), Cp = shfl_down_t(C, s);
        const bool hm = lane >= s, hp = lane + s < 64;
        T k1 = hm ? t_mul(A, iBm) : t_zero(T());
        T k2 = hp ? t_mul(C, iBp) : t_zero(T());
        if (!hm) { Am = t_zero(T()); Cm = t_zero(T()); }
        if (!hp) { Ap = t_zero(T()); Cp = t_zero(T()); }
        tab[lv] = k1;
        tab[6 + lv] = k2;
        B = t_nfma(t_nfma(B, Cm, k1), Ap, k2);
        A = t_neg(t_mul(Am, k1));
        C = t_neg(t_mul(Cp, k2));
    }
    tab[12] = t_recip(B);
    tab[13] = G;
}

template <int E, typename T>
__device__ __forceinline__ void tridiag_apply(const double (&a)[E], const double (&c)[E], const T (&inv)[E],
                                              const T (&tab)[NK2D_TAB], T (&r)[E], int lane) {
    T al[E], be[E];
    al[0] = t_from_real(a[0], T());
#pragma unroll
    for (int i = 1; i < E; ++i) {
        T m = t_mulr(inv[i - 1], a[i]);
        r[i] = t_nfma(r[i], m, r[i - 1]);
        al[i] = t_neg(t_mul(m, al[i - 1]));
    }
    T R;
    if constexpr (E >= 2) {
        be[E - 1] = t_zero(T());
        be[E - 2] = t_from_real(c[E - 2], T());
#pragma unroll
        for (int i = E - 3; i >= 0; --i) {
            T m = t_mulr(inv[i + 1], c[i]);
            r[i] = t_nfma(r[i], m, r[i + 1]);
            al[i] = t_nfma(al[i], m, al[i + 1]);
            be[i] = t_neg(t_mul(m, be[i + 1]));
        }
        T n_r = shfl_down_t(r[0], 1);
        R = t_nfma(r[E - 1], tab[13], n_r);
    } else {
        R = r[0];
    }
    int lv = 0;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1, ++lv) {
        T Rm = shfl_up_t(R, s), Rp = shfl_down_t(R, s);  // multipliers are 0 where no partner
        R = t_nfma(t_nfma(R, Rm, tab[lv]), Rp, tab[6 + lv]);
    }
    T xl = t_mul(R, tab[12]);
    T xp = shfl_up_t(xl, 1);
    if (lane == 0) xp = t_zero(T());
    if constexpr (E >= 2) {
#pragma unroll
        for (int i = 0; i < E - 1; ++i) {
            T v = t_nfma(t_nfma(r[i], al[i], xp), be[i], xl);
            r[i] = t_mul(inv[i], v);
        }
    }
    r[E - 1] = xl;
}

#endif  // __HIPCC__

// ---------------------------------------------------------------------------------
// host-side entry points implemented across the translation units
// ---------------------------------------------------------------------------------
// nk2d_kernels.hip
int nk2d_k_pack_plane(nk2d_ctx* c, const double* src_dev, int nrows, int ncols, double* dst, double fill);
int nk2d_k_pack_state(nk2d_ctx* c, const double* src_dev, double* dst);
int nk2d_k_unpack_state(nk2d_ctx* c, const double* src, double* dst_dev);
int nk2d_k_unpack_plane(nk2d_ctx* c, const double* src, int nrows, int ncols, double* dst_dev);
int nk2d_k_vmix(nk2d_ctx* c, int nt, const double* times, double* const* out);
int nk2d_k_tend(nk2d_ctx* c, const double* y, const double* kv, double* f);
int nk2d_k_jac(nk2d_ctx* c, const double* kv, const double* ylin);
int nk2d_k_jac_apply(nk2d_ctx* c, const double* v, double* out);
int nk2d_k_sweep(nk2d_ctx* c, bool do_real, bool do_cplx, bool first, double cre, double ccr, double cci,
                 const double* br, const double* bcr, const double* bci, int src);
int nk2d_k_factor(nk2d_ctx* c, bool do_real, bool do_cplx, double cre, double ccr, double cci);
int nk2d_k_reduce(nk2d_ctx* c, int ntasks, int nout, double* host_out);
int nk2d_part_sum(nk2d_ctx* c, int ntasks, double* out, const double* part);
int nk2d_r_err_fused(nk2d_ctx* c, double h, int m, int* buf, double* part);
int nk2d_r_commit_tend(nk2d_ctx* c, const double* kv);
int nk2d_r_step_boundary(nk2d_ctx* c, const double* kv_new, bool do_jac, const double* times, double* const* out,
                         double x0, double x1, double x2, int jac_stage = -1, bool with_tend = true);
int nk2d_r_rows_sum(nk2d_ctx* c, const double* rows, int64_t nrows, double* out);
int nk2d_r_newton_final(nk2d_ctx* c, bool do_stage, bool first, double mreal, double mcr, double mci, int src, bool delta,
                        const double* times, double x0, double x1, double x2, int jac_stage);
int nk2d_r_attempt_setup(nk2d_ctx* c, const double* times, double* const* out, double x0, double x1, double x2,
                         int jac_stage = -1);
double nk2d_fingerprint(const nk2d_ctx* c);
int nk2d_frozen_persistent(nk2d_ctx* c, const double* sched, int64_t n, std::vector<char>* err_rows = nullptr);
int nk2d_frozen_cache_pending(const nk2d_ctx* c);
int64_t nk2d_frozen_cache_bytes(const nk2d_ctx* c);
void nk2d_frozen_cache_free(nk2d_ctx* c);
int nk2d_prof_window_begin(nk2d_ctx* c);
int nk2d_prof_window_end(nk2d_ctx* c);
int nk2d_host_interp(int n, const double* xp, const double* fp, double x, double* out);
int nk2d_sweeps_for(nk2d_ctx* c, double c_real);
bool nk2d_has_lateral(const nk2d_ctx* c);
int nk2d_profile_collect(nk2d_ctx* c);
int nk2d_r_predict(nk2d_ctx* c, double x0, double x1, double x2);
int nk2d_r_newton_fused(nk2d_ctx* c, bool do_stage, bool first, bool do_update, double mreal, double mcr,
                        double mci, int src, bool delta);
int nk2d_r_err_rhs(nk2d_ctx* c, double h);
int nk2d_r_err_rhs2(nk2d_ctx* c, const double* err, double h);
int nk2d_r_err_norm(nk2d_ctx* c, const double* err);
int nk2d_r_copy(nk2d_ctx* c, double* dst, const double* src);
int nk2d_r_wnorm(nk2d_ctx* c, const double* a, const double* b, double ca, double cb, const double* ys);
int nk2d_r_axpy(nk2d_ctx* c, const double* a, double s, const double* b, double* out);
int nk2d_r_final(nk2d_ctx* c, const double* y0, double* out);
int nk2d_r_dense(nk2d_ctx* c, double x, double* out);
// nk2d_radau.hip
int nk2d_hist_sample(nk2d_ctx* c, double t_old, double t_new, bool first);
int nk2d_radau_year(nk2d_ctx* c, nk2d_vec x, nk2d_vec fx, nk2d_stats* stats, const double* replay,
                    int64_t replay_n, double* record, int64_t record_cap, int64_t* record_n, bool replay_own = false);
// nk2d_precond.hip
void nk2d_precond_free(nk2d_ctx* c);
