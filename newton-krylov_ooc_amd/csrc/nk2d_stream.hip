// nk2d_stream.hip -- the forward year as a command stream: one resident kernel executes the launches of the host-controlled
// Radau year (nk2d_radau.hip) as commands; see nk2d_stream.h.  Replaces, for the year that produces F(x)
// (/root/reference/nk_ooc/py_driver_2d/model_state.py:102-114, scipy/integrate/_ivp/radau.py:399-539), the launch per phase
// and the stream synchronisation per Newton iteration -- not the controller, which is the host's, decision for decision.
#include "nk2d_stream.h"

#include <algorithm>
#include <condition_variable>
#include <cstddef>
#include <deque>
#include <vector>

#define NK2D_PART_RING 16

// ---------------------------------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long ld_pair_sys(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long ld_pair_dev(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The relay: ONE wave (workgroup 0) polls the host's ring over PCIe -- one reader, whatever the number of workgroups -- and
// forwards every complete command into the ring in HBM, which the workgroups poll.  It leaves behind the EXIT command, or
// when nothing has come for the length of the time limit (the host is gone), raising the abort flag.
__device__ __forceinline__ void stream_relay(const StreamArgs& A, int lane) {
    unsigned seq = A.seq0;
    long long t_begin = (long long)__builtin_amdgcn_s_memrealtime();
    long long spins = 0;
    for (;;) {
        const size_t slot = (size_t)(seq % NK2D_RING_SLOTS) * NK2D_CMD_DWORDS;
        const unsigned long long a = ld_pair_sys(A.h_ring + slot + lane);
        const unsigned long long b = ld_pair_sys(A.h_ring + slot + 64 + lane);
        const unsigned long long d = ld_pair_sys(A.h_ring + slot + 128 + lane);
        // (pair 0 says how many pairs the command has -- once ITS stamp is this command's)
        const unsigned head = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)a, 0);
        const int ndw = ((unsigned)__builtin_amdgcn_readlane((int)(unsigned)(a >> 32), 0) == seq) ? (int)(head >> 16) : NK2D_CMD_DWORDS;
        if (__all((int)(((unsigned)(a >> 32) == seq || lane >= ndw) && ((unsigned)(b >> 32) == seq || 64 + lane >= ndw) &&
                        ((unsigned)(d >> 32) == seq || 128 + lane >= ndw)))) {
            if (lane < ndw) __hip_atomic_store(A.d_ring + slot + lane, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (64 + lane < ndw) __hip_atomic_store(A.d_ring + slot + 64 + lane, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (128 + lane < ndw) __hip_atomic_store(A.d_ring + slot + 128 + lane, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int op = (int)(head & 0xffffu);
            if (op == NK2D_OP_EXIT) return;
            ++seq;
            t_begin = (long long)__builtin_amdgcn_s_memrealtime();
            spins = 0;
            continue;
        }
        const int ab = __builtin_amdgcn_readfirstlane(
            (lane == 0) ? __hip_atomic_load(A.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0);
        const bool late = ((++spins & 15) == 0 || A.spin_ticks == 0) &&
                          (long long)__builtin_amdgcn_s_memrealtime() - t_begin > A.spin_ticks;
        if (late || ab != 0) {
            if (lane == 0) {
                __hip_atomic_store(A.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(A.h_status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
    }
}

template <int E, int KIND>
__device__ __forceinline__ void stream_body(const DevP& P, const StreamArgs& A) {
    __shared__ int lds_ok;
    __shared__ StreamCmd cmd;
    const int lane = threadIdx.x & 63;
    const int tw = uni_i((int)(threadIdx.x >> 6));          // the wave's place in its workgroup = its tracer
    if (blockIdx.x == 0) {
        // (no relay where the host writes its commands straight into the ring in HBM)
        if (tw == 0 && A.h_ring != nullptr) stream_relay(A, lane);
        return;
    }
    const int wg = (int)blockIdx.x - 1;
    const int nw = (int)(blockDim.x >> 6);
    const int j0 = wg * A.cpw, j1 = min(j0 + A.cpw, P.ny);   // this workgroup's ypos columns, every tracer of them
    const int left = (wg > 0) ? wg - 1 : -1, right = (wg < A.nwg - 1) ? wg + 1 : -1;
    // the static coefficients of this workgroup's ypos columns (the same for every tracer of a column) in LDS for as long as the
    // kernel runs: otherwise fetched from L2 by every Newton command (see load_coef_lds)
    // ... and W of its columns (every tracer): the wave's own, read and rewritten by every Newton command and by nothing else of
    // this kernel but the predictions of SETUP / BOUNDARY -- loaded when the kernel starts, stored back when it ends (what runs
    // between two kernels of a year finds it in memory)
    extern __shared__ double dyn_lds[];
    const bool w_in_lds = (A.coef_lds & 2) != 0;
    double* const w_base = dyn_lds + (size_t)A.cpw * NK2D_COEF_LDS_DOUBLES(E);
    const size_t nvw = (size_t)P.ncol * (E * 64);
#define ST_WLDS(tr, j) (w_base + ((size_t)((j) - j0) * P.tc + (tr)) * (3 * E * 64))
    auto w_from_memory = [&](int tr, int j) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double wv[E];
            load_col<E>(A.W + r * nvw, tr * P.ny + j, lane, wv);
            w_lds_put<E>(ST_WLDS(tr, j), r, lane, wv);
        }
    };
    if (A.coef_lds) {
        for (int j = j0 + tw; j < j1; j += nw) {
            ColCoef<E> cf;
            load_coef<E>(P, j, lane, cf);
            store_coef_lds<E>(dyn_lds + (size_t)(j - j0) * NK2D_COEF_LDS_DOUBLES(E), lane, cf);
        }
        if (w_in_lds)
            for (int tr = tw; tr < P.tc; tr += nw)
                for (int j = j0; j < j1; ++j) w_from_memory(tr, j);
        __syncthreads();
    }
    unsigned seq = A.seq0;
    int status = 0;
    long long t_cmd = 0, t_exec = 0, t_nb = 0, n_cmd = 0;     // where the time of this workgroup goes (wave 0's clock)
    long long t_op[4] = {0, 0, 0, 0}, n_op[4] = {0, 0, 0, 0};
    // Wave 0 waits for two things between two commands, with ONE polling loop: that both lateral neighbours have completed
    // the command this workgroup has just completed, and that the next command is in the ring (every pair of its slot carries
    // its stamp) -- which it then copies into LDS for the workgroup.  Before the first command there is nobody to wait for.
    // Returns (to wave 0's lanes) 1, or 0 when a wait ran over the time limit or another workgroup has given up.
    auto wait_and_fetch = [&](unsigned done_seq, bool with_neighbours, long long& ticks_nb, long long& ticks_cmd) -> int {
        const unsigned next = done_seq + 1u;
        const size_t slot = (size_t)(next % NK2D_RING_SLOTS) * NK2D_CMD_DWORDS;
        const int other = with_neighbours ? ((lane == 0) ? left : ((lane == 1) ? right : -1)) : -1;
        bool nb_ok = !with_neighbours;
        long long spins = 0;
        const long long t_begin = (long long)__builtin_amdgcn_s_memrealtime();
        long long t_nb_ok = t_begin;
        for (;;) {
            const unsigned long long a = ld_pair_dev(A.d_ring + slot + lane);
            const unsigned long long b = ld_pair_dev(A.d_ring + slot + 64 + lane);
            const unsigned long long d = ld_pair_dev(A.d_ring + slot + 128 + lane);
            if (!nb_ok) {
                unsigned v = done_seq;
                if (other >= 0) v = __hip_atomic_load(A.flags + (size_t)other * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((int)((int)(v - done_seq) >= 0))) { nb_ok = true; t_nb_ok = (long long)__builtin_amdgcn_s_memrealtime(); }
            }
            const unsigned head = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)a, 0);
            const int ndw = ((unsigned)__builtin_amdgcn_readlane((int)(unsigned)(a >> 32), 0) == next) ? (int)(head >> 16) : NK2D_CMD_DWORDS;
            if (nb_ok && __all((int)(((unsigned)(a >> 32) == next || lane >= ndw) && ((unsigned)(b >> 32) == next || 64 + lane >= ndw) &&
                                     ((unsigned)(d >> 32) == next || 128 + lane >= ndw)))) {
                unsigned* dw = reinterpret_cast<unsigned*>(&cmd);
                if (lane < ndw) dw[lane] = (lane == 0) ? (head & 0xffffu) : (unsigned)a;
                if (64 + lane < ndw) dw[64 + lane] = (unsigned)b;
                if (128 + lane < ndw) dw[128 + lane] = (unsigned)d;
                const long long t_end = (long long)__builtin_amdgcn_s_memrealtime();
                ticks_nb += t_nb_ok - t_begin;
                ticks_cmd += t_end - t_nb_ok;
                return 1;
            }
            const int ab = __builtin_amdgcn_readfirstlane(
                (lane == 0) ? __hip_atomic_load(A.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0);
            const bool late = ((++spins & 63) == 0 || A.spin_ticks == 0) &&
                              (long long)__builtin_amdgcn_s_memrealtime() - t_begin > A.spin_ticks;
            if (late || ab != 0) {
                if (lane == 0) __hip_atomic_store(A.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return 0;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    if (tw == 0) {
        const int good = wait_and_fetch(seq - 1u, false, t_nb, t_cmd);
        if (lane == 0) lds_ok = good;
    }
    __syncthreads();
    if (lds_ok == 0) status = 1;
    while (status == 0) {
        const long long t1 = (long long)__builtin_amdgcn_s_memrealtime();
        const int op = uni_i(cmd.op), flags = uni_i(cmd.flags);
        if (op == NK2D_OP_EXIT) {
            if (threadIdx.x == 0)
                __hip_atomic_store(A.flags + (size_t)wg * 32, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        // ---- its work on this workgroup's columns: wave tw takes tracer tw (a one-wave workgroup every tracer in turn)
        if (op == NK2D_OP_NEWTON) {
            for (int tr = tw; tr < P.tc; tr += nw)
                for (int j = j0; j < j1; ++j) {
                    const LdsSrc L = {dyn_lds + (size_t)(j - j0) * NK2D_COEF_LDS_DOUBLES(E), w_in_lds ? ST_WLDS(tr, j) : nullptr, nullptr, nullptr};
                    if (w_in_lds) {
                        if (flags & NK2D_CMD_FACTOR) newton_fused_body<E, KIND, 1, 1, 1, 0, 3>(P, cmd.u.nf, tr * P.ny + j, lane, nullptr, &L);
                        else newton_fused_body<E, KIND, 0, 1, 1, 0, 3>(P, cmd.u.nf, tr * P.ny + j, lane, nullptr, &L);
                    } else if (A.coef_lds) {
                        if (flags & NK2D_CMD_FACTOR) newton_fused_body<E, KIND, 1, 1, 1, 0, 1>(P, cmd.u.nf, tr * P.ny + j, lane, nullptr, &L);
                        else newton_fused_body<E, KIND, 0, 1, 1, 0, 1>(P, cmd.u.nf, tr * P.ny + j, lane, nullptr, &L);
                    } else {
                        if (flags & NK2D_CMD_FACTOR) newton_fused_body<E, KIND, 1, 1, 1>(P, cmd.u.nf, tr * P.ny + j, lane);
                        else newton_fused_body<E, KIND, 0, 1, 1>(P, cmd.u.nf, tr * P.ny + j, lane);
                    }
                }
        } else if (op == NK2D_OP_NEWTON_FINAL) {
            // the last Newton iteration of a frozen step, which also ends the step (y_new, the next attempt's predicted stage
            // values), and the next attempt's planes -- into the second sets of plane buffers: this command's own stage and sweep
            // parts still read the current ones
            const StreamFinal& S = cmd.u.fn;
            for (int tr = tw; tr < P.tc; tr += nw)
                for (int j = j0; j < j1; ++j) {
                    const LdsSrc L = {dyn_lds + (size_t)(j - j0) * NK2D_COEF_LDS_DOUBLES(E), w_in_lds ? ST_WLDS(tr, j) : nullptr, nullptr, nullptr};
                    if (w_in_lds) {
                        if (flags & NK2D_CMD_FACTOR) newton_fused_body<E, KIND, 1, 1, 1, 1, 3>(P, S.nf, tr * P.ny + j, lane, &S.fin, &L);
                        else newton_fused_body<E, KIND, 0, 1, 1, 1, 3>(P, S.nf, tr * P.ny + j, lane, &S.fin, &L);
                    } else if (A.coef_lds) {
                        if (flags & NK2D_CMD_FACTOR) newton_fused_body<E, KIND, 1, 1, 1, 1, 1>(P, S.nf, tr * P.ny + j, lane, &S.fin, &L);
                        else newton_fused_body<E, KIND, 0, 1, 1, 1, 1>(P, S.nf, tr * P.ny + j, lane, &S.fin, &L);
                    } else {
                        if (flags & NK2D_CMD_FACTOR) newton_fused_body<E, KIND, 1, 1, 1, 1>(P, S.nf, tr * P.ny + j, lane, &S.fin);
                        else newton_fused_body<E, KIND, 0, 1, 1, 1>(P, S.nf, tr * P.ny + j, lane, &S.fin);
                    }
                }
            for (int ti = tw; ti < 3; ti += nw)
                for (int j = j0; j < j1; ++j) {
                    double kv[E];
                    vmix_body_kv<E, 1>(P, S.V, ti * P.ny + j, lane, kv);
                    if (ti == S.J.stage)
                        jac_core<E, 1>(P, kv, S.V.out[ti], S.J.JL, S.J.JU, S.J.JS, S.J.JN, S.J.JC, nullptr, nullptr, j, lane);
                }
        } else if (op == NK2D_OP_ERR) {
            for (int tr = tw; tr < P.tc; tr += nw)
                for (int j = j0; j < j1; ++j) err_fused_body<E, KIND, 1>(P, cmd.u.err, tr * P.ny + j, lane);
        } else if (op == NK2D_OP_SETUP) {
            // the mixing planes of the attempt's three stage times for this workgroup's columns (the wave that computes the
            // plane of the Jacobian's stage derives the Jacobian planes of the column from it), then the predicted stage values
            const StreamSetup& S = cmd.u.su;
            for (int ti = tw; ti < 3; ti += nw)
                for (int j = j0; j < j1; ++j) {
                    double kv[E];
                    vmix_body_kv<E, 1>(P, S.V, ti * P.ny + j, lane, kv);
                    if (ti == S.J.stage)
                        jac_core<E, 1>(P, kv, S.V.out[ti], S.J.JL, S.J.JU, S.J.JS, S.J.JN, S.J.JC, nullptr, nullptr, j, lane);
                }
            for (int tr = tw; tr < P.tc; tr += nw)
                for (int j = j0; j < j1; ++j) {
                    predict_body<E, 1>(S.A, tr * P.ny + j, lane);
                    if (w_in_lds) {      // (the prediction wrote W to memory: into the column's LDS copy from there)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        w_from_memory(tr, j);
                    }
                }
        } else if (op == NK2D_OP_BOUNDARY) {
            const StreamBoundary& S = cmd.u.bd;
            for (int ti = tw; ti < 3; ti += nw)
                for (int j = j0; j < j1; ++j) {
                    double kv[E];
                    vmix_body_kv<E, 1>(P, S.V, ti * P.ny + j, lane, kv);
                    if (ti == S.B.jac_stage)
                        jac_core<E, 1>(P, kv, S.V.out[ti], S.B.JL, S.B.JU, S.B.JS, S.B.JN, S.B.JC, nullptr, nullptr, j, lane);
                }
            if (S.B.do_jac && tw == nw - 1)
                for (int j = j0; j < j1; ++j) jac_body<E, 1>(P, S.B.kv_new, S.B.JL, S.B.JU, S.B.JS, S.B.JN, S.B.JC, nullptr, nullptr, j, lane);
            for (int tr = tw; tr < P.tc; tr += nw)
                for (int j = j0; j < j1; ++j) {
                    const int task = tr * P.ny + j;
                    if (S.B.with_tend) {
                        commit_tend_body<E, KIND, 1>(P, S.B.y, S.B.z2, S.B.kv_new, S.B.ynew, S.B.f, task, lane);
                    } else {
                        double c[E], t0[E];
                        load_col<E, 1>(S.B.y, task, lane, c);
                        load_col<E, 1>(S.B.z2, task, lane, t0);
#pragma unroll
                        for (int e = 0; e < E; ++e) c[e] = c[e] + t0[e];
                        store_col<E, 1>(S.B.ynew, task, lane, c);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's y_new is in memory before it reads it back
                    predict_body<E, 1>(S.A, task, lane);
                    if (w_in_lds) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        w_from_memory(tr, j);
                    }
                }
        }
        else if (op == NK2D_OP_SWEEP) {
            const SweepArgs& S = cmd.u.sw;
            const int nvar = S.ntasks / P.ny, nre = S.nreal / P.ny;
            for (int tr = tw; tr < P.tc; tr += nw)
                for (int j = j0; j < j1; ++j) {
                    if (nre > 0) sweep_body<E, KIND, 1>(P, S, j * nvar + tr, lane);
                    if (nvar > nre) sweep_body<E, KIND, 1>(P, S, j * nvar + nre + tr, lane);
                }
        } else if (op == NK2D_OP_JAC) {
            const StreamJac& S = cmd.u.jac;
            if (tw == 0)
                for (int j = j0; j < j1; ++j) jac_body<E, 1>(P, S.kvp, S.JL, S.JU, S.JS, S.JN, S.JC, S.ylin, S.UPR, j, lane);
        } else if (op >= NK2D_OP_ERR_RHS && op <= NK2D_OP_COPY) {
            const StreamColumns& S = cmd.u.col;
            for (int tr = tw; tr < P.tc; tr += nw)
                for (int j = j0; j < j1; ++j) {
                    const int task = tr * P.ny + j;
                    if (op == NK2D_OP_ERR_RHS) err_rhs_body<E, 1>(S.a, S.b, S.nv, S.h, S.out, task, lane);
                    else if (op == NK2D_OP_ERR_RHS2) err_rhs2_body<E, KIND, 1>(P, S.a, S.b, S.c, S.d, S.nv, S.h, S.out, task, lane);
                    else if (op == NK2D_OP_ERR_NORM) err_norm_body<E, 1>(P, S.a, S.b, S.c, S.part, task, lane);
                    else {
                        double v[E];
                        load_col<E, 1>(S.a, task, lane, v);
                        store_col<E, 1>(S.out, task, lane, v);
                    }
                }
        }
        // ---- hand-over: every wave has drained its write-through stores, the workgroup publishes the command it has
        // completed (and, where the host waits for it, stamps pinned memory), waits for its two lateral neighbours and
        // for the next command
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (A.fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        const long long t2 = (long long)__builtin_amdgcn_s_memrealtime();
        if (tw == 0) {
            if (lane == 0) {
                __hip_atomic_store(A.flags + (size_t)wg * 32, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (flags & NK2D_CMD_NOTIFY) __hip_atomic_store(A.h_done + wg, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            const int good = wait_and_fetch(seq, true, t_nb, t_cmd);
            if (lane == 0) lds_ok = good;
        }
        __syncthreads();
        if (A.fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (lds_ok == 0) { status = 1; break; }
        t_exec += t2 - t1; ++n_cmd;
        {
            const int bucket = (op == NK2D_OP_NEWTON_FINAL) ? 1 : ((op >= 2 && op <= 5) ? op - 2 : -1);
            if (bucket >= 0) { t_op[bucket] += t2 - t1; ++n_op[bucket]; }
        }
        ++seq;
    }
    if (w_in_lds) {      // W back to memory: whatever runs behind this kernel (launches, the next kernel of the year) finds it there
        for (int tr = tw; tr < P.tc; tr += nw)
            for (int j = j0; j < j1; ++j) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    double wv[E];
                    w_lds_get<E>(ST_WLDS(tr, j), r, lane, wv);
                    store_col<E>(A.W + r * nvw, tr * P.ny + j, lane, wv);
                }
            }
    }
#undef ST_WLDS
    if (A.prof && threadIdx.x == 0) {
        unsigned long long* pr = A.prof + (size_t)wg * 12;
        pr[0] += (unsigned long long)t_cmd; pr[1] += (unsigned long long)t_exec; pr[2] += (unsigned long long)t_nb;
        pr[3] += (unsigned long long)n_cmd;
        for (int i = 0; i < 4; ++i) { pr[4 + i] += (unsigned long long)t_op[i]; pr[8 + i] += (unsigned long long)n_op[i]; }
    }
    if (status != 0 && threadIdx.x == 0) {
        A.out[0] = (double)status;
        __hip_atomic_store(A.h_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (wg == 0 && threadIdx.x == 0) A.out[1] = (double)seq;
}

template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK) k_stream(DevP P, StreamArgs A) {
    stream_body<E, KIND>(P, A);
}
// The same within 256 registers, so that a SIMD holds TWO waves (option "stream_two_waves"; phosphorus from five levels per
// lane, where a wave of k_stream holds 280 - 390 registers and the chip therefore 1 024 waves: the 1 248 (tracer, ypos)
// columns of phosphorus at 416 x 416 are then two rounds per command on half as many workgroups).  What does not fit the
// registers lives in scratch memory.
template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) k_stream_w2(DevP P, StreamArgs A) {
    stream_body<E, KIND>(P, A);
}

// ---------------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------------
struct nk2d_stream_state {
    unsigned long long* h_ring = nullptr;   // pinned
    unsigned* h_done = nullptr;             // pinned [nwg]
    unsigned* h_status = nullptr;           // pinned [16]
    unsigned long long* d_ring = nullptr;
    char* d_sync = nullptr;                 // abort flag at 0, workgroup flags from 4096 (one 128-byte line each)
    double* d_out = nullptr;
    unsigned long long* d_prof = nullptr;   // [nwg][4], see StreamArgs
    double* h_out = nullptr;                // pinned [8]
    unsigned seq = 0;                       // stamp of the last command pushed
    unsigned done_upto = 0;                 // every workgroup is known to have completed this command
    std::deque<unsigned> notifies;          // commands flagged NOTIFY that have not been waited for yet
    bool running = false, lost = false;
    bool coef_lds = true;                   // the static coefficients of a workgroup's columns in LDS (NK2D_STREAM_COEF_LDS=0: not)
    bool direct = false;                    // the host writes its commands straight into d_ring (large BAR): no relay hop
    int nwg = 0, cpw = 1, nw = 1;
    bool two_waves = false;                 // k_stream_w2 (two waves to a SIMD) instead of k_stream
    int turn_waves = 0;                     // what the running kernel holds of the process-wide turn
    int64_t launches = 0;
    // pinned buffers for the norm partials, handed out in turn: the controller names its buffers (hPART, hPARTB, ...) and
    // reuses a name as soon as IT is done with it -- also when a command writing there was dropped unread (an iteration
    // queued ahead of a verdict) and may not have run yet.  A launch boundary used to order that; here every command with
    // partials gets the next buffer of a ring, its name resolves to that buffer until the name is used again, and a buffer
    // comes round again only NK2D_PART_RING commands with partials later.
    double* pbuf[NK2D_PART_RING] = {nullptr};
    int pnext = 0;
    struct { const double* name; double* buf; } pmap[NK2D_PART_RING] = {};
};

// ONE command-stream kernel at a time in a process: two of them do not fit the chip together (a kernel holds a SIMD per
// wave), and a kernel that is only partly resident waits for workgroups that cannot start.  Held from the launch to the
// EXIT command; contexts driven by other host threads wait their turn (their years then run back to back).
namespace {
std::mutex g_turn_mutex;
std::condition_variable g_turn_cv;
int g_turn_waves = 0;          // waves of the resident kernels running now
int g_turn_kernels = 0;
const int kTurnCapacity = 640; // waves that may be resident together (of 1 024 SIMDs, a wave each at the registers these kernels hold)
}  // namespace
// Resident kernels (a one-launch year, a command-stream kernel) wait inside for workgroups that must all be on the chip: a
// process runs them one at a time -- or side by side where their waves together leave the chip room (small grids: a few dozen
// waves each; the years of several tracer modules then overlap as their launches used to).  `waves`: what the kernel holds;
// given back with the same number.  Contexts driven by other host threads wait their turn.
void nk2d_turn_take(int waves) {
    std::unique_lock<std::mutex> lk(g_turn_mutex);
    g_turn_cv.wait(lk, [waves] { return g_turn_kernels == 0 || g_turn_waves + waves <= kTurnCapacity; });
    g_turn_waves += waves;
    g_turn_kernels += 1;
}
void nk2d_turn_give(int waves) {
    {
        std::lock_guard<std::mutex> lk(g_turn_mutex);
        g_turn_waves -= waves;
        g_turn_kernels -= 1;
    }
    g_turn_cv.notify_all();
}

bool nk2d_stream_running(const nk2d_ctx* c) { return c->strm && c->strm->running; }
// the shape of the context's resident kernel (0 before its first year as a command stream): ypos columns per workgroup, and
// whether it is the flavour with two waves to a SIMD
int nk2d_stream_columns_per_workgroup(const nk2d_ctx* c) { return c->strm ? c->strm->cpw : 0; }
int nk2d_stream_two_waves(const nk2d_ctx* c) { return (c->strm && c->strm->two_waves) ? 1 : 0; }
unsigned nk2d_stream_last_seq(const nk2d_ctx* c) { return c->strm ? c->strm->seq : 0u; }

// which contexts run their years as command streams: every module kind, host-side decisions (no norm hook: a sharded module's
// controller waits for all-reduces), double precision factor tables
int nk2d_stream_eligible(const nk2d_ctx* c) {
    if (!c->stream_years || c->stream_lost >= 2 || c->norm_hook || c->factor_fp32) return 0;
    return 1;
}

// (the two-waves flavour is instantiated where it can matter: state-dependent sources from five levels per lane)
template <int E, int KIND>
constexpr bool kStreamHasTwoWaves = KIND == 1 && E >= 5;

template <int E, int KIND>
static hipError_t stream_launch_one(nk2d_ctx* c, dim3 grid, dim3 block, DevP& P, StreamArgs& A, int* max_blocks, size_t lds_bytes,
                                    bool two_waves) {
    if constexpr (kStreamHasTwoWaves<E, KIND>) {
        if (two_waves) {
            if (max_blocks) return hipOccupancyMaxActiveBlocksPerMultiprocessor(max_blocks, k_stream_w2<E, KIND>, (int)block.x, lds_bytes);
            hipLaunchKernelGGL((k_stream_w2<E, KIND>), grid, block, lds_bytes, c->stream_, P, A);
            return hipGetLastError();
        }
    } else if (two_waves) {
        return hipErrorInvalidValue;
    }
    if (max_blocks) return hipOccupancyMaxActiveBlocksPerMultiprocessor(max_blocks, k_stream<E, KIND>, (int)block.x, lds_bytes);
    hipLaunchKernelGGL((k_stream<E, KIND>), grid, block, lds_bytes, c->stream_, P, A);
    return hipGetLastError();
}
static hipError_t stream_launch(nk2d_ctx* c, dim3 grid, dim3 block, DevP& P, StreamArgs& A, int* max_blocks, size_t lds_bytes,
                                bool two_waves) {
    hipError_t rc = hipErrorInvalidValue;
    NK2D_DISPATCH_EK(c->E, c->kind, rc = (stream_launch_one<EE, KK>(c, grid, block, P, A, max_blocks, lds_bytes, two_waves)));
    return rc;
}
// dynamic shared memory of a launch whose workgroups own cpw ypos columns each
static size_t stream_lds_bytes(const nk2d_ctx* c, int cpw, bool coef_lds) {
    return coef_lds ? sizeof(double) * (size_t)cpw * (NK2D_COEF_LDS_DOUBLES(c->E) + (size_t)c->tc * 3 * c->E * 64) : 0;
}

static int stream_alloc(nk2d_ctx* c) {
    if (c->strm) return 0;
    nk2d_stream_state* S = new nk2d_stream_state();
    c->strm = S;
    S->nw = std::min(c->tc, NK2D_WAVES_PER_BLOCK);
    // ypos columns per workgroup: one where the chip holds a workgroup per column, more where it does not
    DevP P = make_devp(c);
    StreamArgs A = {};
    hipDeviceProp_t prop;
    NK2D_CHECK(c, hipGetDeviceProperties(&prop, c->dev));
    S->coef_lds = std::getenv("NK2D_STREAM_COEF_LDS") == nullptr || std::atoi(std::getenv("NK2D_STREAM_COEF_LDS")) != 0;
    // (the columns per workgroup and the LDS they need depend on each other: one column first, more until the grid fits)
    auto columns_per_workgroup = [&](bool two_waves, bool& coef_lds, int& cpw, int& nwg) -> int {
        for (cpw = 1;; ++cpw) {
            int per_cu = 0;
            NK2D_CHECK(c, stream_launch(c, dim3(1), dim3(64 * S->nw), P, A, &per_cu, stream_lds_bytes(c, cpw, coef_lds), two_waves));
            const int capacity = per_cu * prop.multiProcessorCount - 1;     // (one workgroup is the relay's)
            nwg = (c->ny + cpw - 1) / cpw;
            if (capacity >= nwg) return 0;
            if (cpw >= c->ny) {
                if (coef_lds) { coef_lds = false; cpw = 0; continue; }     // (without the LDS copy, then)
                return nk2d_fail(c, "command stream: the kernel does not fit a compute unit");
            }
        }
    };
    NK2D_TRY(columns_per_workgroup(false, S->coef_lds, S->cpw, S->nwg));
    // option "stream_two_waves": where the one-wave-per-SIMD kernel needs several rounds of columns per command and the flavour
    // that fits two waves to a SIMD needs fewer, that one
    // (measured, phosphorus 416 x 416: free-running year 0.728 -> 0.678 s, frozen year 0.374 -> 0.357 s, a Newton command 27.1 ->
    // 19.8 us per workgroup -- profiles/r04_stream_two_waves_phosphorus.log; value 2 takes the flavour wherever it exists: tests)
    if (c->stream_two_waves && c->kind == 1 && c->E >= 5 && (S->cpw > 1 || c->stream_two_waves >= 2)) {
        bool coef2 = S->coef_lds;
        int cpw2 = 0, nwg2 = 0;
        NK2D_TRY(columns_per_workgroup(true, coef2, cpw2, nwg2));
        if ((cpw2 < S->cpw || c->stream_two_waves >= 2) && cpw2 <= S->cpw && coef2 == S->coef_lds) {
            S->two_waves = true; S->cpw = cpw2; S->nwg = nwg2;
        }
    }
    NK2D_CHECK(c, hipHostMalloc((void**)&S->h_ring, sizeof(unsigned long long) * NK2D_RING_SLOTS * NK2D_CMD_DWORDS));
    NK2D_CHECK(c, hipHostMalloc((void**)&S->h_done, sizeof(unsigned) * S->nwg));
    NK2D_CHECK(c, hipHostMalloc((void**)&S->h_status, sizeof(unsigned) * 16));
    NK2D_CHECK(c, hipHostMalloc((void**)&S->h_out, sizeof(double) * 8));
    for (int i = 0; i < NK2D_PART_RING; ++i) NK2D_CHECK(c, hipHostMalloc((void**)&S->pbuf[i], sizeof(double) * c->ncol));
    // Where the device's memory is visible to the host (large BAR), the ring in HBM is a fine-grained allocation the host
    // writes its commands into directly: the pairs cross PCIe once, as posted writes, instead of being fetched by the relay
    // wave's reads.  Otherwise (or with NK2D_STREAM_RELAY=1) the relay.
    S->direct = prop.isLargeBar != 0 && std::getenv("NK2D_STREAM_RELAY") == nullptr;
    if (S->direct) {
        const hipError_t rc = hipExtMallocWithFlags((void**)&S->d_ring, sizeof(unsigned long long) * NK2D_RING_SLOTS * NK2D_CMD_DWORDS,
                                                    hipDeviceMallocFinegrained);
        if (rc != hipSuccess) { (void)hipGetLastError(); S->direct = false; S->d_ring = nullptr; }
    }
    if (!S->direct) NK2D_CHECK(c, hipMalloc((void**)&S->d_ring, sizeof(unsigned long long) * NK2D_RING_SLOTS * NK2D_CMD_DWORDS));
    NK2D_CHECK(c, hipMalloc((void**)&S->d_sync, 4096 + (size_t)S->nwg * 128));
    NK2D_CHECK(c, hipMalloc((void**)&S->d_out, sizeof(double) * 8));
    NK2D_CHECK(c, hipMalloc((void**)&S->d_prof, sizeof(unsigned long long) * 12 * S->nwg));
    NK2D_CHECK(c, hipMemset(S->d_prof, 0, sizeof(unsigned long long) * 12 * S->nwg));
    std::memset(S->h_ring, 0, sizeof(unsigned long long) * NK2D_RING_SLOTS * NK2D_CMD_DWORDS);
    std::memset(S->h_done, 0, sizeof(unsigned) * S->nwg);
    std::memset(S->h_status, 0, sizeof(unsigned) * 16);
    NK2D_CHECK(c, hipMemset(S->d_ring, 0, sizeof(unsigned long long) * NK2D_RING_SLOTS * NK2D_CMD_DWORDS));
    NK2D_CHECK(c, hipMemset(S->d_sync, 0, 4096 + (size_t)S->nwg * 128));
    NK2D_CHECK(c, hipMemset(S->d_out, 0, sizeof(double) * 8));
    return 0;
}

int nk2d_stream_ready(nk2d_ctx* c) { return stream_alloc(c); }

void nk2d_stream_free(nk2d_ctx* c) {
    nk2d_stream_state* S = c->strm;
    if (!S) return;
    if (S->running) (void)nk2d_stream_pause(c);
    (void)hipStreamSynchronize(c->stream_);
    if (S->h_ring) (void)hipHostFree(S->h_ring);
    if (S->h_done) (void)hipHostFree(S->h_done);
    if (S->h_status) (void)hipHostFree(S->h_status);
    if (S->h_out) (void)hipHostFree(S->h_out);
    for (int i = 0; i < NK2D_PART_RING; ++i)
        if (S->pbuf[i]) (void)hipHostFree(S->pbuf[i]);
    if (S->d_ring) (void)hipFree(S->d_ring);
    if (S->d_sync) (void)hipFree(S->d_sync);
    if (S->d_out) (void)hipFree(S->d_out);
    if (S->d_prof) (void)hipFree(S->d_prof);
    delete S;
    c->strm = nullptr;
}

// dwords of a command with this op (the header and the member of the union it uses)
static int cmd_dwords(int op) {
    size_t body = 0;
    switch (op) {
        case NK2D_OP_SETUP: body = sizeof(StreamSetup); break;
        case NK2D_OP_NEWTON: body = sizeof(FusedArgs); break;
        case NK2D_OP_ERR: body = sizeof(ErrArgs); break;
        case NK2D_OP_BOUNDARY: body = sizeof(StreamBoundary); break;
        case NK2D_OP_SWEEP: body = sizeof(SweepArgs); break;
        case NK2D_OP_NEWTON_FINAL: body = sizeof(StreamFinal); break;
        case NK2D_OP_JAC: body = sizeof(StreamJac); break;
        case NK2D_OP_EXIT: body = 0; break;
        default: body = sizeof(StreamColumns); break;
    }
    return (int)((offsetof(StreamCmd, u) + body + 3) / 4);
}

// (dword 0 carries the op in its low and the number of dwords in its high half: a reader checks the stamps of that many pairs)
static void ring_write(nk2d_stream_state* S, unsigned seq, const StreamCmd& cmd) {
    unsigned dw[NK2D_CMD_DWORDS] = {0};
    std::memcpy(dw, &cmd, sizeof(StreamCmd));
    const int ndw = cmd_dwords(cmd.op);
    dw[0] = (unsigned)cmd.op | ((unsigned)ndw << 16);
    unsigned long long* slot = (S->direct ? S->d_ring : S->h_ring) + (size_t)(seq % NK2D_RING_SLOTS) * NK2D_CMD_DWORDS;
    for (int k = 0; k < ndw; ++k)
        __atomic_store_n(slot + k, ((unsigned long long)seq << 32) | dw[k], __ATOMIC_RELAXED);
    if (S->direct) __builtin_ia32_sfence();      // (write-combined stores over the BAR: out of the CPU's buffers now)
}

// start the kernel (it will find the commands pushed from now on)
static int stream_start(nk2d_ctx* c) {
    nk2d_stream_state* S = c->strm;
    S->turn_waves = (S->nwg + 1) * S->nw;
    nk2d_turn_take(S->turn_waves);
    StreamArgs A = {};
    A.h_ring = S->direct ? nullptr : S->h_ring; A.d_ring = S->d_ring;
    A.abort_flag = (int*)S->d_sync;
    A.flags = (unsigned*)(S->d_sync + 4096);
    A.h_done = S->h_done; A.h_status = S->h_status; A.out = S->d_out;
    A.seq0 = S->seq + 1;
    A.nwg = S->nwg; A.cpw = S->cpw;
    A.spin_ticks = (long long)(c->barrier_timeout_ms * 1.0e5);
    A.fences = c->year_fences;
    A.prof = S->d_prof;
    A.coef_lds = S->coef_lds ? 3 : 0;
    A.W = c->W;
    DevP P = make_devp(c);
    const hipError_t rc = stream_launch(c, dim3(S->nwg + 1), dim3(64 * S->nw), P, A, nullptr, stream_lds_bytes(c, S->cpw, S->coef_lds),
                                        S->two_waves);
    if (rc != hipSuccess) {
        nk2d_turn_give(S->turn_waves);
        NK2D_CHECK(c, rc);
    }
    S->running = true;
    S->launches++;
    c->stream_launches++;
    c->st.nlaunch++;
    return 0;
}

// wait until every workgroup has completed command `seq` (one flagged NOTIFY), bounded by the time limit of the kernel's
// own waits plus a second
int nk2d_stream_wait(nk2d_ctx* c, unsigned seq) {
    nk2d_stream_state* S = c->strm;
    if (!S || S->lost) return NK2D_RC_STREAM_LOST;
    if ((int)(S->done_upto - seq) >= 0) return 0;
    const auto t0 = std::chrono::steady_clock::now();
    const double limit_s = 1.0e-3 * c->barrier_timeout_ms + 1.0;
    long long spins = 0;
    for (int wg = 0; wg < S->nwg; ++wg) {
        for (;;) {
            const unsigned v = __atomic_load_n(S->h_done + wg, __ATOMIC_ACQUIRE);
            if ((int)(v - seq) >= 0) break;
            if ((++spins & 255) == 0) {
                const bool gave_up = __atomic_load_n(S->h_status, __ATOMIC_RELAXED) != 0;
                const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (gave_up || waited > limit_s) {
                    // tell the kernel to leave (the relay sees the EXIT, a workgroup stuck in a wait its own time limit),
                    // wait for it, and hand the year back
                    S->lost = true;
                    c->stream_timeouts++;
                    (void)nk2d_stream_pause(c);
                    (void)hipStreamSynchronize(c->stream_);
                    return NK2D_RC_STREAM_LOST;
                }
            }
        }
    }
    S->done_upto = seq;
    while (!S->notifies.empty() && (int)(seq - S->notifies.front()) >= 0) S->notifies.pop_front();
    return 0;
}

// Norm partials come back through pinned host memory, one double per column, written by the waves of the command that
// computes them.  The host marks every slot before it pushes that command (a NaN no computation produces) and reads a slot
// once the mark is gone: the value itself says that it has arrived -- no flag whose write would have to be ordered behind
// 832 waves' stores on their way over PCIe.
static const unsigned long long kPoison = 0x7FF8DEADBEEFCAFEull;

// the buffer the next command with partials writes to, under the controller's name for it; marked
double* nk2d_stream_part_take(nk2d_ctx* c, const double* name) {
    nk2d_stream_state* S = c->strm;
    double* buf = S->pbuf[S->pnext];
    S->pnext = (S->pnext + 1) % NK2D_PART_RING;
    int slot = -1;
    for (int i = 0; i < NK2D_PART_RING; ++i) {
        if (S->pmap[i].buf == buf) { S->pmap[i].name = nullptr; S->pmap[i].buf = nullptr; }   // (whoever held it has long read it)
        if (S->pmap[i].name == name) slot = i;
    }
    if (slot < 0)
        for (int i = 0; i < NK2D_PART_RING && slot < 0; ++i)
            if (S->pmap[i].name == nullptr) slot = i;
    S->pmap[slot].name = name;
    S->pmap[slot].buf = buf;
    unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
    for (int i = 0; i < c->ncol; ++i) __atomic_store_n(p + i, kPoison, __ATOMIC_RELAXED);
    return buf;
}

// the buffer the controller's name stands for right now (the name itself where no command has taken it)
const double* nk2d_stream_part_named(const nk2d_ctx* c, const double* name) {
    const nk2d_stream_state* S = c->strm;
    if (S)
        for (int i = 0; i < NK2D_PART_RING; ++i)
            if (S->pmap[i].name == name && S->pmap[i].buf) return S->pmap[i].buf;
    return name;
}

// a launch writes partials to the name itself: the name stands for itself again
void nk2d_stream_part_forget(nk2d_ctx* c, const double* name) {
    nk2d_stream_state* S = c->strm;
    if (!S) return;
    for (int i = 0; i < NK2D_PART_RING; ++i)
        if (S->pmap[i].name == name || name == nullptr) { S->pmap[i].name = nullptr; S->pmap[i].buf = nullptr; }   // (null: every name)
}

int nk2d_stream_wait_part(nk2d_ctx* c, const double* name, int n) {
    nk2d_stream_state* S = c->strm;
    if (!S || S->lost) return NK2D_RC_STREAM_LOST;
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(nk2d_stream_part_named(c, name));
    const auto t0 = std::chrono::steady_clock::now();
    const double limit_s = 1.0e-3 * c->barrier_timeout_ms + 1.0;
    long long spins = 0;
    for (int i = 0; i < n; ++i) {
        while (__atomic_load_n(p + i, __ATOMIC_ACQUIRE) == kPoison) {
            if ((++spins & 255) == 0) {
                const bool gave_up = __atomic_load_n(S->h_status, __ATOMIC_RELAXED) != 0;
                const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (gave_up || waited > limit_s) {
                    S->lost = true;
                    c->stream_timeouts++;
                    (void)nk2d_stream_pause(c);
                    (void)hipStreamSynchronize(c->stream_);
                    return NK2D_RC_STREAM_LOST;
                }
            }
        }
    }
    return 0;
}

int nk2d_stream_push(nk2d_ctx* c, StreamCmd& cmd, bool notify, unsigned* seq_out) {
    NK2D_TRY(stream_alloc(c));
    nk2d_stream_state* S = c->strm;
    if (S->lost) return NK2D_RC_STREAM_LOST;
    if (!S->running) NK2D_TRY(stream_start(c));
    const unsigned seq = S->seq + 1;
    // flow control: a slot is rewritten NK2D_RING_SLOTS commands later -- by then every workgroup must be past it.  Every
    // 32nd command asks for completion stamps; the host never runs more than half a ring ahead of the oldest of them
    if ((seq & 31u) == 0) notify = true;
    while (seq - S->done_upto > NK2D_RING_SLOTS / 2 && !S->notifies.empty()) NK2D_TRY(nk2d_stream_wait(c, S->notifies.front()));
    cmd.flags = (cmd.flags & ~NK2D_CMD_NOTIFY) | (notify ? NK2D_CMD_NOTIFY : 0);
    ring_write(S, seq, cmd);
    S->seq = seq;
    if (notify) S->notifies.push_back(seq);
    if (seq_out) *seq_out = seq;
    c->stream_cmds++;
    return 0;
}

// tell the kernel to finish behind the commands pushed so far; nothing is waited for (what the caller queues on the
// context's stream next is ordered behind the kernel by the stream)
int nk2d_stream_pause(nk2d_ctx* c) {
    nk2d_stream_state* S = c->strm;
    if (!S || !S->running) return 0;
    StreamCmd cmd = {};
    cmd.op = NK2D_OP_EXIT;
    const unsigned seq = S->seq + 1;
    ring_write(S, seq, cmd);
    S->seq = seq;
    S->running = false;
    // (the kernel's workgroups all pass the EXIT command: everything before it is complete once the kernel has ended)
    S->notifies.clear();
    nk2d_turn_give(S->turn_waves);
    return 0;
}

// the end of a year: the kernel ends, the host waits for it and learns whether it had given up on the way
int nk2d_stream_end(nk2d_ctx* c) {
    nk2d_stream_state* S = c->strm;
    if (!S) return 0;
    (void)nk2d_stream_pause(c);
    NK2D_CHECK(c, hipMemcpyAsync(S->h_out, S->d_out, sizeof(double) * 8, hipMemcpyDeviceToHost, c->stream_));
    NK2D_CHECK(c, hipStreamSynchronize(c->stream_));
    S->done_upto = S->seq;
    const bool lost = S->lost || S->h_out[0] != 0.0 || __atomic_load_n(S->h_status, __ATOMIC_RELAXED) != 0;
    if (lost) {
        // leave everything as a fresh start would find it
        if (!S->lost) c->stream_timeouts++;
        S->lost = false;
        std::memset(S->h_status, 0, sizeof(unsigned) * 16);
        NK2D_CHECK(c, hipMemset(S->d_sync, 0, 4096));
        NK2D_CHECK(c, hipMemset(S->d_out, 0, sizeof(double) * 8));
        return NK2D_RC_STREAM_LOST;
    }
    return 0;
}

// where the workgroups' time went since the context was created: microseconds per workgroup (mean over the workgroups)
// waiting for commands, executing them, waiting for the lateral neighbours; commands per workgroup
// out[12]: 0..2 as above, 3 commands, 4..7 microseconds executing SETUP / NEWTON / ERR / BOUNDARY commands, 8..11 their counts
int nk2d_stream_profile(nk2d_ctx* c, double* out12) {
    nk2d_stream_state* S = c->strm;
    for (int i = 0; i < 12; ++i) out12[i] = 0.0;
    if (!S) return 0;
    std::vector<unsigned long long> h((size_t)12 * S->nwg);
    NK2D_CHECK(c, hipMemcpy(h.data(), S->d_prof, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    for (int wg = 0; wg < S->nwg; ++wg)
        for (int i = 0; i < 12; ++i) out12[i] += (double)h[(size_t)12 * wg + i];
    for (int i = 0; i < 12; ++i) out12[i] *= ((i < 3 || (i >= 4 && i < 8)) ? 0.01 : 1.0) / S->nwg;      // ticks of 10 ns -> us
    return 0;
}
