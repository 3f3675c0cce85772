// nk2d_kernels.hip -- per-phase kernels of the py_driver_2d hot path and their host wrappers: layout conversion,
// vertical-mixing coefficient, tendency, Jacobian planes, line-relaxation sweeps of the shifted systems, the fused
// Newton-iteration launches and the elementwise pieces of the Radau IIA step (device functions: nk2d_bodies.h).
#include "nk2d_bodies.h"
#include "nk2d_stream.h"

// ---------------------------------------------------------------------------------
// layout conversion
// ---------------------------------------------------------------------------------
// src: row-major [nrows][ncols] (row = depth level); dst: packed columns
template <int E>
__global__ void k_pack_plane(const double* __restrict__ src, int nrows, int ncols, double* __restrict__ dst, double fill) {
    TASK_PROLOGUE(ncols)
    double v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int k = lane * E + e;
        v[e] = (k < nrows) ? src[(size_t)k * ncols + task] : fill;
    }
    store_col<E>(dst, task, lane, v);
}
template <int E>
__global__ void k_unpack_plane(const double* __restrict__ src, int nrows, int ncols, double* __restrict__ dst) {
    TASK_PROLOGUE(ncols)
    double v[E];
    load_col<E>(src, task, lane, v);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int k = lane * E + e;
        if (k < nrows) dst[(size_t)k * ncols + task] = v[e];
    }
}
// state (tc, nz, ny) <-> packed [tc*ny] columns
template <int E>
__global__ void k_pack_state(const double* __restrict__ src, int nz, int ny, int ncol, double* __restrict__ dst) {
    TASK_PROLOGUE(ncol)
    const int tr = task / ny, j = task - tr * ny;
    double v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int k = lane * E + e;
        v[e] = (k < nz) ? src[((size_t)tr * nz + k) * ny + j] : 0.0;
    }
    store_col<E>(dst, task, lane, v);
}
template <int E>
__global__ void k_unpack_state(const double* __restrict__ src, int nz, int ny, int ncol, double* __restrict__ dst) {
    TASK_PROLOGUE(ncol)
    const int tr = task / ny, j = task - tr * ny;
    double v[E];
    load_col<E>(src, task, lane, v);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int k = lane * E + e;
        if (k < nz) dst[((size_t)tr * nz + k) * ny + j] = v[e];
    }
}

int nk2d_k_pack_plane(nk2d_ctx* c, const double* src_dev, int nrows, int ncols, double* dst, double fill) {
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_pack_plane<EE>, dim3(nk2d_grid(ncols)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                               src_dev, nrows, ncols, dst, fill));
    NK2D_CHECK(c, hipGetLastError());
    return 0;
}
int nk2d_k_unpack_plane(nk2d_ctx* c, const double* src, int nrows, int ncols, double* dst_dev) {
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_unpack_plane<EE>, dim3(nk2d_grid(ncols)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                               src, nrows, ncols, dst_dev));
    NK2D_CHECK(c, hipGetLastError());
    return 0;
}
int nk2d_k_pack_state(nk2d_ctx* c, const double* src_dev, double* dst) {
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_pack_state<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                               src_dev, c->nz, c->ny, c->ncol, dst));
    NK2D_CHECK(c, hipGetLastError());
    return 0;
}
int nk2d_k_unpack_state(nk2d_ctx* c, const double* src, double* dst_dev) {
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_unpack_state<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                               src, c->nz, c->ny, c->ncol, dst_dev));
    NK2D_CHECK(c, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------
// np.interp restated for the host (numpy/core/src/multiarray/compiled_base.c semantics)
// ---------------------------------------------------------------------------------
int nk2d_host_interp(int n, const double* xp, const double* fp, double x, double* out) {
    return nk2d_hm_interp(n, xp, fp, x, out);
}

template <int E>
__global__ void k_vmix(DevP P, VmixArgs A, int nt) {
    TASK_PROLOGUE(P.ny * nt)
    vmix_body<E>(P, A, task, lane);
}

int nk2d_k_vmix(nk2d_ctx* c, int nt, const double* times, double* const* out) {
    if (nt < 1 || nt > 4) return nk2d_fail(c, "nk2d_k_vmix: nt out of range");
    VmixArgs A;
    for (int i = 0; i < nt; ++i) {
        nk2d_host_interp(4, c->d.bld_tvals, c->d.bld_fvals, times[i], &A.frac[i]);
        A.out[i] = out[i];
    }
    vmix_forcing_args(c, nt, times, A);
    A.bldmin = c->d.bldepth_min; A.y0 = c->d.vmix_log_shallow; A.y1 = c->d.vmix_log_deep;
    A.hw = c->d.vmix_half_width;
    DevP P = make_devp(c);
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_vmix<EE>, dim3(nk2d_grid(c->ny * nt)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                               P, A, nt));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}

template <int E, int KIND>
__global__ void k_tend(DevP P, const double* __restrict__ y, const double* __restrict__ kvp, double* __restrict__ f) {
    TASK_PROLOGUE(P.ncol)
    const int tr = task / P.ny, j = task - tr * P.ny;
    ColCoef<E> cf;
    load_coef<E>(P, j, lane, cf);
    double c[E], cs[E], cn[E], kv[E], out[E];
    load_col<E>(y, task, lane, c);
    load_col<E>(y, (j > 0) ? task - 1 : task, lane, cs);
    load_col<E>(y, (j < P.ny - 1) ? task + 1 : task, lane, cn);
    load_col<E>(kvp, j, lane, kv);
    tend_col<E, KIND>(P, cf, c, cs, cn, kv, tr, lane, out);
    if constexpr (KIND == 2) forced_sources<E>(P, kvp, j, lane, c, out);
    if constexpr (KIND == 1) {
        double u1[E], u2[E];
        phos_load_others<E>(P, tr, j, lane, y, u1, u2);
        phos_sources<E>(P, tr, j, lane, c, u1, u2, cf.dzr, out);
    }
    store_col<E>(f, task, lane, out);
}

int nk2d_k_tend(nk2d_ctx* c, const double* y, const double* kv, double* f) {
    DevP P = make_devp(c);
    NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_tend<EE, KK>), dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                               P, y, kv, f));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}

template <int E>
__global__ void k_jac(DevP P, const double* __restrict__ kvp, double* __restrict__ JL, double* __restrict__ JU,
                      double* __restrict__ JS, double* __restrict__ JN, double* __restrict__ JC,
                      const double* __restrict__ ylin, double* __restrict__ UPR) {
    TASK_PROLOGUE(P.ny)
    jac_body<E, 0>(P, kvp, JL, JU, JS, JN, JC, ylin, UPR, task, lane);
}

// ylin: linearisation state (used by the phosphorus module only)
int nk2d_k_jac(nk2d_ctx* c, const double* kv, const double* ylin) {
    DevP P = make_devp(c);
    if (c->kind == 1 && ylin == nullptr) return nk2d_fail(c, "nk2d_k_jac: the phosphorus Jacobian needs a linearisation state");
    if (c->kind == 2 && c->d.sms_nrec > 0 && c->d.sink_thres > 0.0 && ylin == nullptr)
        return nk2d_fail(c, "nk2d_k_jac: a forced module with a sink threshold needs a linearisation state");
    if (c->kind == 0 || (c->kind == 2 && !(c->d.sms_nrec > 0 && c->d.sink_thres > 0.0))) ylin = nullptr;
    if (c->stream_on) {
        StreamCmd cmd = {};
        cmd.op = NK2D_OP_JAC;
        cmd.u.jac.kvp = kv; cmd.u.jac.JL = c->JL; cmd.u.jac.JU = c->JU; cmd.u.jac.JS = c->JS; cmd.u.jac.JN = c->JN;
        cmd.u.jac.JC = c->JC; cmd.u.jac.ylin = ylin; cmd.u.jac.UPR = c->UPR;
        return nk2d_stream_push(c, cmd, false);
    }
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_jac<EE>, dim3(nk2d_grid(c->ny)), dim3(NK2D_BLOCK), 0, nk2d_s(c), P, kv,
                                               c->JL, c->JU, c->JS, c->JN, c->JC, ylin, c->UPR));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}

// out = J v with the planes of the last k_jac (and, for the phosphorus module, its coupling)
template <int E, int KIND>
__global__ void k_jac_apply(DevP P, SweepArgs A, const double* __restrict__ v, double* __restrict__ out) {
    TASK_PROLOGUE(P.ncol)
    const int tr = task / P.ny, j = task - tr * P.ny;
    double jl[E], ju[E], js[E], jn[E], jc[E], a[E], cc[E], x[E], xs[E], xn[E], xp[E], xq[E], r[E];
    load_col<E>(A.JL, j, lane, jl);
    load_col<E>(A.JU, j, lane, ju);
    load_col<E>(A.JS, j, lane, js);
    load_col<E>(A.JN, j, lane, jn);
    load_col<E>(A.JC, j, lane, jc);
    line_offdiag<E, KIND>(P, tr, lane, jl, ju, a, cc);
    load_col<E>(v, task, lane, x);
    load_col<E>(v, (j > 0) ? task - 1 : task, lane, xs);
    load_col<E>(v, (j < P.ny - 1) ? task + 1 : task, lane, xn);
    shift_prev<E>(x, xp, lane, 0.0);
    shift_next<E>(x, xq, lane, 0.0);
    double upr[E], dzr[E];
    if constexpr (KIND == 1) {
        load_col<E>(P.UPR, j, lane, upr);
        load_col<E>(P.DZR, 0, lane, dzr);
    }
    if constexpr (KIND == 2) load_col<E>(P.UPR, j, lane, upr);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        double d = jc[e] - P.decay[tr];
        if (k == 0) d = d - P.surf[tr];
        if constexpr (KIND == 2) d = d - upr[e];
        if constexpr (KIND == 1) {
            if (tr == 0) d = d - upr[e];
            else if (tr == 1) d = d - P.ph_rd;
            else d = d - (P.ph_rp + ((k < P.nz - 1) ? P.ph_vs * dzr[e] : 0.0));
        }
        r[e] = (((d * x[e] - a[e] * xp[e]) - cc[e] * xq[e]) + js[e] * xs[e]) + jn[e] * xn[e];
    }
    if constexpr (KIND == 1) phos_couple<E>(P, tr, j, lane, v, upr, r);
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = ((lane * E + e) < P.nz) ? r[e] : 0.0;
    store_col<E>(out, task, lane, r);
}

template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK) k_factor(DevP P, SweepArgs A) {
    TASK_PROLOGUE(A.ntasks)
    factor_body<E, KIND>(P, A, task, lane);
}

template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK) k_sweep(DevP P, SweepArgs A) {
    TASK_PROLOGUE(A.ntasks)
    sweep_body<E, KIND, 0>(P, A, task, lane);
}

int nk2d_k_jac_apply(nk2d_ctx* c, const double* v, double* out) {
    SweepArgs A = {};
    fill_factor_args(c, A);
    DevP P = make_devp(c);
    NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_jac_apply<EE, KK>), dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c), P, A, v, out));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}

int nk2d_k_factor(nk2d_ctx* c, bool do_real, bool do_cplx, double cre, double ccr, double cci) {
    c->factor_pending = 0;
    SweepArgs A = {};
    fill_factor_args(c, A);
    A.cre = cre; A.ccr = ccr; A.cci = cci;
    A.nreal = do_real ? c->ncol : 0;
    A.ntasks = A.nreal + (do_cplx ? c->ncol : 0);
    A.first = 0;
    if (A.ntasks == 0) return 0;
    DevP P = make_devp(c);
    NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_factor<EE, KK>), dim3(nk2d_grid(A.ntasks)), dim3(NK2D_BLOCK), 0, nk2d_s(c), P, A));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}

// src = index of the ping-pong buffer holding the previous iterate; the new iterate
// goes to buffer 1-src.
int nk2d_k_sweep(nk2d_ctx* c, bool do_real, bool do_cplx, bool first, double cre, double ccr, double cci,
                 const double* br, const double* bcr, const double* bci, int src) {
    SweepArgs A = {};
    fill_factor_args(c, A);
    A.br = br; A.bcr = bcr; A.bci = bci;
    A.xr_old = c->XR[src]; A.xcr_old = c->XCR[src]; A.xci_old = c->XCI[src];
    A.xr_new = c->XR[1 - src]; A.xcr_new = c->XCR[1 - src]; A.xci_new = c->XCI[1 - src];
    A.cre = cre; A.ccr = ccr; A.cci = cci;
    A.nreal = do_real ? c->ncol : 0;
    A.ntasks = A.nreal + (do_cplx ? c->ncol : 0);
    A.first = first ? 1 : 0;
    if (A.ntasks == 0) return 0;
    if (c->stream_on) {
        StreamCmd cmd = {};
        cmd.op = NK2D_OP_SWEEP;
        cmd.u.sw = A;
        cmd.u.sw.f32 = 0;
        NK2D_TRY(nk2d_stream_push(c, cmd, false));
        c->st.nsweeps++;
        return 0;
    }
    DevP P = make_devp(c);
    const bool sample = false;  // the profiled kernel is k_newton_fused
    if (sample) NK2D_CHECK(c, hipEventRecord(c->prof_ev[c->prof_used], nk2d_s(c)));
    const int wpb = c->sweep_wpb;  // waves per block of the sweep kernel
    NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_sweep<EE, KK>), dim3((A.ntasks + wpb - 1) / wpb), dim3(64 * wpb), 0, nk2d_s(c), P, A));
    NK2D_CHECK(c, hipGetLastError());
    if (sample) {
        NK2D_CHECK(c, hipEventRecord(c->prof_ev[c->prof_used + 1], nk2d_s(c)));
        c->prof_used += 2;
    }
    c->st.nlaunch++;
    c->st.nsweeps++;
    return 0;
}

// fold the finished event pairs into the running average (stream must be idle)
int nk2d_profile_collect(nk2d_ctx* c) {
    for (size_t i = 0; i + 1 < c->prof_used; i += 2) {
        float ms = 0.f;
        NK2D_CHECK(c, hipEventElapsedTime(&ms, c->prof_ev[i], c->prof_ev[i + 1]));
        c->prof_ms_sum += ms;
        c->prof_windows++;
        c->prof_cnt += c->prof_win_launches[i / 2];
    }
    c->prof_used = 0;
    c->prof_win_launches.clear();
    return 0;
}

// sweeps needed for the relative accuracy lin_tol from the tabulated contraction bound
int nk2d_sweeps_for(nk2d_ctx* c, double c_real) {
    if (c->rho_tab.empty()) return 1;
    double pos = std::log10(c_real / c->rho_c0) / c->rho_dlog;
    int k = (int)std::floor(pos);
    if (k < 0) return 400;
    if (k >= (int)c->rho_tab.size()) k = (int)c->rho_tab.size() - 1;
    const double rho = c->rho_tab[k];  // grid point below c_real: rho(c_real) <= rho_tab[k]
    if (rho <= 0.0) return 1;  // no horizontal coupling: the line solve is exact
    if (rho >= 0.999) return 400;
    int m = (int)std::ceil(std::log(c->d.lin_tol) / std::log(rho));
    // m = 1 (rho <= lin_tol: short steps) is a Newton iteration in ONE launch; its update writes the spare stage
    // buffer ZN so that it cannot race with the stage reads of the neighbouring columns (nk2d_r_newton_fused).
    // The integrator raises this to two where it cannot swap buffers (device-side decisions, set_lu).
    if (m < 1) m = 1;
    if (m > 400) m = 400;
    return m;
}

bool nk2d_has_lateral(const nk2d_ctx* c) {
    for (double rho : c->rho_tab)
        if (rho > 0.0) return true;
    return false;
}

// ---------------------------------------------------------------------------------
// fixed-order final reduction of per-task partials: RED[r] = sum_task PART[task*nout + r]
// ---------------------------------------------------------------------------------
__global__ void k_reduce(const double* __restrict__ part, int ntasks, int nout, double* __restrict__ out) {
    __shared__ double sh[NK2D_BLOCK];
    for (int r = 0; r < nout; ++r) {
        double s = 0.0;
        for (int i = threadIdx.x; i < ntasks; i += NK2D_BLOCK) s += part[(size_t)i * nout + r];
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int o = NK2D_BLOCK / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[r] = sh[0];
        __syncthreads();
    }
}

// sum of the per-column partials the producing kernel wrote into pinned host memory; the caller
// has waited for that kernel.  Same association as k_reduce (strided partial sums, then a
// binary tree), so that the host- and the device-controlled integrators see bit-identical norms.
int nk2d_part_sum(nk2d_ctx* c, int ntasks, double* out, const double* part) {
    if (part == nullptr) part = c->hPART;
    if (c->strm) part = nk2d_stream_part_named(c, part);    // (the pinned buffer a command wrote them to under this name)
    *out = nk2d_hm_part_sum(part, ntasks, NK2D_BLOCK);
    return 0;
}

int nk2d_k_reduce(nk2d_ctx* c, int ntasks, int nout, double* host_out) {
    if (c->part_on_host && host_out && nout == 1) {
        // host-controlled integrator: no reduction launch
        if (c->strm && nk2d_stream_part_named(c, c->hPART) != c->hPART) {
            // the partials are a command's (marked before it went out): wait for them, not for the kernel
            NK2D_TRY(nk2d_stream_wait_part(c, c->hPART, ntasks));
        } else {
            NK2D_CHECK(c, hipStreamSynchronize(nk2d_s(c)));
        }
        return nk2d_part_sum(c, ntasks, host_out, nullptr);
    }
    // a result the host waits for goes straight into the pinned, device-visible host buffer: no
    // separate device-to-host copy (a blit kernel of its own on this runtime) behind the reduction
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(NK2D_BLOCK), 0, nk2d_s(c), c->PART, ntasks, nout,
                       host_out ? c->hRED : c->RED);
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    if (host_out) {
        NK2D_CHECK(c, hipStreamSynchronize(nk2d_s(c)));
        std::memcpy(host_out, c->hRED, sizeof(double) * nout);
    }
    return 0;
}

template <int E>
__global__ void k_predict(int ncol, PredictArgs A) {
    TASK_PROLOGUE(ncol)
    predict_body<E>(A, task, lane);
}

template <int E>
__global__ void k_attempt_setup(DevP P, VmixArgs V, int nblk_vmix, PredictArgs A, JacOut J) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    if ((int)blockIdx.x < nblk_vmix) {
        const int task = blockIdx.x * wpb + wave;
        if (task < P.ny * 3) {
            double kv[E];
            vmix_body_kv<E>(P, V, task, lane, kv);
            const int ti = task / P.ny;
            if (ti == J.stage)
                jac_core<E, 0>(P, kv, V.out[ti], J.JL, J.JU, J.JS, J.JN, J.JC, nullptr, nullptr, task - ti * P.ny, lane);
        }
    } else {
        const int task = (blockIdx.x - nblk_vmix) * wpb + wave;
        if (task < P.ncol) predict_body<E>(A, task, lane);
    }
}

template <int E, int KIND, int FACTOR, int STAGE>
__global__ void __launch_bounds__(NK2D_BLOCK) k_newton_fused(DevP P, FusedArgs A) {
    TASK_PROLOGUE(P.ncol)
    newton_fused_body<E, KIND, FACTOR, STAGE, 0>(P, A, task, lane);
}


// the launch that ends a frozen step (FinalArgs): column workgroups first, then the workgroups of the next attempt's
// planes
template <int E, int KIND, int FACTOR, int STAGE>
__global__ void __launch_bounds__(NK2D_BLOCK) k_newton_final(DevP P, FusedArgs A, FinalArgs Fin, VmixArgs V, JacOut J) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    if ((int)blockIdx.x < Fin.nblk_cols) {
        const int task = blockIdx.x * wpb + wave;
        if (task < P.ncol) newton_fused_body<E, KIND, FACTOR, STAGE, 0, 1>(P, A, task, lane, &Fin);
        return;
    }
    const int ptask = (blockIdx.x - Fin.nblk_cols) * wpb + wave;
    if (ptask < P.ny * 3) plane_task<E>(P, V, J, ptask, lane);
}

template <int E, int KIND, int FACTOR, int STAGE>
__global__ void __launch_bounds__(NK2D_BLOCK, 2) k_newton_team(DevP P, FusedArgs A) {
    __shared__ TeamLds<E, 3> S;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int task = (int)blockIdx.x;
    if (task >= P.ncol) return;
    newton_team_body<E, KIND, FACTOR, STAGE, 4, 0>(P, A, S, task, w, lane, nullptr);
}

// pairs: two waves per column (newton_team_body, NW = 2); FIN: the launch also ends a frozen step (plane workgroups of
// 128 threads behind the Fin.nblk_cols column workgroups)
template <int E, int KIND, int FACTOR, int STAGE, int FIN>
__global__ void __launch_bounds__(128, 2) k_newton_pair(DevP P, FusedArgs A, FinalArgs Fin, VmixArgs V, JacOut J) {
    __shared__ TeamLds<E, 0> S;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if constexpr (FIN) {
        if ((int)blockIdx.x >= Fin.nblk_cols) {
            const int ptask = ((int)blockIdx.x - Fin.nblk_cols) * 2 + w;
            if (ptask < P.ny * 3) {
                double kv[E];
                vmix_body_kv<E>(P, V, ptask, lane, kv);
                const int ti = ptask / P.ny;
                if (ti == J.stage)
                    jac_core<E, 0>(P, kv, V.out[ti], J.JL, J.JU, J.JS, J.JN, J.JC, nullptr, nullptr, ptask - ti * P.ny, lane);
            }
            return;
        }
    }
    const int task = (int)blockIdx.x;
    if (task >= P.ncol) return;
    newton_team_body<E, KIND, FACTOR, STAGE, 2, FIN>(P, A, S, task, w, lane, FIN ? &Fin : nullptr);
}

template <int E>
__global__ void k_err_rhs(int ncol, const double* __restrict__ f, const double* __restrict__ z, size_t nv, double h,
                          double* __restrict__ out) {
    TASK_PROLOGUE(ncol)
    err_rhs_body<E, 0>(f, z, nv, h, out, task, lane);
}

template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK) k_err_fused(DevP P, ErrArgs A) {
    TASK_PROLOGUE(P.ncol)
    err_fused_body<E, KIND, 0>(P, A, task, lane);
}

template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK)
    k_err_rhs2(DevP P, const double* __restrict__ y, const double* __restrict__ err, const double* __restrict__ kvp,
               const double* __restrict__ z, size_t nv, double h, double* __restrict__ out) {
    TASK_PROLOGUE(P.ncol)
    err_rhs2_body<E, KIND, 0>(P, y, err, kvp, z, nv, h, out, task, lane);
}

template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK)
    k_commit_tend(DevP P, const double* __restrict__ y, const double* __restrict__ z2, const double* __restrict__ kvp,
                  double* __restrict__ ynew, double* __restrict__ f) {
    TASK_PROLOGUE(P.ncol)
    commit_tend_body<E, KIND, 0>(P, y, z2, kvp, ynew, f, task, lane);
}

template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK) k_step_boundary(DevP P, VmixArgs V, BoundaryArgs B, PredictArgs A) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    int blk = blockIdx.x;
    if (blk < B.nblk_vmix) {
        const int task = blk * wpb + wave;
        if (task < P.ny * 3) {
            double kv[E];
            vmix_body_kv<E>(P, V, task, lane, kv);
            // option "jac_stage": the Jacobian of the coming attempt from the plane of one of ITS stage times -- the wave
            // that has just computed that column derives the Jacobian planes of the column from it
            const int ti = task / P.ny;
            if (ti == B.jac_stage)
                jac_core<E, 0>(P, kv, V.out[ti], B.JL, B.JU, B.JS, B.JN, B.JC, nullptr, nullptr, task - ti * P.ny, lane);
        }
        return;
    }
    blk -= B.nblk_vmix;
    if (blk < B.nblk_jac) {
        const int task = blk * wpb + wave;
        if (task < P.ny) jac_body<E, 0>(P, B.kv_new, B.JL, B.JU, B.JS, B.JN, B.JC, nullptr, nullptr, task, lane);
        return;
    }
    blk -= B.nblk_jac;
    const int task = blk * wpb + wave;
    if (task >= P.ncol) return;
    if (B.with_tend) {
        commit_tend_body<E, KIND, 0>(P, B.y, B.z2, B.kv_new, B.ynew, B.f, task, lane);
    } else {
        double c[E], t0[E];
        load_col<E>(B.y, task, lane, c);
        load_col<E>(B.z2, task, lane, t0);
#pragma unroll
        for (int e = 0; e < E; ++e) c[e] = c[e] + t0[e];
        store_col<E>(B.ynew, task, lane, c);
    }
    predict_body<E>(A, task, lane);
}

// sums of rows of per-column partials: out[r] = sum_i rows[r * n + i] in the association of k_reduce (a frozen year keeps
// the norm partials of the last two Newton iterations of every step and checks them after the year, nk2d_radau.hip)
__global__ void k_rows_sum(const double* __restrict__ rows, int n, double* __restrict__ out) {
    __shared__ double sh[NK2D_BLOCK];
    const double v = block_sum(rows + (size_t)blockIdx.x * n, n, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int nk2d_r_rows_sum(nk2d_ctx* c, const double* rows, int64_t nrows, double* out) {
    hipLaunchKernelGGL(k_rows_sum, dim3((unsigned)nrows), dim3(NK2D_BLOCK), 0, nk2d_s(c), rows, c->ncol, out);
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}

// commit of the step just taken + (optionally) the Jacobian at t_new + set-up of the attempt that starts at t_new
// with stage times `times` (planes into out[0..2]) and dense-output abscissae x0..x2.  Buffers are taken in their
// roles BEFORE the caller swaps them: y_new goes to YOLD, the predicted stage values to ZP.
int nk2d_r_step_boundary(nk2d_ctx* c, const double* kv_new, bool do_jac, const double* times, double* const* out,
                         double x0, double x1, double x2, int jac_stage, bool with_tend) {
    VmixArgs V;
    for (int i = 0; i < 3; ++i) {
        nk2d_host_interp(4, c->d.bld_tvals, c->d.bld_fvals, times[i], &V.frac[i]);
        V.out[i] = out[i];
    }
    vmix_forcing_args(c, 3, times, V);
    V.bldmin = c->d.bldepth_min; V.y0 = c->d.vmix_log_shallow; V.y1 = c->d.vmix_log_deep;
    V.hw = c->d.vmix_half_width;
    BoundaryArgs B;
    B.y = c->Y; B.z2 = c->Z + 2 * c->nv; B.kv_new = kv_new; B.ynew = c->YOLD; B.f = c->F;
    B.JL = c->JL; B.JU = c->JU; B.JS = c->JS; B.JN = c->JN; B.JC = c->JC;
    B.do_jac = do_jac ? 1 : 0;
    B.jac_stage = jac_stage;
    B.with_tend = with_tend ? 1 : 0;
    if (jac_stage >= 0 && do_jac) return nk2d_fail(c, "nk2d_r_step_boundary: Jacobian at t_new and at a stage time requested together");
    B.nblk_vmix = nk2d_grid(c->ny * 3);
    B.nblk_jac = do_jac ? nk2d_grid(c->ny) : 0;
    PredictArgs A;
    A.y = c->YOLD; A.yold = c->Y; A.zp = c->Z; A.z = c->ZP; A.w = c->W;
    A.nv = c->nv;
    A.x0 = x0; A.x1 = x1; A.x2 = x2;
    if (c->stream_on) {
        StreamCmd cmd = {};
        cmd.op = NK2D_OP_BOUNDARY;
        cmd.u.bd.V = V; cmd.u.bd.B = B; cmd.u.bd.A = A;
        return nk2d_stream_push(c, cmd, false);
    }
    DevP P = make_devp(c);
    NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_step_boundary<EE, KK>), dim3(B.nblk_vmix + B.nblk_jac + nk2d_grid(c->ncol)),
                                                         dim3(NK2D_BLOCK), 0, nk2d_s(c), P, V, B, A));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}

template <int E>
__global__ void k_err_norm(DevP P, const double* __restrict__ y, const double* __restrict__ z2p,
                           const double* __restrict__ err, double* __restrict__ part) {
    TASK_PROLOGUE(P.ncol)
    err_norm_body<E, 0>(P, y, z2p, err, part, task, lane);
}

// sum(((ca a + cb b) / (atol + |ys| rtol))^2), used by the initial-step heuristic
template <int E>
__global__ void k_wnorm(DevP P, const double* __restrict__ a, const double* __restrict__ b, double ca, double cb,
                        const double* __restrict__ ys, double* __restrict__ part) {
    TASK_PROLOGUE(P.ncol)
    double aa[E], bb[E], yy[E];
    load_col<E>(a, task, lane, aa);
    if (b) load_col<E>(b, task, lane, bb);
    load_col<E>(ys, task, lane, yy);
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double sc = P.atol + fabs(yy[e]) * P.rtol;
        double v = b ? (ca * aa[e] + cb * bb[e]) : aa[e];
        v = v / sc;
        acc += v * v;
    }
    acc = wave_sum(acc);
    if (lane == 0) part[task] = acc;
}

// out = a + s*b
template <int E>
__global__ void k_axpy(int ncol, const double* __restrict__ a, double s, const double* __restrict__ b,
                       double* __restrict__ out) {
    TASK_PROLOGUE(ncol)
    double aa[E], bb[E];
    load_col<E>(a, task, lane, aa);
    load_col<E>(b, task, lane, bb);
#pragma unroll
    for (int e = 0; e < E; ++e) aa[e] = aa[e] + s * bb[e];
    store_col<E>(out, task, lane, aa);
}

// y(T) - y0 with y(T) = y_old + Q [1,1,1] (radau.py:557-570, ivp.py:718-722), region masked
template <int E>
__global__ void k_final(int ncol, int ny, const double* __restrict__ yold, const double* __restrict__ zp, size_t nv,
                        const double* __restrict__ y0, const int32_t* __restrict__ mask, double* __restrict__ out) {
    TASK_PROLOGUE(ncol)
    const int j = task % ny;
    double yo[E], z0[E], z1[E], z2[E], yy[E];
    load_col<E>(yold, task, lane, yo);
    load_col<E>(zp, task, lane, z0);
    load_col<E>(zp + nv, task, lane, z1);
    load_col<E>(zp + 2 * nv, task, lane, z2);
    load_col<E>(y0, task, lane, yy);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        double q[3];
#pragma unroll
        for (int cidx = 0; cidx < 3; ++cidx) q[cidx] = (z0[e] * cP[0][cidx] + z1[e] * cP[1][cidx]) + z2[e] * cP[2][cidx];
        double v = (q[0] + q[1]) + q[2];
        v = v + yo[e];
        v = v - yy[e];
        const int m = mask[(size_t)j * (E * 64) + e * 64 + lane];
        yo[e] = (m != 0) ? v : 0.0;
    }
    store_col<E>(out, task, lane, yo);
}

// dense output y(t) = y_old + Q [x, x^2, x^3], Q = Z^T P  (radau.py:557-570), for t_eval samples
template <int E>
__global__ void k_dense(int ncol, const double* __restrict__ yold, const double* __restrict__ zp, size_t nv, double x,
                        double* __restrict__ out) {
    TASK_PROLOGUE(ncol)
    double yo[E], z0[E], z1[E], z2[E];
    load_col<E>(yold, task, lane, yo);
    load_col<E>(zp, task, lane, z0);
    load_col<E>(zp + nv, task, lane, z1);
    load_col<E>(zp + 2 * nv, task, lane, z2);
    const double p1 = x, p2 = p1 * x, p3 = p2 * x;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        double q[3];
#pragma unroll
        for (int cidx = 0; cidx < 3; ++cidx) q[cidx] = (z0[e] * cP[0][cidx] + z1[e] * cP[1][cidx]) + z2[e] * cP[2][cidx];
        double v = (q[0] * p1 + q[1] * p2) + q[2] * p3;
        yo[e] = v + yo[e];
    }
    store_col<E>(out, task, lane, yo);
}

int nk2d_r_dense(nk2d_ctx* c, double x, double* out) {
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_dense<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                              c->ncol, c->YOLD, c->ZP, c->nv, x, out));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
int nk2d_r_predict(nk2d_ctx* c, double x0, double x1, double x2) {
    PredictArgs A = predict_args(c, x0, x1, x2);
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_predict<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                               c->ncol, A));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
// stage planes at times[0..2] into out[0..2] and the predicted Z, W in one launch
int nk2d_r_attempt_setup(nk2d_ctx* c, const double* times, double* const* out, double x0, double x1, double x2,
                         int jac_stage) {
    VmixArgs V;
    for (int i = 0; i < 3; ++i) {
        nk2d_host_interp(4, c->d.bld_tvals, c->d.bld_fvals, times[i], &V.frac[i]);
        V.out[i] = out[i];
    }
    vmix_forcing_args(c, 3, times, V);
    V.bldmin = c->d.bldepth_min; V.y0 = c->d.vmix_log_shallow; V.y1 = c->d.vmix_log_deep;
    V.hw = c->d.vmix_half_width;
    PredictArgs A = predict_args(c, x0, x1, x2);
    DevP P = make_devp(c);
    const int nblk_vmix = nk2d_grid(c->ny * 3);
    JacOut J = {c->JL, c->JU, c->JS, c->JN, c->JC, jac_stage};
    if (c->stream_on) {     // a command of the resident kernel instead of a launch (nk2d_stream.h)
        StreamCmd cmd = {};
        cmd.op = NK2D_OP_SETUP;
        cmd.u.su.V = V; cmd.u.su.A = A; cmd.u.su.J = J;
        return nk2d_stream_push(c, cmd, false);
    }
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_attempt_setup<EE>, dim3(nblk_vmix + nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0,
                                              nk2d_s(c), P, V, nblk_vmix, A, J));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}

static int launch_fused(nk2d_ctx* c, const DevP& P, const FusedArgs& A, bool do_factor, bool do_stage) {
    if (c->team == 2) {    // a pair of waves per column (k_newton_pair)
        FinalArgs Fin = {};
        VmixArgs V = {};
        JacOut J = {};
        const dim3 grid(c->ncol), block(128);
        if (do_factor) {
            NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_pair<EE, KK, 1, 1, 0>), grid, block, 0, nk2d_s(c), P, A, Fin, V, J));
        } else if (do_stage) {
            NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_pair<EE, KK, 0, 1, 0>), grid, block, 0, nk2d_s(c), P, A, Fin, V, J));
        } else {
            NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_pair<EE, KK, 0, 0, 0>), grid, block, 0, nk2d_s(c), P, A, Fin, V, J));
        }
    } else if (c->team) {    // one workgroup of four waves per column (k_newton_team)
        if (do_factor) {
            NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_team<EE, KK, 1, 1>), dim3(c->ncol), dim3(NK2D_BLOCK), 0, nk2d_s(c), P, A));
        } else if (do_stage) {
            NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_team<EE, KK, 0, 1>), dim3(c->ncol), dim3(NK2D_BLOCK), 0, nk2d_s(c), P, A));
        } else {
            NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_team<EE, KK, 0, 0>), dim3(c->ncol), dim3(NK2D_BLOCK), 0, nk2d_s(c), P, A));
        }
    } else if (do_factor) {  // always a launch with the stage part
        NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_fused<EE, KK, 1, 1>), dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c), P, A));
    } else if (do_stage || c->kind != 1) {
        NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_fused<EE, KK, 0, 1>), dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c), P, A));
    } else {          // phosphorus, sweep-only launch: the lean instantiation
        NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_fused<EE, 1, 0, 0>), dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c), P, A));
    }
    NK2D_CHECK(c, hipGetLastError());
    return 0;
}

// one launch of the fused Newton iteration; src = ping-pong buffer with the previous
// sweep's iterate, the new iterate goes to 1-src unless do_update consumes it
int nk2d_r_newton_fused(nk2d_ctx* c, bool do_stage, bool first, bool do_update, double mreal, double mcr,
                        double mci, int src, bool delta) {
    FusedArgs A;
    fill_fused_args(c, A, do_stage, first, do_update, mreal, mcr, mci, src, delta);
    // stage and update in ONE launch (single-sweep solve): the update must not overwrite stage values the
    // neighbouring columns are still reading -- it writes the spare buffer, the buffers swap after the launch
    // (a hooked controller that queues whole iterations ahead of a verdict, option "hook_spec_depth": the update of a
    // several-sweep iteration goes the same way, so that it can be dropped)
    const bool swap_z = do_update && (do_stage ? c->single_swap != 0 : c->swap_updates != 0);
    A.st.zout = swap_z ? c->ZN : c->Z;
    A.part = c->part_on_host ? (c->part_cur ? c->part_cur : c->hPART) : (c->part_cur ? c->part_cur : c->PART);
    // pivots / PCR tables of a new (h, J): computed inside the first launch that uses them
    bool do_factor = c->factor_pending != 0;
    if (do_factor && !do_stage) {  // not expected: the first launch after an "LU" event evaluates the stages
        NK2D_TRY(nk2d_k_factor(c, true, true, c->lu_cre, c->lu_ccr, c->lu_cci));
        do_factor = false;
    }
    c->factor_pending = 0;
    DevP P = make_devp(c);
    {
        const double words = fused_words(c, do_stage, first, do_update, delta, do_factor);
        c->sweep_launches++;
        c->fused_bytes_all += 8.0 * words;
        if (!do_factor) {   // launch shapes of the kernel without the factorisation, for nk2d_profile_replay
            const int shape = (do_stage && do_update) ? 0 : (do_stage ? 1 : (do_update ? 2 : 3));
            c->shape_cnt[shape]++;
            c->shape_bytes[shape] += 8.0 * words;
        }
        if (c->win_open) {
            c->win_launches++;
            c->win_bytes += 8.0 * words;
        }
    }
    if (c->stream_on) {
        StreamCmd cmd = {};
        cmd.op = NK2D_OP_NEWTON;
        cmd.flags = do_factor ? NK2D_CMD_FACTOR : 0;
        cmd.u.nf = A;
        // the norm partials of the update come back through pinned memory: marked before the command goes out
        if (do_update && c->part_on_host) cmd.u.nf.part = nk2d_stream_part_take(c, A.part);
        NK2D_TRY(nk2d_stream_push(c, cmd, false));
        c->st.nlaunch--;     // (a command, not a launch)
    } else {
        NK2D_TRY(launch_fused(c, P, A, do_factor, do_stage));
    }
    if (swap_z) std::swap(c->Z, c->ZN);
    c->st.nlaunch++;
    c->st.nsweeps++;
    return 0;
}

// the launch that ends the last Newton iteration of a frozen step AND the step (k_newton_final): next attempt with stage
// times `times`, dense-output abscissae x0..x2, Jacobian from the plane of stage jac_stage (-1: none due).  Buffers
// are swapped into their new roles here: Y <-> YOLD, Z <-> ZN (when this launch also evaluated the stages), the stage
// planes and -- when derived -- the Jacobian planes with their second sets.
int nk2d_r_newton_final(nk2d_ctx* c, bool do_stage, bool first, double mreal, double mcr, double mci, int src, bool delta,
                        const double* times, double x0, double x1, double x2, int jac_stage) {
    if (c->kind == 1) return nk2d_fail(c, "nk2d_r_newton_final: not for modules whose Jacobian reads the state");
    FusedArgs A;
    fill_fused_args(c, A, do_stage, first, true, mreal, mcr, mci, src, delta);
    A.part = c->part_cur ? c->part_cur : c->PART;
    bool do_factor = c->factor_pending != 0;
    if (do_factor && !do_stage) {
        NK2D_TRY(nk2d_k_factor(c, true, true, c->lu_cre, c->lu_ccr, c->lu_cci));
        do_factor = false;
    }
    c->factor_pending = 0;
    FinalArgs Fin;
    Fin.ynew = c->YOLD;
    Fin.znext = do_stage ? c->ZN : c->Z;
    Fin.x0 = x0; Fin.x1 = x1; Fin.x2 = x2;
    Fin.nblk_cols = nk2d_grid(c->ncol);
    VmixArgs V;
    for (int i = 0; i < 3; ++i) {
        nk2d_host_interp(4, c->d.bld_tvals, c->d.bld_fvals, times[i], &V.frac[i]);
        V.out[i] = c->KVN[i];
    }
    V.frac[3] = 0.0; V.out[3] = nullptr;
    vmix_forcing_args(c, 3, times, V);
    V.bldmin = c->d.bldepth_min; V.y0 = c->d.vmix_log_shallow; V.y1 = c->d.vmix_log_deep;
    V.hw = c->d.vmix_half_width;
    JacOut J = {c->JB[0], c->JB[1], c->JB[2], c->JB[3], c->JB[4], jac_stage};
    DevP P = make_devp(c);
    {
        const double words = fused_words(c, do_stage, first, true, delta, do_factor);
        // counted with the path's launches and bytes (end-to-end figures); not in the shape tallies of
        // k_newton_fused<E, KIND, 0, 1>: this is a kernel of its own, with the planes' work in it
        c->sweep_launches++;
        c->fused_bytes_all += 8.0 * words;
    }
    if (c->stream_on) {
        StreamCmd cmd = {};
        cmd.op = NK2D_OP_NEWTON_FINAL;
        cmd.flags = do_factor ? NK2D_CMD_FACTOR : 0;
        cmd.u.fn.nf = A; cmd.u.fn.fin = Fin; cmd.u.fn.V = V; cmd.u.fn.J = J;
        NK2D_TRY(nk2d_stream_push(c, cmd, false));
        c->st.nlaunch--;
    } else {
    const dim3 grid(Fin.nblk_cols + nk2d_grid(c->ny * 3));
#define NK2D_FINAL_LAUNCH(KK)                                                                                              \
    if (do_factor) {                                                                                                       \
        NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_final<EE, KK, 1, 1>), grid, dim3(NK2D_BLOCK), 0, nk2d_s(c), P, A, Fin, V, J)); \
    } else if (do_stage) {                                                                                                 \
        NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_final<EE, KK, 0, 1>), grid, dim3(NK2D_BLOCK), 0, nk2d_s(c), P, A, Fin, V, J)); \
    } else {                                                                                                               \
        NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_final<EE, KK, 0, 0>), grid, dim3(NK2D_BLOCK), 0, nk2d_s(c), P, A, Fin, V, J)); \
    }
    if (c->team == 2) {
        Fin.nblk_cols = c->ncol;
        const dim3 pgrid(c->ncol + (c->ny * 3 + 1) / 2), pblock(128);
#define NK2D_PAIR_FINAL(KK)                                                                                                \
        if (do_factor) {                                                                                                   \
            NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_pair<EE, KK, 1, 1, 1>), pgrid, pblock, 0, nk2d_s(c), P, A, Fin, V, J)); \
        } else if (do_stage) {                                                                                             \
            NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_pair<EE, KK, 0, 1, 1>), pgrid, pblock, 0, nk2d_s(c), P, A, Fin, V, J)); \
        } else {                                                                                                           \
            NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_pair<EE, KK, 0, 0, 1>), pgrid, pblock, 0, nk2d_s(c), P, A, Fin, V, J)); \
        }
        if (c->kind == 2) { NK2D_PAIR_FINAL(2) } else { NK2D_PAIR_FINAL(0) }
#undef NK2D_PAIR_FINAL
    } else if (c->kind == 2) { NK2D_FINAL_LAUNCH(2) } else { NK2D_FINAL_LAUNCH(0) }
#undef NK2D_FINAL_LAUNCH
    NK2D_CHECK(c, hipGetLastError());
    }
    std::swap(c->Y, c->YOLD);
    if (do_stage) std::swap(c->Z, c->ZN);
    for (int i = 0; i < 3; ++i) std::swap(c->KV[i], c->KVN[i]);
    if (jac_stage >= 0) {
        std::swap(c->JL, c->JB[0]); std::swap(c->JU, c->JB[1]); std::swap(c->JS, c->JB[2]);
        std::swap(c->JN, c->JB[3]); std::swap(c->JC, c->JB[4]);
    }
    c->st.nlaunch++;
    c->st.nsweeps++;
    return 0;
}

// Back-to-back replay of the dominant kernel for the roofline line of bench.py.  The timing windows above hold one
// or two launches each (the host reads a norm after every Newton iteration), so the ~4.6 us an event pair costs is a
// quarter of every reading.  Here n launches of ONE shape are queued with nothing between them inside ONE event
// pair, on the state the last forward year left behind (stage values, W, planes, factorisation of its last step):
//   shape 0: stage + sweep + update (the single-launch iteration of a one-sweep solve)
//   shape 1: stage + first sweep (first launch of a two-sweep solve, delta form)
//   shape 2: second sweep + update (its last launch)
// The updates go to scratch (W -> a copy in ZP, Z -> ZN) so that every launch reads the same inputs.  avg_us includes
// the hand-over between consecutive launches, which the per-kernel durations of rocprofv3 do not.
int nk2d_profile_replay(nk2d_ctx* c, int shape, int n, double* avg_us, double* bytes_per_launch) {
    if (shape < 0 || shape > 2 || n < 1) return nk2d_fail(c, "nk2d_profile_replay: shape must be 0..2 and n >= 1");
    NK2D_CHECK(c, hipSetDevice(c->dev));
    if (c->lu_cre == 0.0) return nk2d_fail(c, "nk2d_profile_replay: run a forward year first");
    if (!c->timer_ready) {
        NK2D_CHECK(c, hipEventCreate(&c->timer_ev[0]));
        NK2D_CHECK(c, hipEventCreate(&c->timer_ev[1]));
        c->timer_ready = 1;
    }
    const bool do_stage = shape != 2, do_update = shape != 1, first = shape != 2, delta = shape != 0;
    NK2D_CHECK(c, hipMemcpyAsync(c->ZP, c->W, 3 * c->nv * sizeof(double), hipMemcpyDeviceToDevice, nk2d_s(c)));
    FusedArgs A;
    fill_fused_args(c, A, do_stage, first, do_update, c->lu_cre, c->lu_ccr, c->lu_cci, 0, delta);
    A.st.w = c->ZP;
    A.st.zout = c->ZN;
    DevP P = make_devp(c);
    for (int i = 0; i < 3; ++i) NK2D_TRY(launch_fused(c, P, A, false, do_stage));   // warm-up
    NK2D_CHECK(c, hipEventRecord(c->timer_ev[0], nk2d_s(c)));
    for (int i = 0; i < n; ++i) NK2D_TRY(launch_fused(c, P, A, false, do_stage));
    NK2D_CHECK(c, hipEventRecord(c->timer_ev[1], nk2d_s(c)));
    NK2D_CHECK(c, hipEventSynchronize(c->timer_ev[1]));
    float ms = 0.f;
    NK2D_CHECK(c, hipEventElapsedTime(&ms, c->timer_ev[0], c->timer_ev[1]));
    if (avg_us) *avg_us = 1000.0 * ms / n;
    if (bytes_per_launch) *bytes_per_launch = 8.0 * fused_words(c, do_stage, first, do_update, delta, false);
    return 0;
}

// Timing windows of the dominant kernel: an event pair around the launches of one simplified-
// Newton iteration that are queued back to back with nothing else between them (the first
// iteration of a step attempt: stage+sweep ... sweep+update).  Per-launch time = (elapsed - reading
// of an empty event pair) / launches in the window, so it includes the hand-over from one launch
// to the next, as the per-kernel durations of rocprofv3 do.  Every prof_every-th window is timed.
int nk2d_prof_window_begin(nk2d_ctx* c) {
    c->win_open = 0;
    if (c->stream_on) return 0;     // (no launches to time: the year's kernel is timed as a whole)
    if (c->prof_every <= 0 || c->prof_used + 2 > c->prof_ev.size()) return 0;
    // windows whose first launch also factorises (k_newton_fused<E, KIND, 1>, a different and heavier
    // kernel) are not timed: the windows measure k_newton_fused<E, KIND, 0> only
    if (c->factor_pending) return 0;
    if ((c->win_seq++ % c->prof_every) != 0) return 0;
    NK2D_CHECK(c, hipEventRecord(c->prof_ev[c->prof_used], nk2d_s(c)));
    c->win_open = 1;
    c->win_launches = 0;
    c->win_bytes = 0.0;
    return 0;
}
int nk2d_prof_window_end(nk2d_ctx* c) {
    if (!c->win_open) return 0;
    c->win_open = 0;
    NK2D_CHECK(c, hipEventRecord(c->prof_ev[c->prof_used + 1], nk2d_s(c)));
    c->prof_win_launches.push_back(c->win_launches);
    c->sweep_bytes += c->win_bytes;
    c->prof_used += 2;
    return 0;
}

int nk2d_r_err_rhs(nk2d_ctx* c, double h) {
    if (c->stream_on) {
        StreamCmd cmd = {};
        cmd.op = NK2D_OP_ERR_RHS;
        cmd.u.col.a = c->F; cmd.u.col.b = c->Z; cmd.u.col.nv = c->nv; cmd.u.col.h = h; cmd.u.col.out = c->BR;
        return nk2d_stream_push(c, cmd, false);
    }
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_err_rhs<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                               c->ncol, c->F, c->Z, c->nv, h, c->BR));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
// error estimate of a solve with m <= 2 sweeps in m launches; the solution ends in XR[*buf] and
// its norm partials in PART / hPART
int nk2d_r_err_fused(nk2d_ctx* c, double h, int m, int* buf, double* part) {
    ErrArgs A = {};
    fill_factor_args(c, A.sw);
    A.f = c->F; A.z = c->Z; A.y = c->Y;
    A.nv = c->nv;
    A.h = h;
    A.part = part ? part : (c->part_on_host ? c->hPART : c->PART);
    DevP P = make_devp(c);
    int src = 0;
    for (int it = 0; it < m; ++it) {
        A.sw.xr_old = c->XR[src];
        A.sw.xr_new = c->XR[1 - src];
        A.stage = it;
        A.last = (it == m - 1) ? 1 : 0;
        if (c->stream_on) {
            StreamCmd cmd = {};
            cmd.op = NK2D_OP_ERR;
            cmd.u.err = A;
            if (A.last && c->part_on_host) cmd.u.err.part = nk2d_stream_part_take(c, A.part);
            NK2D_TRY(nk2d_stream_push(c, cmd, false));
            c->st.nsweeps++;
            src = 1 - src;
            continue;
        }
        NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_err_fused<EE, KK>), dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c), P, A));
        NK2D_CHECK(c, hipGetLastError());
        c->st.nlaunch++;
        c->st.nsweeps++;
        src = 1 - src;
    }
    *buf = src;
    c->st.nsolve++;
    return 0;
}

int nk2d_r_err_rhs2(nk2d_ctx* c, const double* err, double h) {
    if (c->stream_on) {
        StreamCmd cmd = {};
        cmd.op = NK2D_OP_ERR_RHS2;
        cmd.u.col.a = c->Y; cmd.u.col.b = err; cmd.u.col.c = c->KV[3]; cmd.u.col.d = c->Z;
        cmd.u.col.nv = c->nv; cmd.u.col.h = h; cmd.u.col.out = c->BR;
        return nk2d_stream_push(c, cmd, false);
    }
    DevP P = make_devp(c);
    NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_err_rhs2<EE, KK>), dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c), P,
                                               c->Y, err, c->KV[3], c->Z, c->nv, h, c->BR));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
// y_new = Y + Z[2] -> YOLD (the spare buffer), F = fun(., y_new) with the plane kv
int nk2d_r_commit_tend(nk2d_ctx* c, const double* kv) {
    DevP P = make_devp(c);
    NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_commit_tend<EE, KK>), dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c), P,
                                               c->Y, c->Z + 2 * c->nv, kv, c->YOLD, c->F));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
// dst <- src, a state vector (the copy of the error vector before the second estimate of a rejected step)
int nk2d_r_copy(nk2d_ctx* c, double* dst, const double* src) {
    if (c->stream_on) {
        StreamCmd cmd = {};
        cmd.op = NK2D_OP_COPY;
        cmd.u.col.a = src; cmd.u.col.out = dst;
        return nk2d_stream_push(c, cmd, false);
    }
    NK2D_CHECK(c, hipMemcpyAsync(dst, src, sizeof(double) * c->nv, hipMemcpyDeviceToDevice, nk2d_s(c)));
    return 0;
}

int nk2d_r_err_norm(nk2d_ctx* c, const double* err) {
    if (c->stream_on && c->part_on_host) {
        StreamCmd cmd = {};
        cmd.op = NK2D_OP_ERR_NORM;
        cmd.u.col.a = c->Y; cmd.u.col.b = c->Z + 2 * c->nv; cmd.u.col.c = err;
        cmd.u.col.part = nk2d_stream_part_take(c, c->hPART);
        return nk2d_stream_push(c, cmd, false);
    }
    DevP P = make_devp(c);
    if (c->strm && c->part_on_host) nk2d_stream_part_forget(c, c->hPART);
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_err_norm<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c), P,
                                               c->Y, c->Z + 2 * c->nv, err, c->part_on_host ? c->hPART : c->PART));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
int nk2d_r_wnorm(nk2d_ctx* c, const double* a, const double* b, double ca, double cb, const double* ys) {
    DevP P = make_devp(c);
    if (c->strm && c->part_on_host) nk2d_stream_part_forget(c, c->hPART);
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_wnorm<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c), P, a,
                                               b, ca, cb, ys, c->part_on_host ? c->hPART : c->PART));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
int nk2d_r_axpy(nk2d_ctx* c, const double* a, double s, const double* b, double* out) {
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_axpy<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                               c->ncol, a, s, b, out));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
int nk2d_r_final(nk2d_ctx* c, const double* y0, double* out) {
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_final<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                               c->ncol, c->ny, c->YOLD, c->ZP, c->nv, y0, c->MASK, out));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}

