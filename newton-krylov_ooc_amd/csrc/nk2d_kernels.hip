// nk2d_kernels.hip -- model kernels of the py_driver_2d hot path for gfx950:
// layout conversion, vertical-mixing coefficient, fused advection/mixing tendency,
// Jacobian planes, line-relaxation sweeps of the shifted systems and the elementwise
// pieces of the Radau IIA step.  One wavefront owns one (tracer, ypos) column, see
// nk2d_common.h.  Compiled with -ffp-contract=off: the tendency and coefficient
// kernels keep the reference's operation order (nk_ooc/py_driver_2d/advection.py:51-76,
// horiz_mix.py:50-71, vert_mix.py:24-87, iage.py:22-41); fused multiply-adds are
// written out explicitly only inside the tridiagonal solves.
#include <atomic>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include "nk2d_common.h"
#include "nk2d_hostmath.h"

#include <cmath>
#include <cstring>

struct DevP {
    int nz, ny, tc, ncol;
    const double *VV, *KH, *WT, *WB, *DZR, *ZM0, *ZM1, *DM, *DMR, *DYR, *BLDMAX;
    double surf[NK2D_MAX_TRACERS], starget[NK2D_MAX_TRACERS], decay[NK2D_MAX_TRACERS], csrc;
    double atol, rtol;
    const int* guard;  // guarded kernels return at once when *guard != 0
    // phosphorus module (kind 1): parameters, light limitation plane, d uptake / d po4 at t_jac
    double ph_hs, ph_mu, ph_sig, ph_rd, ph_rp, ph_vs;
    const double *LIGHT, *UPR;
    // forced module with forcing files (kind 2): record sets, flags, 1 / sink_thres (0: none); np = doubles per
    // plane = offset of the source plane inside a KV bundle (the restoring targets follow at 2 np)
    const double *SMSREC, *RESTREC;
    int f_sms, f_restore;
    double f_thres_r;
    size_t np;
    int xcd;   // 1: XCD-contiguous column ranges (TASK_PROLOGUE_XCD), 0: workgroup b takes block b (option "xcd_map")
};

static DevP make_devp(const nk2d_ctx* c) {
    DevP p;
    p.nz = c->nz; p.ny = c->ny; p.tc = c->tc; p.ncol = c->ncol;
    p.VV = c->VV; p.KH = c->KH; p.WT = c->WT; p.WB = c->WB; p.DZR = c->DZR;
    p.ZM0 = c->ZM0; p.ZM1 = c->ZM1; p.DM = c->DM; p.DMR = c->DMR; p.DYR = c->DYR;
    p.BLDMAX = c->BLDMAX;
    for (int i = 0; i < NK2D_MAX_TRACERS; ++i) {
        p.surf[i] = c->d.surf_rate[i]; p.starget[i] = c->d.surf_target[i]; p.decay[i] = c->d.decay_rate[i];
    }
    p.csrc = c->d.const_src;
    p.atol = c->d.atol; p.rtol = c->d.rtol;
    p.guard = c->cur_guard;
    p.xcd = c->xcd_map;
    p.ph_hs = c->d.phos_params[0]; p.ph_mu = c->d.phos_params[1]; p.ph_sig = c->d.phos_params[2];
    p.ph_rd = c->d.phos_params[3]; p.ph_rp = c->d.phos_params[4]; p.ph_vs = c->d.phos_params[5];
    p.LIGHT = c->LIGHT; p.UPR = c->UPR;
    p.SMSREC = c->SMSREC; p.RESTREC = c->RESTREC;
    p.f_sms = (c->kind == 2) ? c->d.sms_nrec : 0;
    p.f_restore = (c->kind == 2) ? c->d.restore_nrec : 0;
    p.f_thres_r = (c->kind == 2 && c->d.sink_thres > 0.0) ? 1.0 / c->d.sink_thres : 0.0;
    p.np = c->np;
    return p;
}

// Radau IIA constants (scipy/integrate/_ivp/radau.py:11-40, values as evaluated by CPython)
__constant__ double cTI[3][3] = {
    {4.17871859155190428, 0.32768282076106237, 0.52337644549944951},
    {-4.17871859155190428, -0.32768282076106237, 0.47662355450055044},
    {0.50287263494578682, -2.57192694985560522, 0.59603920482822492}};
__constant__ double cT[3][3] = {
    {0.09443876248897524, -0.14125529502095421, 0.03002919410514742},
    {0.25021312296533332, 0.20412935229379994, -0.38294211275726192},
    {1.0, 1.0, 0.0}};
__constant__ double cP[3][3] = {
    {10.048809399827414, -25.62959144707664, 15.580782047249224},
    {-1.382142733160748, 10.296258113743303, -8.914115380582556},
    {0.3333333333333333, -2.6666666666666665, 3.3333333333333335}};
__constant__ double cE[3] = {-10.048809399827414, 1.382142733160748, -0.3333333333333333};

#define GUARD_RETURN(g) \
    if ((g) != nullptr && *(g) != 0) return;

#define TASK_PROLOGUE(ntasks)                                              \
    const int lane = threadIdx.x & 63;                                     \
    const int task = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); \
    if (task >= (ntasks)) return;

// XCD-aware variant for the kernels that read the neighbouring columns: workgroups go to the eight XCDs round-robin
// (blockIdx % 8), each XCD with an L2 of its own.  Workgroup b works on the virtual block (b % 8) * (gridDim / 8) + b / 8,
// so that one XCD owns a contiguous range of columns and a column's neighbours are fetched into the same L2 (all but
// the eight range ends) instead of into two or three of them.  The grid must be a multiple of 8 (nk2d_grid_xcd).
#define TASK_PROLOGUE_XCD(ntasks)                                                          \
    const int lane = threadIdx.x & 63;                                                     \
    const int vblk_ = P.xcd ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x; \
    const int task = vblk_ * (blockDim.x >> 6) + (threadIdx.x >> 6);                       \
    if (task >= (ntasks)) return;

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
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_pack_plane<EE>, dim3(nk2d_grid(ncols)), dim3(NK2D_BLOCK), 0, c->stream,
                                               src_dev, nrows, ncols, dst, fill));
    NK2D_CHECK(c, hipGetLastError());
    return 0;
}
int nk2d_k_unpack_plane(nk2d_ctx* c, const double* src, int nrows, int ncols, double* dst_dev) {
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_unpack_plane<EE>, dim3(nk2d_grid(ncols)), dim3(NK2D_BLOCK), 0, c->stream,
                                               src, nrows, ncols, dst_dev));
    NK2D_CHECK(c, hipGetLastError());
    return 0;
}
int nk2d_k_pack_state(nk2d_ctx* c, const double* src_dev, double* dst) {
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_pack_state<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream,
                                               src_dev, c->nz, c->ny, c->ncol, dst));
    NK2D_CHECK(c, hipGetLastError());
    return 0;
}
int nk2d_k_unpack_state(nk2d_ctx* c, const double* src, double* dst_dev) {
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_unpack_state<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream,
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

// ---------------------------------------------------------------------------------
// vertical mixing coefficient (vert_mix.py:44-87 with spatial_axis.py:136-187)
// ---------------------------------------------------------------------------------
struct VmixArgs {
    double frac[4];
    double* out[4];
    double bldmin, y0, y1, hw;
    // kind 2: bracketing records of the forcing sets at each time, x_new - x_lo and x_hi - x_lo
    int srec[4], rrec[4];
    double sdx[4], sden[4], rdx[4], rden[4];
};

// host: bracketing interval of x in the increasing knots xs[0..n) as scipy's interp1d picks it
// (searchsorted, clipped to [1, n-1]: the end intervals extrapolate)
static void forcing_bracket(int n, const double* xs, double x, int* lo, double* dx, double* den) {
    nk2d_hm_bracket(n, xs, x, lo, dx, den);
}
static void vmix_forcing_args(const nk2d_ctx* c, int nt, const double* times, VmixArgs& A) {
    for (int i = 0; i < 4; ++i) { A.srec[i] = A.rrec[i] = 0; A.sdx[i] = A.rdx[i] = 0.0; A.sden[i] = A.rden[i] = 1.0; }
    if (c->kind != 2) return;
    for (int i = 0; i < nt; ++i) {
        if (c->d.sms_nrec > 0) forcing_bracket(c->d.sms_nrec, c->sms_t, times[i], &A.srec[i], &A.sdx[i], &A.sden[i]);
        if (c->d.restore_nrec > 0)
            forcing_bracket(c->d.restore_nrec, c->rest_t, times[i], &A.rrec[i], &A.rdx[i], &A.rden[i]);
    }
}

__device__ __forceinline__ double ramp2(double x, double x0, double x1, double y0, double y1, double slope) {
    if (x > x1) return y1;
    if (x < x0) return y0;
    if (x == x1) return y1;
    if (x == x0) return y0;
    return slope * (x - x0) + y0;
}

// vertical mixing coefficient of ypos column j at the time whose seasonal fraction is `frac`
template <int E>
__device__ __forceinline__ void vmix_col_regs(const DevP& P, double bldmin, double vy0, double vy1, double hw, double frac,
                                              int j, int lane, double (&kv)[E]) {
    const double bld = bldmin + (P.BLDMAX[j] - bldmin) * frac;
    const double x0 = bld - hw, x1 = bld + hw;
    const double y0 = vy0, y1 = vy1;
    const double slope = (y1 - y0) / (x1 - x0);
    double zm0[E], zm1[E], dm[E], dmr[E], wb[E];
    load_col<E>(P.ZM0, 0, lane, zm0);
    load_col<E>(P.ZM1, 0, lane, zm1);
    load_col<E>(P.DM, 0, lane, dm);
    load_col<E>(P.DMR, 0, lane, dmr);
    load_col<E>(P.WB, j, lane, wb);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        double val = 0.0;
        if (k < P.nz - 1) {
            const double e0 = zm0[e], e1 = zm1[e];
            const double ye0 = ramp2(e0, x0, x1, y0, y1, slope);
            const double ye1 = ramp2(e1, x0, x1, y0, y1, slope);
            double res = 0.5 * (ye0 + ye1);
            const bool in0 = (e0 <= x0) && (x0 < e1);
            const bool in1 = (e0 <= x1) && (x1 < e1);
            if (in0) {
                double s = (x0 - e0) * (0.5 * (ye0 + y0));
                if (in1) {
                    s = s + (x1 - x0) * (0.5 * (y0 + y1));
                    s = s + (e1 - x1) * (0.5 * (y1 + ye1));
                } else {
                    s = s + (e1 - x0) * (0.5 * (y0 + ye1));
                }
                res = s * dmr[e];
            } else if (in1) {
                double s = (x1 - e0) * (0.5 * (ye0 + y1));
                s = s + (e1 - x1) * (0.5 * (y1 + ye1));
                res = s * dmr[e];
            }
            double kk = exp(res);
            const double pec = ((0.5 * dm[e]) * fabs(wb[e])) / kk;
            kk = kk * ((pec > 1.0) ? pec : 1.0);
            val = kk * dmr[e];
        }
        kv[e] = val;
    }
}
template <int E, int MP = 0>
__device__ __forceinline__ void vmix_col(const DevP& P, double bldmin, double vy0, double vy1, double hw, double frac,
                                         double* __restrict__ out, int j, int lane) {
    double kv[E];
    vmix_col_regs<E>(P, bldmin, vy0, vy1, hw, frac, j, lane, kv);
    store_col<E, MP>(out, j, lane, kv);
}

// one (time, ypos column) task of a plane launch; kv: the vertical mixing column it computed, for callers that go on with it
template <int E, int MP = 0>
__device__ __forceinline__ void vmix_body_kv(const DevP& P, const VmixArgs& A, int task, int lane, double (&kv)[E]) {
    const int ti = task / P.ny, j = task - ti * P.ny;
    vmix_col_regs<E>(P, A.bldmin, A.y0, A.y1, A.hw, A.frac[ti], j, lane, kv);
    store_col<E, MP>(A.out[ti], j, lane, kv);
    // forcing fields of the same time (kind 2), linear in time between two records:
    // slope = (y_hi - y_lo) / (x_hi - x_lo), y = slope (x - x_lo) + y_lo  (scipy interp1d, utils.py:529-531)
    if (P.f_sms > 0) {
        double lo[E], hi[E], val[E];
        load_col<E>(P.SMSREC + (size_t)A.srec[ti] * P.np, j, lane, lo);
        load_col<E>(P.SMSREC + (size_t)(A.srec[ti] + 1) * P.np, j, lane, hi);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const double slope = (hi[e] - lo[e]) / A.sden[ti];
            val[e] = ((lane * E + e) < P.nz) ? slope * A.sdx[ti] + lo[e] : 0.0;
        }
        store_col<E>(A.out[ti] + P.np, j, lane, val);
    }
    if (P.f_restore > 0 && lane == 0) {
        const double lo = P.RESTREC[(size_t)A.rrec[ti] * P.ny + j], hi = P.RESTREC[(size_t)(A.rrec[ti] + 1) * P.ny + j];
        const double slope = (hi - lo) / A.rden[ti];
        A.out[ti][2 * P.np + j] = slope * A.rdx[ti] + lo;
    }
}

template <int E, int MP = 0>
__device__ __forceinline__ void vmix_body(const DevP& P, const VmixArgs& A, int task, int lane) {
    double kv[E];
    vmix_body_kv<E, MP>(P, A, task, lane, kv);
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
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_vmix<EE>, dim3(nk2d_grid(c->ny * nt)), dim3(NK2D_BLOCK), 0, c->stream,
                                               P, A, nt));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}

// ---------------------------------------------------------------------------------
// tendency of one column: advection + horizontal mixing + vertical mixing + sources
// ---------------------------------------------------------------------------------
template <int E>
struct ColCoef {
    double vS[E], vN[E], khS[E], khN[E], wT[E], wB[E], dzr[E];
    double dyr;
};

template <int E>
__device__ __forceinline__ void load_coef(const DevP& P, int j, int lane, ColCoef<E>& cf) {
    load_col<E>(P.VV, j, lane, cf.vS);
    load_col<E>(P.VV, j + 1, lane, cf.vN);
    load_col<E>(P.KH, j, lane, cf.khS);
    load_col<E>(P.KH, j + 1, lane, cf.khN);
    load_col<E>(P.WT, j, lane, cf.wT);
    load_col<E>(P.WB, j, lane, cf.wB);
    load_col<E>(P.DZR, 0, lane, cf.dzr);
    cf.dyr = P.DYR[j];
}

// c: own column, cs / cn: columns j-1 / j+1 (any finite values at the walls, their
// face coefficients are zero), kv: vertical mixing coeff between level k and k+1
template <int E, int KIND = 0>
__device__ __forceinline__ void tend_col(const DevP& P, const ColCoef<E>& cf, const double (&c)[E],
                                         const double (&cs)[E], const double (&cn)[E], const double (&kv)[E],
                                         int tr, int lane, double (&out)[E]) {
    double cprev[E], cnext[E], kvprev[E];
    shift_prev<E>(c, cprev, lane, 0.0);
    shift_next<E>(c, cnext, lane, 0.0);
    shift_prev<E>(kv, kvprev, lane, 0.0);
    const double surf = P.surf[tr], starget = P.starget[tr], decay = P.decay[tr];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        // advection, flux form (advection.py:58-74)
        const double fyS = (0.5 * (c[e] + cs[e])) * cf.vS[e];
        const double fyN = (0.5 * (cn[e] + c[e])) * cf.vN[e];
        double t = cf.dyr * (fyS - fyN);
        const double fzT = (0.5 * (c[e] + cprev[e])) * cf.wT[e];
        const double fzB = (0.5 * (cnext[e] + c[e])) * cf.wB[e];
        t = t + cf.dzr[e] * (fzB - fzT);
        // horizontal mixing (horiz_mix.py:60-69)
        const double gS = cf.khS[e] * (c[e] - cs[e]);
        const double gN = cf.khN[e] * (cn[e] - c[e]);
        t = t + cf.dyr * (gN - gS);
        // vertical mixing (vert_mix.py:33-40)
        const double hT = kvprev[e] * (c[e] - cprev[e]);
        const double hB = kv[e] * (cnext[e] - c[e]);
        t = t + cf.dzr[e] * (hB - hT);
        // module sources (iage.py:31-39, forced.py:114-139); kind 2 adds them in forced_sources
        if constexpr (KIND != 2) {
            if (k == 0 && surf != 0.0) t = t + surf * (starget - c[e]);
            if (decay != 0.0) t = t + (-decay * c[e]);
            if (P.csrc != 0.0) t = t + P.csrc;
        }
        out[e] = (k < P.nz) ? t : 0.0;
    }
}

// sources of the forced module with forcing files, in the reference's order (forced.py:125-153): surface
// restoring towards the constant or the time-dependent target, then the constant / decay / file source,
// the latter scaled down where it is a sink and the tracer is below the threshold.  kvb: the KV bundle of
// the evaluation time (source plane at np, restoring targets at 2 np).
template <int E>
__device__ __forceinline__ void forced_sources(const DevP& P, const double* __restrict__ kvb, int j, int lane,
                                               const double (&c)[E], double (&out)[E]) {
    double sms[E];
    if (P.f_sms > 0) load_col<E>(kvb + P.np, j, lane, sms);
    const double surf = P.surf[0], decay = P.decay[0];
    const double target = (P.f_restore > 0) ? kvb[2 * P.np + j] : P.starget[0];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        double t = out[e];
        if (k == 0 && surf != 0.0) t = t + surf * (target - c[e]);
        if (P.csrc != 0.0) t = t + P.csrc;
        if (decay != 0.0) t = t + (-decay * c[e]);
        if (P.f_sms > 0) {
            double s = sms[e];
            if (P.f_thres_r != 0.0) {
                const double tmp = P.f_thres_r * c[e];
                if (s < 0.0 && tmp > 0.0 && tmp < 1.0) s = s * tmp;
            }
            t = t + s;
        }
        out[e] = (k < P.nz) ? t : 0.0;
    }
}

// phosphorus sources added to the transport tendency of tracer tr (0 po4, 1 dop, 2 pop) in the
// reference's order (phosphorus.py:66-88): light- and po4-limited uptake, remineralisation of
// dop and pop, sinking of pop.  po4 / dop / pop: the module's tracers at this wave's ypos column.
template <int E>
__device__ __forceinline__ void phos_tend(const DevP& P, int tr, int j, int lane, const double (&po4)[E],
                                          const double (&dop)[E], const double (&pop)[E], const double (&dzr)[E],
                                          double (&out)[E]) {
    double light[E], popprev[E];
    load_col<E>(P.LIGHT, j, lane, light);
    shift_prev<E>(pop, popprev, lane, 0.0);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        const double lim = po4[e] / (po4[e] + P.ph_hs);
        const double uptake = (P.ph_mu * light[e]) * lim;
        const double dop_remin = P.ph_rd * dop[e], pop_remin = P.ph_rp * pop[e];
        double t = out[e];
        if (tr == 0) {
            t = t - uptake;
            t = t + (dop_remin + pop_remin);
        } else if (tr == 1) {
            t = t + P.ph_sig * uptake;
            t = t - dop_remin;
        } else {
            t = t + (1.0 - P.ph_sig) * uptake;
            t = t - pop_remin;
            const double sT = (k > 0) ? P.ph_vs * popprev[e] : 0.0;
            const double sB = (k < P.nz - 1) ? P.ph_vs * pop[e] : 0.0;
            t = t + dzr[e] * (sT - sB);
        }
        out[e] = (k < P.nz) ? t : 0.0;
    }
}

// The wave of tracer tr already holds its own tracer at column j (`own`, formed as a + b by the
// caller); the other two tracers of the module at that column are a (+ b when b != nullptr):
//   tr 0 (po4): others dop, pop;  tr 1 (dop): others po4, pop;  tr 2 (pop): others po4, dop
template <int E>
__device__ __forceinline__ void phos_load_others(const DevP& P, int tr, int j, int lane, const double* __restrict__ a,
                                                 double (&u1)[E], double (&u2)[E]) {
    const int o1 = (tr == 0) ? 1 : 0, o2 = (tr == 2) ? 1 : 2;
    load_col<E>(a, o1 * P.ny + j, lane, u1);
    load_col<E>(a, o2 * P.ny + j, lane, u2);
}
template <int E>
__device__ __forceinline__ void phos_add(double (&u1)[E], double (&u2)[E], const double (&v1)[E], const double (&v2)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) { u1[e] = u1[e] + v1[e]; u2[e] = u2[e] + v2[e]; }
}
// phosphorus sources of tracer tr from its own values and the two others (in the order above)
template <int E>
__device__ __forceinline__ void phos_sources(const DevP& P, int tr, int j, int lane, const double (&own)[E],
                                             const double (&u1)[E], const double (&u2)[E], const double (&dzr)[E],
                                             double (&out)[E]) {
    if (tr == 0) phos_tend<E>(P, 0, j, lane, own, u1, u2, dzr, out);
    else if (tr == 1) phos_tend<E>(P, 1, j, lane, u1, own, u2, dzr, out);
    else phos_tend<E>(P, 2, j, lane, u1, u2, own, dzr, out);
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
    NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_tend<EE, KK>), dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream,
                                               P, y, kv, f));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}

// ---------------------------------------------------------------------------------
// Jacobian planes (advection.py:111-173, horiz_mix.py:100-142, vert_mix.py:140-182)
// up = d tend[k]/d c[k-1], dn = .../d c[k+1], south = .../d c[j-1], north = .../d c[j+1]
// ---------------------------------------------------------------------------------
template <int E, int MP>
__device__ __forceinline__ void jac_core(const DevP& P, const double (&kv)[E], const double* __restrict__ kvp,
                                         double* __restrict__ JL, double* __restrict__ JU, double* __restrict__ JS,
                                         double* __restrict__ JN, double* __restrict__ JC, const double* __restrict__ ylin,
                                         double* __restrict__ UPR, int task, int lane);

template <int E, int MP = 0>
__device__ __forceinline__ void jac_body(const DevP& P, const double* __restrict__ kvp, double* __restrict__ JL,
                                         double* __restrict__ JU, double* __restrict__ JS, double* __restrict__ JN,
                                         double* __restrict__ JC, const double* __restrict__ ylin,
                                         double* __restrict__ UPR, int task, int lane) {
    double kv[E];
    load_col<E, MP>(kvp, task, lane, kv);
    jac_core<E, MP>(P, kv, kvp, JL, JU, JS, JN, JC, ylin, UPR, task, lane);
}

// the five Jacobian diagonals of ypos column j (tracer independent part) from its vertical mixing column, in registers
template <int E>
__device__ __forceinline__ void jac_cols(const DevP& P, const double (&kv)[E], int j, int lane, double (&up)[E],
                                         double (&dn)[E], double (&so)[E], double (&no)[E], double (&ce)[E]) {
    ColCoef<E> cf;
    load_coef<E>(P, j, lane, cf);
    double kvprev[E];
    shift_prev<E>(kv, kvprev, lane, 0.0);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        const bool valid = k < P.nz;
        const double a_up = (k > 0 && valid) ? (-0.5 * cf.wT[e]) * cf.dzr[e] : 0.0;
        const double a_s = (j > 0 && valid) ? (0.5 * cf.vS[e]) * cf.dyr : 0.0;
        const double a_n = (j < P.ny - 1 && valid) ? (-0.5 * cf.vN[e]) * cf.dyr : 0.0;
        const double a_dn = (k < P.nz - 1) ? (0.5 * cf.wB[e]) * cf.dzr[e] : 0.0;
        const double a_c = ((a_up + a_s) + a_n) + a_dn;
        const double h_s = (j > 0 && valid) ? cf.khS[e] * cf.dyr : 0.0;
        const double h_n = (j < P.ny - 1 && valid) ? cf.khN[e] * cf.dyr : 0.0;
        const double h_c = -(h_s + h_n);
        const double v_up = (k > 0 && valid) ? kvprev[e] * cf.dzr[e] : 0.0;
        const double v_dn = (k < P.nz - 1) ? kv[e] * cf.dzr[e] : 0.0;
        const double v_c = -(v_up + v_dn);
        up[e] = a_up + v_up;
        dn[e] = a_dn + v_dn;
        so[e] = a_s + h_s;
        no[e] = a_n + h_n;
        ce[e] = (a_c + h_c) + v_c;
    }
}

// the same from a vertical mixing column held in registers (kvp: its bundle in memory, read only for the source plane of
// a forced module with a thresholded sink)
template <int E, int MP = 0>
__device__ __forceinline__ void jac_core(const DevP& P, const double (&kv)[E], const double* __restrict__ kvp,
                                         double* __restrict__ JL, double* __restrict__ JU, double* __restrict__ JS,
                                         double* __restrict__ JN, double* __restrict__ JC, const double* __restrict__ ylin,
                                         double* __restrict__ UPR, int task, int lane) {
    const int j = task;
    double up[E], dn[E], so[E], no[E], ce[E];
    jac_cols<E>(P, kv, j, lane, up, dn, so, no, ce);
    store_col<E, MP>(JL, j, lane, up);
    store_col<E, MP>(JU, j, lane, dn);
    store_col<E, MP>(JS, j, lane, so);
    store_col<E, MP>(JN, j, lane, no);
    store_col<E, MP>(JC, j, lane, ce);
    if (ylin != nullptr && P.f_sms > 0) {
        // forced module, file source with a sink threshold: UPR = -d sms / d tracer at the linearisation state
        // and the time of the bundle (forced.py:188-202); zero without a threshold
        double cc[E], sms[E], upr[E];
        load_col<E>(ylin, j, lane, cc);
        load_col<E>(kvp + P.np, j, lane, sms);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const double tmp = P.f_thres_r * cc[e];
            const bool on = P.f_thres_r != 0.0 && sms[e] < 0.0 && tmp > 0.0 && tmp < 1.0;
            upr[e] = (on && (lane * E + e) < P.nz) ? -(P.f_thres_r * sms[e]) : 0.0;
        }
        store_col<E>(UPR, j, lane, upr);
    } else if (ylin != nullptr) {
        // d uptake / d po4 at the linearisation state (phosphorus.py:97-103)
        double po4[E], light[E], upr[E];
        load_col<E>(ylin, j, lane, po4);
        load_col<E>(P.LIGHT, j, lane, light);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const double den = po4[e] + P.ph_hs;
            const double lim_d = P.ph_hs / (den * den);
            upr[e] = ((lane * E + e) < P.nz) ? (P.ph_mu * light[e]) * lim_d : 0.0;
        }
        store_col<E>(UPR, j, lane, upr);
    }
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
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_jac<EE>, dim3(nk2d_grid(c->ny)), dim3(NK2D_BLOCK), 0, c->stream, P, kv,
                                               c->JL, c->JU, c->JS, c->JN, c->JC, ylin, c->UPR));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}

// ---------------------------------------------------------------------------------
// line-relaxation sweep for (c I - J) x = b:
//   x_new[:, j] = T_j^-1 ( b[:, j] + S x_old[:, j-1] + N x_old[:, j+1] ),
//   T_j = tridiag(-JL, c - JC + extra, -JU) of column j.
// Real tasks first, complex tasks after; one wave per (system, tracer, column).
// ---------------------------------------------------------------------------------
struct SweepArgs {
    const double *JL, *JU, *JS, *JN, *JC;
    const double *br, *bcr, *bci;
    const double *xr_old, *xcr_old, *xci_old;
    double *xr_new, *xcr_new, *xci_new;
    // cached factorisation (k_factor)
    double *fr_inv, *fc_invr, *fc_invi, *fr_tab, *fc_tabr, *fc_tabi;
    // single precision copies read by the fused Newton launches (see nk2d_set_option "factor_fp32")
    float *fr_inv32, *fc_invr32, *fc_invi32, *fr_tab32, *fc_tabr32, *fc_tabi32;
    int f32;
    double cre, ccr, cci;
    int nreal, ntasks, first;
};

template <int E>
__device__ __forceinline__ void load_tab(const double* __restrict__ tab, int col, int lane, double (&t)[NK2D_TAB]) {
    const double* p = tab + (size_t)col * (NK2D_TAB * 64) + lane;
#pragma unroll
    for (int i = 0; i < NK2D_TAB; ++i) t[i] = p[i * 64];
}

// fp32-stored copies of the factorisation, widened on load
template <int E>
__device__ __forceinline__ void load_col32(const float* __restrict__ base, size_t col, int lane, double (&o)[E]) {
    const float* p = base + col * (size_t)(E * 64) + lane;
#pragma unroll
    for (int e = 0; e < E; ++e) o[e] = (double)p[e * 64];
}
template <int E>
__device__ __forceinline__ void store_col32(float* __restrict__ base, size_t col, int lane, const double (&v)[E]) {
    float* p = base + col * (size_t)(E * 64) + lane;
#pragma unroll
    for (int e = 0; e < E; ++e) p[e * 64] = (float)v[e];
}
__device__ __forceinline__ void load_tab32(const float* __restrict__ tab, int col, int lane, double (&t)[NK2D_TAB]) {
    const float* p = tab + (size_t)col * (NK2D_TAB * 64) + lane;
#pragma unroll
    for (int i = 0; i < NK2D_TAB; ++i) t[i] = (double)p[i * 64];
}
__device__ __forceinline__ void store_tab32(float* __restrict__ tab, int col, int lane, const double (&t)[NK2D_TAB]) {
    float* p = tab + (size_t)col * (NK2D_TAB * 64) + lane;
#pragma unroll
    for (int i = 0; i < NK2D_TAB; ++i) p[i * 64] = (float)t[i];
}

// sub / super diagonal of the column tridiagonal of tracer tr: -(JL + module part), -JU
template <int E, int KIND>
__device__ __forceinline__ void line_offdiag(const DevP& P, int tr, int lane, const double (&jl)[E], const double (&ju)[E],
                                             double (&a)[E], double (&cc)[E]) {
    double dzr[E];
    if constexpr (KIND == 1) load_col<E>(P.DZR, 0, lane, dzr);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        const bool valid = k < P.nz;
        double lo = jl[e];
        if constexpr (KIND == 1) {
            if (tr == 2 && k > 0) lo = lo + P.ph_vs * dzr[e];  // pop sinking in from above (phosphorus.py:142-150)
        }
        a[e] = valid ? -lo : 0.0;
        cc[e] = valid ? -ju[e] : 0.0;
    }
}

// real part of the diagonal of the column tridiagonal: shift - JC + module terms; identity rows
// past the column end
template <int E, int KIND>
__device__ __forceinline__ void line_diag_from(const DevP& P, const double (&jc)[E], int tr, int j, int lane,
                                               double shift_re, double (&dre)[E]) {
    double upr[E], dzr[E];
    if constexpr (KIND == 1) {
        load_col<E>(P.UPR, j, lane, upr);
        load_col<E>(P.DZR, 0, lane, dzr);
    }
    if constexpr (KIND == 2) load_col<E>(P.UPR, j, lane, upr);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        double d = (shift_re - jc[e]) + P.decay[tr];
        if (k == 0) d = d + P.surf[tr];
        if constexpr (KIND == 2) d = d + upr[e];
        if constexpr (KIND == 1) {
            if (tr == 0) d = d + upr[e];
            else if (tr == 1) d = d + P.ph_rd;
            else d = d + (P.ph_rp + ((k < P.nz - 1) ? P.ph_vs * dzr[e] : 0.0));
        }
        dre[e] = (k < P.nz) ? d : 1.0;
    }
}
template <int E, int KIND, int MP = 0>
__device__ __forceinline__ void line_diag(const DevP& P, const double* __restrict__ JC, int tr, int j, int lane,
                                          double shift_re, double (&dre)[E]) {
    double jc[E];
    load_col<E, MP>(JC, j, lane, jc);
    line_diag_from<E, KIND>(P, jc, tr, j, lane, shift_re, dre);
}

// coupling between the tracers of the phosphorus module, kept on the right-hand side of the
// line relaxation: r += (d tend[tr] / d other tracers) * x_old  (phosphorus.py:119-140)
template <int E>
__device__ __forceinline__ void phos_couple(const DevP& P, int tr, int j, int lane, const double* __restrict__ xold,
                                            const double (&upr)[E], double (&r)[E]) {
    if (tr == 0) {
        double x1[E], x2[E];
        load_col<E>(xold, P.ny + j, lane, x1);
        load_col<E>(xold, 2 * P.ny + j, lane, x2);
#pragma unroll
        for (int e = 0; e < E; ++e) r[e] = __builtin_fma(P.ph_rd, x1[e], __builtin_fma(P.ph_rp, x2[e], r[e]));
    } else {
        double x0[E];
        load_col<E>(xold, j, lane, x0);
        const double frac = (tr == 1) ? P.ph_sig : 1.0 - P.ph_sig;
#pragma unroll
        for (int e = 0; e < E; ++e) r[e] = __builtin_fma(frac * upr[e], x0[e], r[e]);
    }
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

// pivots and PCR tables of every column's tridiagonal T_j = tridiag(-JL, c - JC + extra, -JU)
// for the real and/or the complex shift; one launch per SciPy "LU" event
template <int E, int KIND>
__device__ __forceinline__ void factor_body(const DevP& P, const SweepArgs& A, int task, int lane) {
    // the (system, tracer) variants of one ypos column sit in adjacent waves of a block, so
    // that their identical Jacobian-plane loads hit in the CU's L1
    const int nvar = A.ntasks / P.ny, j = task / nvar, var = task - j * nvar;
    const bool is_c = var >= A.nreal / P.ny;
    const int tr = is_c ? var - A.nreal / P.ny : var;
    const int col = tr * P.ny + j;
    double jl[E], ju[E], a[E], cc[E], dre[E];
    load_col<E>(A.JL, j, lane, jl);
    load_col<E>(A.JU, j, lane, ju);
    line_offdiag<E, KIND>(P, tr, lane, jl, ju, a, cc);
    line_diag<E, KIND>(P, A.JC, tr, j, lane, is_c ? A.ccr : A.cre, dre);
    if (!is_c) {
        double inv[E], tab[NK2D_TAB];
        tridiag_factor<E, double>(a, cc, dre, inv, tab, lane);
        store_col<E>(A.fr_inv, col, lane, inv);
        double* p = A.fr_tab + (size_t)col * (NK2D_TAB * 64) + lane;
#pragma unroll
        for (int i = 0; i < NK2D_TAB; ++i) p[i * 64] = tab[i];
        if (A.f32) {
            store_col32<E>(A.fr_inv32, col, lane, inv);
            store_tab32(A.fr_tab32, col, lane, tab);
        }
    } else {
        cplx d[E], inv[E], tab[NK2D_TAB];
#pragma unroll
        for (int e = 0; e < E; ++e) d[e] = c_make(dre[e], ((lane * E + e) < P.nz) ? A.cci : 0.0);
        tridiag_factor<E, cplx>(a, cc, d, inv, tab, lane);
        double re[E], im[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { re[e] = inv[e].re; im[e] = inv[e].im; }
        store_col<E>(A.fc_invr, col, lane, re);
        store_col<E>(A.fc_invi, col, lane, im);
        double* pr = A.fc_tabr + (size_t)col * (NK2D_TAB * 64) + lane;
        double* pi = A.fc_tabi + (size_t)col * (NK2D_TAB * 64) + lane;
        double tre[NK2D_TAB], tim[NK2D_TAB];
#pragma unroll
        for (int i = 0; i < NK2D_TAB; ++i) { pr[i * 64] = tab[i].re; pi[i * 64] = tab[i].im; tre[i] = tab[i].re; tim[i] = tab[i].im; }
        if (A.f32) {
            store_col32<E>(A.fc_invr32, col, lane, re);
            store_col32<E>(A.fc_invi32, col, lane, im);
            store_tab32(A.fc_tabr32, col, lane, tre);
            store_tab32(A.fc_tabi32, col, lane, tim);
        }
    }
}

template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK) k_factor(DevP P, SweepArgs A) {
    TASK_PROLOGUE(A.ntasks)
    factor_body<E, KIND>(P, A, task, lane);
}

template <int E, int KIND, int MP = 0>
__device__ __forceinline__ void sweep_body(const DevP& P, const SweepArgs& A, int task, int lane) {
    const int nvar = A.ntasks / P.ny, j = task / nvar, var = task - j * nvar;
    const bool is_c = var >= A.nreal / P.ny;
    const int tr = is_c ? var - A.nreal / P.ny : var;
    const int col = tr * P.ny + j;
    double jl[E], ju[E], a[E], cc[E];
    load_col<E, MP>(A.JL, j, lane, jl);
    load_col<E, MP>(A.JU, j, lane, ju);
    line_offdiag<E, KIND>(P, tr, lane, jl, ju, a, cc);
    const int cs_col = (j > 0) ? col - 1 : col, cn_col = (j < P.ny - 1) ? col + 1 : col;
    double js[E], jn[E], upr[E];
    if (!A.first) {
        load_col<E, MP>(A.JS, j, lane, js);
        load_col<E, MP>(A.JN, j, lane, jn);
        if constexpr (KIND == 1) load_col<E>(P.UPR, j, lane, upr);
    }
    if (!is_c) {
        double r[E], inv[E], tab[NK2D_TAB];
        load_col<E>(A.br, col, lane, r);
        load_col<E>(A.fr_inv, col, lane, inv);
        load_tab<E>(A.fr_tab, col, lane, tab);
        if (!A.first) {
            double xs[E], xn[E];
            load_col<E, MP>(A.xr_old, cs_col, lane, xs);
            load_col<E, MP>(A.xr_old, cn_col, lane, xn);
#pragma unroll
            for (int e = 0; e < E; ++e) r[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], r[e]));
            if constexpr (KIND == 1) phos_couple<E>(P, tr, j, lane, A.xr_old, upr, r);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) r[e] = ((lane * E + e) < P.nz) ? r[e] : 0.0;
        tridiag_apply<E, double>(a, cc, inv, tab, r, lane);
        store_col<E, MP>(A.xr_new, col, lane, r);
    } else {
        cplx r[E], inv[E], tab[NK2D_TAB];
        double rr[E], ri[E], t0[E], t1[E], tr0[NK2D_TAB], ti0[NK2D_TAB];
        load_col<E>(A.bcr, col, lane, rr);
        load_col<E>(A.bci, col, lane, ri);
        load_col<E>(A.fc_invr, col, lane, t0);
        load_col<E>(A.fc_invi, col, lane, t1);
        load_tab<E>(A.fc_tabr, col, lane, tr0);
        load_tab<E>(A.fc_tabi, col, lane, ti0);
#pragma unroll
        for (int e = 0; e < E; ++e) inv[e] = c_make(t0[e], t1[e]);
#pragma unroll
        for (int i = 0; i < NK2D_TAB; ++i) tab[i] = c_make(tr0[i], ti0[i]);
        if (!A.first) {
            double xs[E], xn[E];
            load_col<E, MP>(A.xcr_old, cs_col, lane, xs);
            load_col<E, MP>(A.xcr_old, cn_col, lane, xn);
#pragma unroll
            for (int e = 0; e < E; ++e) rr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], rr[e]));
            load_col<E, MP>(A.xci_old, cs_col, lane, xs);
            load_col<E, MP>(A.xci_old, cn_col, lane, xn);
#pragma unroll
            for (int e = 0; e < E; ++e) ri[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], ri[e]));
            if constexpr (KIND == 1) {
                phos_couple<E>(P, tr, j, lane, A.xcr_old, upr, rr);
                phos_couple<E>(P, tr, j, lane, A.xci_old, upr, ri);
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool valid = (lane * E + e) < P.nz;
            r[e] = c_make(valid ? rr[e] : 0.0, valid ? ri[e] : 0.0);
        }
        tridiag_apply<E, cplx>(a, cc, inv, tab, r, lane);
#pragma unroll
        for (int e = 0; e < E; ++e) { rr[e] = r[e].re; ri[e] = r[e].im; }
        store_col<E, MP>(A.xcr_new, col, lane, rr);
        store_col<E, MP>(A.xci_new, col, lane, ri);
    }
}

template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK) k_sweep(DevP P, SweepArgs A) {
    GUARD_RETURN(P.guard)
    TASK_PROLOGUE(A.ntasks)
    sweep_body<E, KIND, 0>(P, A, task, lane);
}

static void fill_factor_args(const nk2d_ctx* c, SweepArgs& A) {
    A.JL = c->JL; A.JU = c->JU; A.JS = c->JS; A.JN = c->JN; A.JC = c->JC;
    A.fr_inv = c->FR_INV; A.fc_invr = c->FC_INVR; A.fc_invi = c->FC_INVI;
    A.fr_tab = c->FR_TAB; A.fc_tabr = c->FC_TABR; A.fc_tabi = c->FC_TABI;
    A.fr_inv32 = c->FR32_INV; A.fc_invr32 = c->FC32_INVR; A.fc_invi32 = c->FC32_INVI;
    A.fr_tab32 = c->FR32_TAB; A.fc_tabr32 = c->FC32_TABR; A.fc_tabi32 = c->FC32_TABI;
    A.f32 = c->factor_fp32;
}

int nk2d_k_jac_apply(nk2d_ctx* c, const double* v, double* out) {
    SweepArgs A = {};
    fill_factor_args(c, A);
    DevP P = make_devp(c);
    NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_jac_apply<EE, KK>), dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream, P, A, v, out));
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
    P.guard = nullptr;
    NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_factor<EE, KK>), dim3(nk2d_grid(A.ntasks)), dim3(NK2D_BLOCK), 0, c->stream, P, A));
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
    DevP P = make_devp(c);
    const bool sample = false;  // the profiled kernel is k_newton_fused
    if (sample) NK2D_CHECK(c, hipEventRecord(c->prof_ev[c->prof_used], c->stream));
    const int wpb = c->sweep_wpb;  // waves per block of the sweep kernel
    NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_sweep<EE, KK>), dim3((A.ntasks + wpb - 1) / wpb), dim3(64 * wpb), 0, c->stream, P, A));
    NK2D_CHECK(c, hipGetLastError());
    if (sample) {
        NK2D_CHECK(c, hipEventRecord(c->prof_ev[c->prof_used + 1], c->stream));
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
    *out = nk2d_hm_part_sum(part, ntasks, NK2D_BLOCK);
    return 0;
}

int nk2d_k_reduce(nk2d_ctx* c, int ntasks, int nout, double* host_out) {
    if (c->part_on_host && host_out && nout == 1) {
        // host-controlled integrator: no reduction launch
        NK2D_CHECK(c, hipStreamSynchronize(c->stream));
        return nk2d_part_sum(c, ntasks, host_out, nullptr);
    }
    // a result the host waits for goes straight into the pinned, device-visible host buffer: no
    // separate device-to-host copy (a blit kernel of its own on this runtime) behind the reduction
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(NK2D_BLOCK), 0, c->stream, c->PART, ntasks, nout,
                       host_out ? c->hRED : c->RED);
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    if (host_out) {
        NK2D_CHECK(c, hipStreamSynchronize(c->stream));
        std::memcpy(host_out, c->hRED, sizeof(double) * nout);
    }
    return 0;
}

// ---------------------------------------------------------------------------------
// device-side control of the simplified Newton iteration (radau.py:113-133): the final
// reduction of the ||dW / scale|| partials also takes SciPy's convergence / divergence
// decisions, so that the host can queue all NEWTON_MAXITER iterations and the error
// estimate without reading anything back; later kernels test the `done` / `skip_err`
// flags at entry and return at once.
// ---------------------------------------------------------------------------------
__device__ double block_sum(const double* __restrict__ part, int ntasks, double* sh) {
    double s = 0.0;
    for (int i = threadIdx.x; i < ntasks; i += NK2D_BLOCK) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = NK2D_BLOCK / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    return sh[0];
}

__global__ void k_ctl_reset(double* __restrict__ d, int* __restrict__ ic, double tol, double n_total) {
    if (threadIdx.x == 0) {
        d[0] = 0.0; d[1] = 0.0; d[2] = 0.0; d[3] = 0.0; d[4] = tol; d[5] = 3.0 * n_total; d[6] = n_total;
        ic[0] = 0; ic[1] = 0; ic[2] = 0; ic[3] = 0; ic[4] = 0; ic[5] = 1; ic[6] = 0;
    }
}

__global__ void k_reduce_newton(const double* __restrict__ part, int ntasks, double* __restrict__ d, int* __restrict__ ic) {
    __shared__ double sh[NK2D_BLOCK];
    if (ic[3] != 0) return;  // already decided
    const double sum = block_sum(part, ntasks, sh);
    if (threadIdx.x != 0) return;
    const int k = ic[0];
    const double tol = d[4];
    const double dW_norm = sqrt(sum) / sqrt(d[5]);
    d[2] = dW_norm;
    bool has_rate = ic[2] != 0;
    double rate = d[1];
    if (!(dW_norm == dW_norm)) { ic[3] = 1; ic[6] = k + 1; return; }  // NaN: diverged
    if (ic[1] != 0) { rate = dW_norm / d[0]; has_rate = true; d[1] = rate; ic[2] = 1; }
    if (has_rate) {
        double pw = 1.0;
        for (int i = 0; i < 6 - k; ++i) pw *= rate;  // rate ** (NEWTON_MAXITER - k)
        if (rate >= 1.0 || pw / (1.0 - rate) * dW_norm > tol) { ic[3] = 1; ic[6] = k + 1; return; }
    }
    if (dW_norm == 0.0 || (has_rate && rate / (1.0 - rate) * dW_norm < tol)) {
        ic[3] = 1; ic[4] = 1; ic[5] = 0; ic[6] = k + 1;
        return;
    }
    d[0] = dW_norm;
    ic[1] = 1;
    ic[0] = k + 1;
    if (k + 1 >= 6) { ic[3] = 1; ic[6] = 6; }
}

__global__ void k_reduce_err(const double* __restrict__ part, int ntasks, double* __restrict__ d, const int* __restrict__ ic) {
    __shared__ double sh[NK2D_BLOCK];
    if (ic[5] != 0) return;
    const double sum = block_sum(part, ntasks, sh);
    if (threadIdx.x == 0) d[3] = sum;
}

int nk2d_r_ctl_reset(nk2d_ctx* c, double newton_tol, double n_total) {
    hipLaunchKernelGGL(k_ctl_reset, dim3(1), dim3(64), 0, c->stream, c->DCTL, c->ICTL, newton_tol, n_total);
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
int nk2d_r_reduce_newton(nk2d_ctx* c) {
    hipLaunchKernelGGL(k_reduce_newton, dim3(1), dim3(NK2D_BLOCK), 0, c->stream, c->PART, c->ncol, c->DCTL, c->ICTL);
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
int nk2d_r_reduce_err(nk2d_ctx* c) {
    hipLaunchKernelGGL(k_reduce_err, dim3(1), dim3(NK2D_BLOCK), 0, c->stream, c->PART, c->ncol, c->DCTL, c->ICTL);
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
int nk2d_r_ctl_read(nk2d_ctx* c, double* dctl8, int* ictl8) {
    NK2D_CHECK(c, hipMemcpyAsync(c->hCTL, c->DCTL, 64, hipMemcpyDeviceToHost, c->stream));
    NK2D_CHECK(c, hipMemcpyAsync(c->hCTL + 8, c->ICTL, 32, hipMemcpyDeviceToHost, c->stream));
    NK2D_CHECK(c, hipStreamSynchronize(c->stream));
    std::memcpy(dctl8, c->hCTL, 64);
    std::memcpy(ictl8, c->hCTL + 8, 32);
    return 0;
}

// copy of the control block into pinned slot `slot`, in stream order, plus an event the host
// can wait on while later (speculative) work is already queued
int nk2d_r_ctl_snapshot(nk2d_ctx* c, int slot) {
    double* dst = c->hSNAP + (size_t)slot * 16;
    NK2D_CHECK(c, hipMemcpyAsync(dst, c->DCTL, 64, hipMemcpyDeviceToHost, c->stream));
    NK2D_CHECK(c, hipMemcpyAsync(dst + 8, c->ICTL, 32, hipMemcpyDeviceToHost, c->stream));
    NK2D_CHECK(c, hipEventRecord(c->snap_ev[slot], c->stream));
    return 0;
}
int nk2d_r_ctl_wait(nk2d_ctx* c, int slot, double* dctl8, int* ictl8) {
    NK2D_CHECK(c, hipEventSynchronize(c->snap_ev[slot]));
    const double* src = c->hSNAP + (size_t)slot * 16;
    std::memcpy(dctl8, src, 64);
    std::memcpy(ictl8, src + 8, 32);
    return 0;
}

// ---------------------------------------------------------------------------------
// Radau IIA elementwise kernels (scipy/integrate/_ivp/radau.py)
// ---------------------------------------------------------------------------------
// Z0 from the previous step's collocation polynomial, W = TI Z0 (radau.py:445-448,95)
struct PredictArgs {
    const double *y, *yold, *zp;
    double *z, *w;
    size_t nv;
    double x0, x1, x2;
};

template <int E, int MP = 0>
__device__ __forceinline__ void predict_body(const PredictArgs& A, int task, int lane) {
    const double* __restrict__ y = A.y;
    const double* __restrict__ yold = A.yold;
    const double* __restrict__ zp = A.zp;
    double* __restrict__ z = A.z;
    double* __restrict__ w = A.w;
    const size_t nv = A.nv;
    const double x0 = A.x0, x1 = A.x1, x2 = A.x2;
    double yy[E], yo[E], z0[E], z1[E], z2[E];
    load_col<E, MP>(y, task, lane, yy);
    load_col<E, MP>(yold, task, lane, yo);
    load_col<E, MP>(zp, task, lane, z0);
    load_col<E, MP>(zp + nv, task, lane, z1);
    load_col<E, MP>(zp + 2 * nv, task, lane, z2);
    const double xs[3] = {x0, x1, x2};
    double o[3][E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        double q[3];
#pragma unroll
        for (int cidx = 0; cidx < 3; ++cidx) q[cidx] = (z0[e] * cP[0][cidx] + z1[e] * cP[1][cidx]) + z2[e] * cP[2][cidx];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double p1 = xs[i], p2 = p1 * xs[i], p3 = p2 * xs[i];
            double v = (q[0] * p1 + q[1] * p2) + q[2] * p3;
            v = v + yo[e];
            o[i][e] = v - yy[e];
        }
    }
    double wv[E];
#pragma unroll
    for (int i = 0; i < 3; ++i) store_col<E, MP>(z + i * nv, task, lane, o[i]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int e = 0; e < E; ++e) wv[e] = (cTI[r][0] * o[0][e] + cTI[r][1] * o[1][e]) + cTI[r][2] * o[2][e];
        store_col<E>(w + r * nv, task, lane, wv);
    }
}

template <int E>
__global__ void k_predict(int ncol, PredictArgs A) {
    TASK_PROLOGUE(ncol)
    predict_body<E>(A, task, lane);
}

// start of a step attempt in one launch: the vertical mixing planes at the three stage times
// (first blocks) and the predicted stage values (remaining blocks) are independent of each other
struct JacOut {
    double *JL, *JU, *JS, *JN, *JC;
    int stage;   // >= 0: the waves computing the plane of this stage time also derive the Jacobian planes from it
};

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

// arguments of the stage part of k_newton_fused: stage tendencies F_i = fun(t + c_i h, y + Z_i)
// and transformed residuals f_real = F^T TI_REAL - M_real W0,
// f_complex = F^T TI_COMPLEX - M_complex (W1 + i W2)  (radau.py:104-111)
struct StageArgs {
    const double *y, *z, *w;
    double* zout;            // where the update writes Z = T W (z itself, or the spare buffer of a single-launch iteration)
    const double* kv[3];
    double *br, *bcr, *bci;
    size_t nv;
    double mreal, mcr, mci;
};

// ---------------------------------------------------------------------------------
// Fused simplified-Newton iteration.  One wave owns one (tracer, ypos) column and runs,
// depending on the flags, the pieces of a Newton iteration that need no data from other
// columns between them:
//   do_stage  : stage tendencies + transformed residuals (radau.py:104-111) -> right-hand sides
//   (always)  : one line-relaxation sweep of the real AND the complex system of the column
//               (first: no lateral terms)
//   do_update : W += dW, Z = T W, sum((dW/scale)^2) partial (radau.py:113-129)
// With m sweeps per solve a Newton iteration is m launches (stage fused into the first,
// update into the last) instead of m + 2.
// ---------------------------------------------------------------------------------
// The launch that ends the last Newton iteration of a FROZEN step (a replayed year knows it is the last) also ends the
// step: y_new = y + Z_2, the predicted stage values of the next attempt and W = TI Z_0 from the collocation polynomial of
// this step -- all of it the column's own data, already in the registers of the update -- and, in workgroups behind
// the column workgroups, the next attempt's mixing planes with the Jacobian planes derived from one of them.  A
// step boundary launch of its own disappears (2 600 of 12 000 launches of a 416^2 year).  The planes go to a second set
// of buffers: this launch's own stage and sweep parts still read the current ones.
struct FinalArgs {
    double* ynew;           // y + Z_2 (the buffer that becomes Y)
    double* znext;          // predicted stage values of the next attempt, 3 nv (never the Z the stage part reads)
    double x0, x1, x2;      // dense-output abscissae of the next attempt's stage times
    int nblk_cols;          // workgroups of the columns; the plane workgroups follow
};

struct FusedArgs {
    StageArgs st;
    SweepArgs sw;
    double* part;
    int do_stage, do_update;
    // two-sweep solves: the first launch keeps only x1 = T^-1 b, the second computes
    // x2 = x1 + T^-1 (lateral couplings of x1) and never needs the right-hand sides back
    int delta;
};

// STAGE = 0: instantiation for launches without the stage part.  For the phosphorus module the
// full kernel needs more registers than two waves per SIMD leave while its 3 ny columns are more
// waves than the chip has SIMDs; the stage-less instantiation fits and runs in one round.
template <int E, int KIND, int FACTOR, int STAGE, int MP = 0, int FINAL = 0>
__device__ __forceinline__ void newton_fused_body(const DevP& P, const FusedArgs& A, int task, int lane,
                                                  const FinalArgs* fin = nullptr) {
    const int tr = task / P.ny, j = task - tr * P.ny;
    const int cs_col = (j > 0) ? task - 1 : task, cn_col = (j < P.ny - 1) ? task + 1 : task;
    double fr[E], fcr[E], fci[E];
    if (STAGE && A.do_stage) {
        ColCoef<E> cf;
        load_coef<E>(P, j, lane, cf);
        double y0[E], ys[E], yn[E];
        load_col<E, MP>(A.st.y, task, lane, y0);
        load_col<E, MP>(A.st.y, cs_col, lane, ys);
        load_col<E, MP>(A.st.y, cn_col, lane, yn);
#pragma unroll
        for (int e = 0; e < E; ++e) { fr[e] = 0.0; fcr[e] = 0.0; fci[e] = 0.0; }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double c[E], cs[E], cn[E], kv[E], f[E];
            load_col<E, MP>(A.st.z + i * A.st.nv, task, lane, c);
            load_col<E, MP>(A.st.z + i * A.st.nv, cs_col, lane, cs);
            load_col<E, MP>(A.st.z + i * A.st.nv, cn_col, lane, cn);
            load_col<E, MP>(A.st.kv[i], j, lane, kv);
#pragma unroll
            for (int e = 0; e < E; ++e) { c[e] = y0[e] + c[e]; cs[e] = ys[e] + cs[e]; cn[e] = yn[e] + cn[e]; }
            tend_col<E, KIND>(P, cf, c, cs, cn, kv, tr, lane, f);
            if constexpr (KIND == 2) forced_sources<E>(P, A.st.kv[i], j, lane, c, f);
            if constexpr (KIND == 1) {
                double u1[E], u2[E], v1[E], v2[E];
                phos_load_others<E>(P, tr, j, lane, A.st.y, u1, u2);
                phos_load_others<E>(P, tr, j, lane, A.st.z + i * A.st.nv, v1, v2);
                phos_add<E>(u1, u2, v1, v2);
                phos_sources<E>(P, tr, j, lane, c, u1, u2, cf.dzr, f);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                fr[e] = fr[e] + f[e] * cTI[0][i];
                fcr[e] = fcr[e] + f[e] * cTI[1][i];
                fci[e] = fci[e] + f[e] * cTI[2][i];
            }
        }
        double w0[E], w1[E], w2[E];
        load_col<E>(A.st.w, task, lane, w0);
        load_col<E>(A.st.w + A.st.nv, task, lane, w1);
        load_col<E>(A.st.w + 2 * A.st.nv, task, lane, w2);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            fr[e] = fr[e] - A.st.mreal * w0[e];
            fcr[e] = fcr[e] - (A.st.mcr * w1[e] - A.st.mci * w2[e]);
            fci[e] = fci[e] - (A.st.mcr * w2[e] + A.st.mci * w1[e]);
        }
        if (!A.do_update && !A.delta) {  // later sweeps read the right-hand sides back
            store_col<E>(A.st.br, task, lane, fr);
            store_col<E>(A.st.bcr, task, lane, fcr);
            store_col<E>(A.st.bci, task, lane, fci);
        }
    } else if (A.delta) {
#pragma unroll
        for (int e = 0; e < E; ++e) { fr[e] = 0.0; fcr[e] = 0.0; fci[e] = 0.0; }
    } else {
        load_col<E>(A.sw.br, task, lane, fr);
        load_col<E>(A.sw.bcr, task, lane, fcr);
        load_col<E>(A.sw.bci, task, lane, fci);
    }
    double a[E], cc[E];
    {
        double jl[E], ju[E];
        load_col<E, MP>(A.sw.JL, j, lane, jl);
        load_col<E, MP>(A.sw.JU, j, lane, ju);
        line_offdiag<E, KIND>(P, tr, lane, jl, ju, a, cc);
    }
    if (!A.sw.first) {
        double js[E], jn[E], xs[E], xn[E];
        load_col<E, MP>(A.sw.JS, j, lane, js);
        load_col<E, MP>(A.sw.JN, j, lane, jn);
        load_col<E, MP>(A.sw.xr_old, cs_col, lane, xs);
        load_col<E, MP>(A.sw.xr_old, cn_col, lane, xn);
#pragma unroll
        for (int e = 0; e < E; ++e) fr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fr[e]));
        load_col<E, MP>(A.sw.xcr_old, cs_col, lane, xs);
        load_col<E, MP>(A.sw.xcr_old, cn_col, lane, xn);
#pragma unroll
        for (int e = 0; e < E; ++e) fcr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fcr[e]));
        load_col<E, MP>(A.sw.xci_old, cs_col, lane, xs);
        load_col<E, MP>(A.sw.xci_old, cn_col, lane, xn);
#pragma unroll
        for (int e = 0; e < E; ++e) fci[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fci[e]));
        if constexpr (KIND == 1) {
            double upr[E];
            load_col<E>(P.UPR, j, lane, upr);
            phos_couple<E>(P, tr, j, lane, A.sw.xr_old, upr, fr);
            phos_couple<E>(P, tr, j, lane, A.sw.xcr_old, upr, fcr);
            phos_couple<E>(P, tr, j, lane, A.sw.xci_old, upr, fci);
        }
    }
    // real system.  FACTOR: this launch is the first one after SciPy's "LU" event -- the pivots and
    // PCR tables are computed here and stored for the launches that follow (no k_factor launch)
    {
        double inv[E], tab[NK2D_TAB];
        if constexpr (FACTOR) {
            double dre[E];
            line_diag<E, KIND, MP>(P, A.sw.JC, tr, j, lane, A.sw.cre, dre);
            tridiag_factor<E, double>(a, cc, dre, inv, tab, lane);
            store_col<E>(A.sw.fr_inv, task, lane, inv);
            double* p = A.sw.fr_tab + (size_t)task * (NK2D_TAB * 64) + lane;
#pragma unroll
            for (int i = 0; i < NK2D_TAB; ++i) p[i * 64] = tab[i];
            if (A.sw.f32) {
                store_col32<E>(A.sw.fr_inv32, task, lane, inv);
                store_tab32(A.sw.fr_tab32, task, lane, tab);
            }
        } else if (A.sw.f32) {
            load_col32<E>(A.sw.fr_inv32, task, lane, inv);
            load_tab32(A.sw.fr_tab32, task, lane, tab);
        } else {
            load_col<E>(A.sw.fr_inv, task, lane, inv);
            load_tab<E>(A.sw.fr_tab, task, lane, tab);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) fr[e] = ((lane * E + e) < P.nz) ? fr[e] : 0.0;
        tridiag_apply<E, double>(a, cc, inv, tab, fr, lane);
    }
    // complex system
    {
        cplx r[E], inv[E], tab[NK2D_TAB];
        if constexpr (FACTOR) {
            double dre[E];
            cplx d[E];
            line_diag<E, KIND, MP>(P, A.sw.JC, tr, j, lane, A.sw.ccr, dre);
#pragma unroll
            for (int e = 0; e < E; ++e) d[e] = c_make(dre[e], ((lane * E + e) < P.nz) ? A.sw.cci : 0.0);
            tridiag_factor<E, cplx>(a, cc, d, inv, tab, lane);
            double re[E], im[E];
#pragma unroll
            for (int e = 0; e < E; ++e) { re[e] = inv[e].re; im[e] = inv[e].im; }
            store_col<E>(A.sw.fc_invr, task, lane, re);
            store_col<E>(A.sw.fc_invi, task, lane, im);
            double* pr = A.sw.fc_tabr + (size_t)task * (NK2D_TAB * 64) + lane;
            double* pi = A.sw.fc_tabi + (size_t)task * (NK2D_TAB * 64) + lane;
            double tre[NK2D_TAB], tim[NK2D_TAB];
#pragma unroll
            for (int i = 0; i < NK2D_TAB; ++i) { pr[i * 64] = tab[i].re; pi[i * 64] = tab[i].im; tre[i] = tab[i].re; tim[i] = tab[i].im; }
            if (A.sw.f32) {
                store_col32<E>(A.sw.fc_invr32, task, lane, re);
                store_col32<E>(A.sw.fc_invi32, task, lane, im);
                store_tab32(A.sw.fc_tabr32, task, lane, tre);
                store_tab32(A.sw.fc_tabi32, task, lane, tim);
            }
        } else {
            double t0[E], t1[E], tr0[NK2D_TAB], ti0[NK2D_TAB];
            if (A.sw.f32) {
                load_col32<E>(A.sw.fc_invr32, task, lane, t0);
                load_col32<E>(A.sw.fc_invi32, task, lane, t1);
                load_tab32(A.sw.fc_tabr32, task, lane, tr0);
                load_tab32(A.sw.fc_tabi32, task, lane, ti0);
            } else {
                load_col<E>(A.sw.fc_invr, task, lane, t0);
                load_col<E>(A.sw.fc_invi, task, lane, t1);
                load_tab<E>(A.sw.fc_tabr, task, lane, tr0);
                load_tab<E>(A.sw.fc_tabi, task, lane, ti0);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) inv[e] = c_make(t0[e], t1[e]);
#pragma unroll
            for (int i = 0; i < NK2D_TAB; ++i) tab[i] = c_make(tr0[i], ti0[i]);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool valid = (lane * E + e) < P.nz;
            r[e] = c_make(valid ? fcr[e] : 0.0, valid ? fci[e] : 0.0);
        }
        tridiag_apply<E, cplx>(a, cc, inv, tab, r, lane);
#pragma unroll
        for (int e = 0; e < E; ++e) { fcr[e] = r[e].re; fci[e] = r[e].im; }
    }
    if (!A.do_stage && A.delta) {  // correction of the first sweep's solution
        double x1[E];
        load_col<E, MP>(A.sw.xr_old, task, lane, x1);
#pragma unroll
        for (int e = 0; e < E; ++e) fr[e] = x1[e] + fr[e];
        load_col<E, MP>(A.sw.xcr_old, task, lane, x1);
#pragma unroll
        for (int e = 0; e < E; ++e) fcr[e] = x1[e] + fcr[e];
        load_col<E, MP>(A.sw.xci_old, task, lane, x1);
#pragma unroll
        for (int e = 0; e < E; ++e) fci[e] = x1[e] + fci[e];
    }
    if (!A.do_update) {
        store_col<E, MP>(A.sw.xr_new, task, lane, fr);
        store_col<E, MP>(A.sw.xcr_new, task, lane, fcr);
        store_col<E, MP>(A.sw.xci_new, task, lane, fci);
        return;
    }
    // dW = (fr, fcr, fci): norm partial, W += dW, Z = T W
    double yy[E], w0[E], w1[E], w2[E];
    load_col<E, MP>(A.st.y, task, lane, yy);
    load_col<E>(A.st.w, task, lane, w0);
    load_col<E>(A.st.w + A.st.nv, task, lane, w1);
    load_col<E>(A.st.w + 2 * A.st.nv, task, lane, w2);
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double sc = P.atol + fabs(yy[e]) * P.rtol;
        const double d0 = fr[e] / sc, d1 = fcr[e] / sc, d2 = fci[e] / sc;
        acc += (d0 * d0 + d1 * d1) + d2 * d2;
        w0[e] = w0[e] + fr[e];
        w1[e] = w1[e] + fcr[e];
        w2[e] = w2[e] + fci[e];
    }
    acc = wave_sum(acc);
    if (lane == 0) st_mp<MP>(A.part + task, acc);
    double* wout = const_cast<double*>(A.st.w);
    if constexpr (FINAL) {
        // end of a frozen step (FinalArgs): the operations of commit_tend_body (y_new) and predict_body, on registers
        double z0[E], z1[E], z2[E], yn[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            z0[e] = (cT[0][0] * w0[e] + cT[0][1] * w1[e]) + cT[0][2] * w2[e];
            z1[e] = (cT[1][0] * w0[e] + cT[1][1] * w1[e]) + cT[1][2] * w2[e];
            z2[e] = (cT[2][0] * w0[e] + cT[2][1] * w1[e]) + cT[2][2] * w2[e];
            yn[e] = yy[e] + z2[e];
        }
        store_col<E, MP>(fin->ynew, task, lane, yn);
        const double xs[3] = {fin->x0, fin->x1, fin->x2};
        double o[3][E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            double q[3];
#pragma unroll
            for (int cidx = 0; cidx < 3; ++cidx) q[cidx] = (z0[e] * cP[0][cidx] + z1[e] * cP[1][cidx]) + z2[e] * cP[2][cidx];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double p1 = xs[i], p2 = p1 * xs[i], p3 = p2 * xs[i];
                double v = (q[0] * p1 + q[1] * p2) + q[2] * p3;
                v = v + yy[e];
                o[i][e] = v - yn[e];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) store_col<E, MP>(fin->znext + i * A.st.nv, task, lane, o[i]);
        double wv[E];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int e = 0; e < E; ++e) wv[e] = (cTI[r][0] * o[0][e] + cTI[r][1] * o[1][e]) + cTI[r][2] * o[2][e];
            store_col<E>(wout + r * A.st.nv, task, lane, wv);
        }
        return;
    }
    double* zout = A.st.zout;
    store_col<E>(wout, task, lane, w0);
    store_col<E>(wout + A.st.nv, task, lane, w1);
    store_col<E>(wout + 2 * A.st.nv, task, lane, w2);
    double zz[E];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int e = 0; e < E; ++e) zz[e] = (cT[r][0] * w0[e] + cT[r][1] * w1[e]) + cT[r][2] * w2[e];
        store_col<E, MP>(zout + r * A.st.nv, task, lane, zz);
    }
}

template <int E, int KIND, int FACTOR, int STAGE>
__global__ void __launch_bounds__(NK2D_BLOCK) k_newton_fused(DevP P, FusedArgs A) {
    GUARD_RETURN(P.guard)
    TASK_PROLOGUE_XCD(P.ncol)
    newton_fused_body<E, KIND, FACTOR, STAGE, 0>(P, A, task, lane);
}

// Work of the NEXT step that depends on time alone, hidden behind the column waves of this step's launches (a frozen
// year knows every step ahead; the launches without the factorisation hold 232 registers, so a second wave fits on
// every SIMD beside the 832 column waves of a 416^2 launch):
//   * its mixing planes at the three stage times and the Jacobian planes derived from one of them ride on the first
//     launch of this step that does not factorise (k_newton_fused_pl; the exp of a mixing column is a 5 us chain);
//   * its line factorisation -- pivots and PCR tables of both systems of every column, from those Jacobian planes -- on
//     the launch that ends this step (PreFactor tasks of k_newton_final, the work of k_factor), into the second set of
//     factor buffers.
// The next step then opens with the launch that LOADS its factorisation instead of the factorising instantiation (303
// registers, one wave per SIMD, 22.9 us instead of 16.5 us at 416^2).  Where the planes could not ride ahead (a step of
// one launch) the final launch computes them as before and the next step factorises for itself.
struct PreFactor {
    int mode;        // 0: plane tasks behind the columns (the round-2 launch); 1: nothing; 2: factor tasks (planes done earlier)
    SweepArgs sa;    // Jacobian planes of the next step, its shifts, the second set of factor buffers
};

// one (stage time, ypos column) task of the next attempt's planes; the plane the Jacobian derives from comes first
template <int E>
__device__ __forceinline__ void plane_task(const DevP& P, const VmixArgs& V, const JacOut& J, int ptask, int lane) {
    int ti = ptask / P.ny;
    const int j = ptask - ti * P.ny;
    if (J.stage > 0) ti = (ti == 0) ? J.stage : ((ti <= J.stage) ? ti - 1 : ti);
    double kv[E];
    vmix_body_kv<E>(P, V, ti * P.ny + j, lane, kv);
    if (ti == J.stage) jac_core<E, 0>(P, kv, V.out[ti], J.JL, J.JU, J.JS, J.JN, J.JC, nullptr, nullptr, j, lane);
}

// a Newton-iteration launch without the factorisation, with the plane tasks of the next attempt behind its columns
template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK) k_newton_fused_pl(DevP P, FusedArgs A, int nblk_cols, VmixArgs V, JacOut J) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    if ((int)blockIdx.x < nblk_cols) {
        const int task = blockIdx.x * wpb + wave;
        if (task < P.ncol) newton_fused_body<E, KIND, 0, 1, 0>(P, A, task, lane);
        return;
    }
    const int ptask = (blockIdx.x - nblk_cols) * wpb + wave;
    if (ptask < P.ny * 3) plane_task<E>(P, V, J, ptask, lane);
}

// the launch that ends a frozen step (FinalArgs): column workgroups first, then the workgroups of the next attempt's
// planes or of its line factorisation (PreFactor)
template <int E, int KIND, int FACTOR, int STAGE>
__global__ void __launch_bounds__(NK2D_BLOCK) k_newton_final(DevP P, FusedArgs A, FinalArgs Fin, VmixArgs V, JacOut J, PreFactor F) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    if ((int)blockIdx.x < Fin.nblk_cols) {
        const int task = blockIdx.x * wpb + wave;
        if (task < P.ncol) newton_fused_body<E, KIND, FACTOR, STAGE, 0, 1>(P, A, task, lane, &Fin);
        return;
    }
    const int ptask = (blockIdx.x - Fin.nblk_cols) * wpb + wave;
    if (F.mode == 0) {
        if (ptask < P.ny * 3) plane_task<E>(P, V, J, ptask, lane);
    } else if (F.mode == 2) {
        if (ptask < F.sa.ntasks) factor_body<E, KIND>(P, F.sa, ptask, lane);
    }
}

// ---------------------------------------------------------------------------------
// Column team: the same launch as k_newton_fused -- same arguments, bit-identical results -- with one
// WORKGROUP of four waves per column instead of one wave.  At 416 x 416 k_newton_fused is 416 waves on a
// chip with 1024 SIMDs, each walking through some forty dependent column loads and five arithmetic
// phases: it is bound by the latency of that chain, not by bytes.  The team cuts the chain:
//   phase 1   waves 0..2: stage tendency F_i of stage i = wave (a third of the stage loads each);
//             wave 3 fetches both line factorisations and W meanwhile: the complex one stays in its
//             registers, the real one and W go to LDS for wave 0 and the update
//   phase 2   wave 0: real right-hand side + real line solve; wave 3: the complex ones
//   phase 3   waves 0..2: W_r += dW_r, Z_r = (T W)_r and the squared scaled increments of component r;
//             wave 3 adds them up in k_newton_fused's order
// F_i, dW and the squares travel through LDS.  The two roles live in disjoint branches on the (scalar)
// wave index, so each is register-allocated on its own: one wave per column needs > 256 VGPRs at seven
// levels per lane, a team wave fits 256 and two workgroups share a CU.  Every wave passes the same number
// of barriers on either branch.  The arithmetic of every value is the one of newton_fused_body, operation
// for operation, so either kernel can run any launch of a year.
// ---------------------------------------------------------------------------------
template <int E, int WR>      // WR: rows of W kept in LDS (3 for four-wave teams, 0 for pairs: 40 KB, four workgroups per CU)
struct TeamLds {
    double F[3][E * 64];   // stage tendencies; later the squared scaled increments
    double D[3][E * 64];   // dW of the real system, real and imaginary part of the complex one
    double W[WR > 0 ? WR : 1][WR > 0 ? E * 64 : 1];   // W before the update (stage launches of four-wave teams)
    double a[E * 64], c[E * 64], inv[E * 64];   // real system: off-diagonals, pivot reciprocals (FACTOR: the diagonal)
    double tab[NK2D_TAB * 64];
};

template <int E>
__device__ __forceinline__ void lds_put(double* s, int lane, const double (&v)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) s[e * 64 + lane] = v[e];
}
template <int E>
__device__ __forceinline__ void lds_get(const double* s, int lane, double (&v)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = s[e * 64 + lane];
}

// NW = 4: waves 0..2 take a stage each, wave 3 the complex system.  NW = 2 (a "pair"): wave 0 takes the three stages
// one after the other, wave 1 the complex system -- for modules with more columns than four-wave teams fit the chip
// at once (iage 416^2: 832 columns = 1 664 pair waves of <= 256 VGPRs, one round).  FIN (pairs only): the launch also
// ends a frozen step (FinalArgs; the plane workgroups follow the nblk_cols column workgroups, as in k_newton_final).
// MP: the accessors of a persistent kernel (1: write-through stores, L1-bypassing loads; 2: plain stores, L1-bypassing loads)
// for everything that another wave reads in a later phase.
template <int E, int KIND, int FACTOR, int STAGE, int NW, int FIN, int MP = 0>
__device__ __forceinline__ void newton_team_body(const DevP& P, const FusedArgs& A, TeamLds<E, (NW == 4 ? 3 : 0)>& S, int task, int w, int lane,
                                                 const FinalArgs* fin) {
    constexpr int CW = NW - 1;      // the complex wave
    constexpr int NS = NW - 1;      // stage waves
    const int tr = task / P.ny, j = task - tr * P.ny;
    const int cs_col = (j > 0) ? task - 1 : task, cn_col = (j < P.ny - 1) ? task + 1 : task;
    const bool stage = STAGE && A.do_stage;

    if (w == CW) {
        // =========================== complex system; supplier of the real one ===========================
        double a[E], cc[E];
        {
            double jl[E], ju[E];
            load_col<E>(A.sw.JL, j, lane, jl);
            load_col<E>(A.sw.JU, j, lane, ju);
            line_offdiag<E, KIND>(P, tr, lane, jl, ju, a, cc);
        }
        cplx cinv[E], ctab[NK2D_TAB];
        double dre[E];   // FACTOR: real part of the complex diagonal
        if (stage) {
            // what wave 0 and the update need, through LDS
            double rinv[E], rtab[NK2D_TAB], w0[E];
            if constexpr (FACTOR) {
                double drr[E];
                line_diag<E, KIND, 0>(P, A.sw.JC, tr, j, lane, A.sw.cre, drr);
#pragma unroll
                for (int e = 0; e < E; ++e) rinv[e] = drr[e];
            } else if (A.sw.f32) {
                load_col32<E>(A.sw.fr_inv32, task, lane, rinv);
                load_tab32(A.sw.fr_tab32, task, lane, rtab);
            } else {
                load_col<E>(A.sw.fr_inv, task, lane, rinv);
                load_tab<E>(A.sw.fr_tab, task, lane, rtab);
            }
            if constexpr (NW == 4) load_col<E, MP>(A.st.w, task, lane, w0);
            lds_put<E>(S.a, lane, a);
            lds_put<E>(S.c, lane, cc);
            lds_put<E>(S.inv, lane, rinv);
            if constexpr (!FACTOR) {
#pragma unroll
                for (int i = 0; i < NK2D_TAB; ++i) S.tab[i * 64 + lane] = rtab[i];
            }
            if constexpr (NW == 4) lds_put<E>(S.W[0], lane, w0);
        }
        if constexpr (FACTOR) {
            line_diag<E, KIND, 0>(P, A.sw.JC, tr, j, lane, A.sw.ccr, dre);
        } else {
            double t0[E], t1[E], tr0[NK2D_TAB], ti0[NK2D_TAB];
            if (A.sw.f32) {
                load_col32<E>(A.sw.fc_invr32, task, lane, t0);
                load_col32<E>(A.sw.fc_invi32, task, lane, t1);
                load_tab32(A.sw.fc_tabr32, task, lane, tr0);
                load_tab32(A.sw.fc_tabi32, task, lane, ti0);
            } else {
                load_col<E>(A.sw.fc_invr, task, lane, t0);
                load_col<E>(A.sw.fc_invi, task, lane, t1);
                load_tab<E>(A.sw.fc_tabr, task, lane, tr0);
                load_tab<E>(A.sw.fc_tabi, task, lane, ti0);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) cinv[e] = c_make(t0[e], t1[e]);
#pragma unroll
            for (int i = 0; i < NK2D_TAB; ++i) ctab[i] = c_make(tr0[i], ti0[i]);
        }
        double fcr[E], fci[E];
        if (stage) {
            double w1[E], w2[E];
            load_col<E, MP>(A.st.w + A.st.nv, task, lane, w1);
            load_col<E, MP>(A.st.w + 2 * A.st.nv, task, lane, w2);
            if constexpr (NW == 4) {
                lds_put<E>(S.W[1], lane, w1);
                lds_put<E>(S.W[2], lane, w2);
            }
            __syncthreads();   // barrier 1: stage tendencies are in LDS
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double f0 = S.F[0][e * 64 + lane], f1 = S.F[1][e * 64 + lane], f2 = S.F[2][e * 64 + lane];
                double sr = 0.0, si = 0.0;
                sr = sr + f0 * cTI[1][0];
                si = si + f0 * cTI[2][0];
                sr = sr + f1 * cTI[1][1];
                si = si + f1 * cTI[2][1];
                sr = sr + f2 * cTI[1][2];
                si = si + f2 * cTI[2][2];
                fcr[e] = sr - (A.st.mcr * w1[e] - A.st.mci * w2[e]);
                fci[e] = si - (A.st.mcr * w2[e] + A.st.mci * w1[e]);
            }
            if (!A.do_update && !A.delta) {  // later sweeps read the right-hand sides back
                store_col<E, MP>(A.st.bcr, task, lane, fcr);
                store_col<E, MP>(A.st.bci, task, lane, fci);
            }
        } else if (A.delta) {
#pragma unroll
            for (int e = 0; e < E; ++e) { fcr[e] = 0.0; fci[e] = 0.0; }
        } else {
            load_col<E, MP>(A.sw.bcr, task, lane, fcr);
            load_col<E, MP>(A.sw.bci, task, lane, fci);
        }
        if (!A.sw.first) {
            double js[E], jn[E], xs[E], xn[E];
            load_col<E>(A.sw.JS, j, lane, js);
            load_col<E>(A.sw.JN, j, lane, jn);
            load_col<E, MP>(A.sw.xcr_old, cs_col, lane, xs);
            load_col<E, MP>(A.sw.xcr_old, cn_col, lane, xn);
#pragma unroll
            for (int e = 0; e < E; ++e) fcr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fcr[e]));
            load_col<E, MP>(A.sw.xci_old, cs_col, lane, xs);
            load_col<E, MP>(A.sw.xci_old, cn_col, lane, xn);
#pragma unroll
            for (int e = 0; e < E; ++e) fci[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fci[e]));
            if constexpr (KIND == 1) {
                double upr[E];
                load_col<E>(P.UPR, j, lane, upr);
                phos_couple<E>(P, tr, j, lane, A.sw.xcr_old, upr, fcr);
                phos_couple<E>(P, tr, j, lane, A.sw.xci_old, upr, fci);
            }
        }
        if constexpr (FACTOR) {
            cplx d[E];
#pragma unroll
            for (int e = 0; e < E; ++e) d[e] = c_make(dre[e], ((lane * E + e) < P.nz) ? A.sw.cci : 0.0);
            tridiag_factor<E, cplx>(a, cc, d, cinv, ctab, lane);
            double re[E], im[E];
#pragma unroll
            for (int e = 0; e < E; ++e) { re[e] = cinv[e].re; im[e] = cinv[e].im; }
            store_col<E>(A.sw.fc_invr, task, lane, re);
            store_col<E>(A.sw.fc_invi, task, lane, im);
            double* pr = A.sw.fc_tabr + (size_t)task * (NK2D_TAB * 64) + lane;
            double* pi = A.sw.fc_tabi + (size_t)task * (NK2D_TAB * 64) + lane;
            double tre[NK2D_TAB], tim[NK2D_TAB];
#pragma unroll
            for (int i = 0; i < NK2D_TAB; ++i) { pr[i * 64] = ctab[i].re; pi[i * 64] = ctab[i].im; tre[i] = ctab[i].re; tim[i] = ctab[i].im; }
            if (A.sw.f32) {
                store_col32<E>(A.sw.fc_invr32, task, lane, re);
                store_col32<E>(A.sw.fc_invi32, task, lane, im);
                store_tab32(A.sw.fc_tabr32, task, lane, tre);
                store_tab32(A.sw.fc_tabi32, task, lane, tim);
            }
        }
        cplx r[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool valid = (lane * E + e) < P.nz;
            r[e] = c_make(valid ? fcr[e] : 0.0, valid ? fci[e] : 0.0);
        }
        tridiag_apply<E, cplx>(a, cc, cinv, ctab, r, lane);
#pragma unroll
        for (int e = 0; e < E; ++e) { fcr[e] = r[e].re; fci[e] = r[e].im; }
        if (!A.do_stage && A.delta) {  // correction of the first sweep's solution
            double x1[E];
            load_col<E, MP>(A.sw.xcr_old, task, lane, x1);
#pragma unroll
            for (int e = 0; e < E; ++e) fcr[e] = x1[e] + fcr[e];
            load_col<E, MP>(A.sw.xci_old, task, lane, x1);
#pragma unroll
            for (int e = 0; e < E; ++e) fci[e] = x1[e] + fci[e];
        }
        if (!A.do_update) {
            store_col<E, MP>(A.sw.xcr_new, task, lane, fcr);
            store_col<E, MP>(A.sw.xci_new, task, lane, fci);
            return;
        }
        lds_put<E>(S.D[1], lane, fcr);
        lds_put<E>(S.D[2], lane, fci);
        __syncthreads();   // barrier 2: dW complete
        __syncthreads();   // barrier 3: squared scaled increments complete
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) acc += (S.F[0][e * 64 + lane] + S.F[1][e * 64 + lane]) + S.F[2][e * 64 + lane];
        acc = wave_sum(acc);
        if (lane == 0) st_mp<MP>(A.part + task, acc);
        return;
    }

    // =========================== stage waves: stages, real system (wave 0), update ===========================
    double yy[E];
    double wpre[E];      // pairs: W_0 of the real right-hand side, fetched before the stages
    if constexpr (NW == 2) {
        if (stage) load_col<E, MP>(A.st.w, task, lane, wpre);
    }
    if (STAGE && A.do_stage) {
        ColCoef<E> cf;
        load_coef<E>(P, j, lane, cf);
        double ys[E], yn[E];
        load_col<E, MP>(A.st.y, task, lane, yy);
        load_col<E, MP>(A.st.y, cs_col, lane, ys);
        load_col<E, MP>(A.st.y, cn_col, lane, yn);
        for (int i = w; i < 3; i += NS) {
            const double* __restrict__ zi = A.st.z + (size_t)i * A.st.nv;
            const double* __restrict__ kvi = (i == 0) ? A.st.kv[0] : ((i == 1) ? A.st.kv[1] : A.st.kv[2]);
            double c[E], cs[E], cn[E], kv[E], f[E];
            load_col<E, MP>(zi, task, lane, c);
            load_col<E, MP>(zi, cs_col, lane, cs);
            load_col<E, MP>(zi, cn_col, lane, cn);
            load_col<E>(kvi, j, lane, kv);
#pragma unroll
            for (int e = 0; e < E; ++e) { c[e] = yy[e] + c[e]; cs[e] = ys[e] + cs[e]; cn[e] = yn[e] + cn[e]; }
            tend_col<E, KIND>(P, cf, c, cs, cn, kv, tr, lane, f);
            if constexpr (KIND == 2) forced_sources<E>(P, kvi, j, lane, c, f);
            if constexpr (KIND == 1) {
                double u1[E], u2[E], v1[E], v2[E];
                phos_load_others<E>(P, tr, j, lane, A.st.y, u1, u2);
                phos_load_others<E>(P, tr, j, lane, zi, v1, v2);
                phos_add<E>(u1, u2, v1, v2);
                phos_sources<E>(P, tr, j, lane, c, u1, u2, cf.dzr, f);
            }
            lds_put<E>(S.F[i], lane, f);
        }
        __syncthreads();   // barrier 1
    }
    double w0[E], w1[E], w2[E];
    if (A.do_update) {
        if constexpr (NW == 4) {
            if (stage) {
                lds_get<E>(S.W[0], lane, w0);
                lds_get<E>(S.W[1], lane, w1);
                lds_get<E>(S.W[2], lane, w2);
            }
        }
        if (NW == 2 || !stage) {
            if (!stage) load_col<E, MP>(A.st.y, task, lane, yy);
            load_col<E, MP>(A.st.w, task, lane, w0);
            load_col<E, MP>(A.st.w + A.st.nv, task, lane, w1);
            load_col<E, MP>(A.st.w + 2 * A.st.nv, task, lane, w2);
        }
    }
    if (w == 0) {
        double a[E], cc[E], rinv[E], rtab[NK2D_TAB], fr[E];
        if (stage) {
            lds_get<E>(S.a, lane, a);
            lds_get<E>(S.c, lane, cc);
            lds_get<E>(S.inv, lane, rinv);
            if constexpr (!FACTOR) {
#pragma unroll
                for (int i = 0; i < NK2D_TAB; ++i) rtab[i] = S.tab[i * 64 + lane];
            }
            double wr0[E];
            if constexpr (NW == 4) {
                lds_get<E>(S.W[0], lane, wr0);
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) wr0[e] = wpre[e];
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                double s = 0.0;
                s = s + S.F[0][e * 64 + lane] * cTI[0][0];
                s = s + S.F[1][e * 64 + lane] * cTI[0][1];
                s = s + S.F[2][e * 64 + lane] * cTI[0][2];
                fr[e] = s - A.st.mreal * wr0[e];
            }
            if (!A.do_update && !A.delta) store_col<E, MP>(A.st.br, task, lane, fr);
        } else {
            double jl[E], ju[E];
            load_col<E>(A.sw.JL, j, lane, jl);
            load_col<E>(A.sw.JU, j, lane, ju);
            line_offdiag<E, KIND>(P, tr, lane, jl, ju, a, cc);
            if (A.sw.f32) {
                load_col32<E>(A.sw.fr_inv32, task, lane, rinv);
                load_tab32(A.sw.fr_tab32, task, lane, rtab);
            } else {
                load_col<E>(A.sw.fr_inv, task, lane, rinv);
                load_tab<E>(A.sw.fr_tab, task, lane, rtab);
            }
            if (A.delta) {
#pragma unroll
                for (int e = 0; e < E; ++e) fr[e] = 0.0;
            } else {
                load_col<E, MP>(A.sw.br, task, lane, fr);
            }
        }
        if (!A.sw.first) {
            double js[E], jn[E], xs[E], xn[E];
            load_col<E>(A.sw.JS, j, lane, js);
            load_col<E>(A.sw.JN, j, lane, jn);
            load_col<E, MP>(A.sw.xr_old, cs_col, lane, xs);
            load_col<E, MP>(A.sw.xr_old, cn_col, lane, xn);
#pragma unroll
            for (int e = 0; e < E; ++e) fr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fr[e]));
            if constexpr (KIND == 1) {
                double upr[E];
                load_col<E>(P.UPR, j, lane, upr);
                phos_couple<E>(P, tr, j, lane, A.sw.xr_old, upr, fr);
            }
        }
        if constexpr (FACTOR) {   // S.inv holds the diagonal
            double dre[E];
#pragma unroll
            for (int e = 0; e < E; ++e) dre[e] = rinv[e];
            tridiag_factor<E, double>(a, cc, dre, rinv, rtab, lane);
            store_col<E>(A.sw.fr_inv, task, lane, rinv);
            double* p = A.sw.fr_tab + (size_t)task * (NK2D_TAB * 64) + lane;
#pragma unroll
            for (int i = 0; i < NK2D_TAB; ++i) p[i * 64] = rtab[i];
            if (A.sw.f32) {
                store_col32<E>(A.sw.fr_inv32, task, lane, rinv);
                store_tab32(A.sw.fr_tab32, task, lane, rtab);
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) fr[e] = ((lane * E + e) < P.nz) ? fr[e] : 0.0;
        tridiag_apply<E, double>(a, cc, rinv, rtab, fr, lane);
        if (!A.do_stage && A.delta) {
            double x1[E];
            load_col<E, MP>(A.sw.xr_old, task, lane, x1);
#pragma unroll
            for (int e = 0; e < E; ++e) fr[e] = x1[e] + fr[e];
        }
        if (!A.do_update) store_col<E, MP>(A.sw.xr_new, task, lane, fr);
        else lds_put<E>(S.D[0], lane, fr);
    }
    if (!A.do_update) return;
    __syncthreads();   // barrier 2
    {
        double d0[E], d1[E], d2[E];
        lds_get<E>(S.D[0], lane, d0);
        lds_get<E>(S.D[1], lane, d1);
        lds_get<E>(S.D[2], lane, d2);
        // squared scaled increments of this wave's components (before the update below changes nothing they read)
        for (int r = w; r < 3; r += NS) {
            double q[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double sc = P.atol + fabs(yy[e]) * P.rtol;
                const double dr = (r == 0) ? d0[e] : ((r == 1) ? d1[e] : d2[e]);
                const double dq = dr / sc;
                q[e] = dq * dq;
            }
            lds_put<E>(S.F[r], lane, q);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            w0[e] = w0[e] + d0[e];
            w1[e] = w1[e] + d1[e];
            w2[e] = w2[e] + d2[e];
        }
        if constexpr (FIN) {
            // end of a frozen step, as in newton_fused_body<..., FINAL>: commit, prediction of the next attempt
            double z0[E], z1[E], z2[E], yn[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                z0[e] = (cT[0][0] * w0[e] + cT[0][1] * w1[e]) + cT[0][2] * w2[e];
                z1[e] = (cT[1][0] * w0[e] + cT[1][1] * w1[e]) + cT[1][2] * w2[e];
                z2[e] = (cT[2][0] * w0[e] + cT[2][1] * w1[e]) + cT[2][2] * w2[e];
                yn[e] = yy[e] + z2[e];
            }
            if (w == 0) store_col<E, MP>(fin->ynew, task, lane, yn);     // (a four-wave team: every stage wave holds all of this;
                                                                         //  wave r stores row r)
            const double xs[3] = {fin->x0, fin->x1, fin->x2};
            double o[3][E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                double qq[3];
#pragma unroll
                for (int cidx = 0; cidx < 3; ++cidx) qq[cidx] = (z0[e] * cP[0][cidx] + z1[e] * cP[1][cidx]) + z2[e] * cP[2][cidx];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double p1 = xs[i], p2 = p1 * xs[i], p3 = p2 * xs[i];
                    double v = (qq[0] * p1 + qq[1] * p2) + qq[2] * p3;
                    v = v + yy[e];
                    o[i][e] = v - yn[e];
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i % NS == w) store_col<E, MP>(fin->znext + i * A.st.nv, task, lane, o[i]);
            double wv[E];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (r % NS != w) continue;
#pragma unroll
                for (int e = 0; e < E; ++e) wv[e] = (cTI[r][0] * o[0][e] + cTI[r][1] * o[1][e]) + cTI[r][2] * o[2][e];
                store_col<E, MP>(const_cast<double*>(A.st.w) + (size_t)r * A.st.nv, task, lane, wv);
            }
        } else {
            for (int r = w; r < 3; r += NS) {
                double zz[E], wr[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    wr[e] = (r == 0) ? w0[e] : ((r == 1) ? w1[e] : w2[e]);
                    zz[e] = (cT[r][0] * w0[e] + cT[r][1] * w1[e]) + cT[r][2] * w2[e];
                }
                store_col<E, MP>(const_cast<double*>(A.st.w) + (size_t)r * A.st.nv, task, lane, wr);
                store_col<E, MP>(A.st.zout + (size_t)r * A.st.nv, task, lane, zz);
            }
        }
    }
    __syncthreads();   // barrier 3
}

template <int E, int KIND, int FACTOR, int STAGE>
__global__ void __launch_bounds__(NK2D_BLOCK, 2) k_newton_team(DevP P, FusedArgs A) {
    GUARD_RETURN(P.guard)
    __shared__ TeamLds<E, 3> S;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int task = P.xcd ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;   // see TASK_PROLOGUE_XCD
    if (task >= P.ncol) return;
    newton_team_body<E, KIND, FACTOR, STAGE, 4, 0>(P, A, S, task, w, lane, nullptr);
}

// pairs: two waves per column (newton_team_body, NW = 2); FIN: the launch also ends a frozen step (plane workgroups of
// 128 threads behind the Fin.nblk_cols column workgroups)
template <int E, int KIND, int FACTOR, int STAGE, int FIN>
__global__ void __launch_bounds__(128, 2) k_newton_pair(DevP P, FusedArgs A, FinalArgs Fin, VmixArgs V, JacOut J) {
    GUARD_RETURN(P.guard)
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

// error estimate right-hand side  f + Z^T E / h   (radau.py:478-479)
template <int E, int MP = 0>
__device__ __forceinline__ void err_rhs_body(const double* __restrict__ f, const double* __restrict__ z, size_t nv,
                                             double h, double* __restrict__ out, int task, int lane) {
    double ff[E], z0[E], z1[E], z2[E];
    load_col<E>(f, task, lane, ff);
    load_col<E, MP>(z, task, lane, z0);
    load_col<E, MP>(z + nv, task, lane, z1);
    load_col<E, MP>(z + 2 * nv, task, lane, z2);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double ze = ((z0[e] * cE[0] + z1[e] * cE[1]) + z2[e] * cE[2]) / h;
        ff[e] = ff[e] + ze;
    }
    store_col<E>(out, task, lane, ff);
}

template <int E>
__global__ void k_err_rhs(int ncol, const double* __restrict__ f, const double* __restrict__ z, size_t nv, double h,
                          double* __restrict__ out, const int* __restrict__ guard) {
    GUARD_RETURN(guard)
    TASK_PROLOGUE(ncol)
    err_rhs_body<E, 0>(f, z, nv, h, out, task, lane);
}

// ---------------------------------------------------------------------------------
// Fused error estimate (radau.py:477-481) for solves of at most two sweeps:
//   launch 0: right-hand side f + Z^T E / h formed in registers, first line sweep (no lateral
//             terms) -> x1
//   launch 1: x2 = x1 + T^-1 (lateral couplings of x1)                    (two-sweep solves only)
//   last    : sum((x / (atol + max(|y|, |y + Z2|) rtol))^2) partial, x stored for the filter pass
// Two launches per step instead of four (right-hand side, two sweeps, norm).
// ---------------------------------------------------------------------------------
struct ErrArgs {
    SweepArgs sw;          // real system: planes, factor, ping-pong iterates
    const double *f, *z, *y;
    size_t nv;
    double h;
    double* part;
    int stage;             // 0: first launch, 1: second
    int last;              // this launch ends the solve
};

template <int E, int KIND, int MP = 0>
__device__ __forceinline__ void err_fused_body(const DevP& P, const ErrArgs& A, int task, int lane) {
    const int tr = task / P.ny, j = task - tr * P.ny;
    double r[E], x1[E], a[E], cc[E];
    {
        double jl[E], ju[E];
        load_col<E, MP>(A.sw.JL, j, lane, jl);
        load_col<E, MP>(A.sw.JU, j, lane, ju);
        line_offdiag<E, KIND>(P, tr, lane, jl, ju, a, cc);
    }
    if (A.stage == 0) {
        double z0[E], z1[E], z2[E];
        load_col<E>(A.f, task, lane, r);
        load_col<E, MP>(A.z, task, lane, z0);
        load_col<E, MP>(A.z + A.nv, task, lane, z1);
        load_col<E, MP>(A.z + 2 * A.nv, task, lane, z2);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const double ze = ((z0[e] * cE[0] + z1[e] * cE[1]) + z2[e] * cE[2]) / A.h;
            r[e] = r[e] + ze;
        }
    } else {
        const int cs_col = (j > 0) ? task - 1 : task, cn_col = (j < P.ny - 1) ? task + 1 : task;
        double js[E], jn[E], xs[E], xn[E];
        load_col<E, MP>(A.sw.JS, j, lane, js);
        load_col<E, MP>(A.sw.JN, j, lane, jn);
        load_col<E, MP>(A.sw.xr_old, cs_col, lane, xs);
        load_col<E, MP>(A.sw.xr_old, cn_col, lane, xn);
        load_col<E, MP>(A.sw.xr_old, task, lane, x1);
#pragma unroll
        for (int e = 0; e < E; ++e) r[e] = __builtin_fma(jn[e], xn[e], js[e] * xs[e]);
        if constexpr (KIND == 1) {
            double upr[E];
            load_col<E>(P.UPR, j, lane, upr);
            phos_couple<E>(P, tr, j, lane, A.sw.xr_old, upr, r);
        }
    }
    {
        double inv[E], tab[NK2D_TAB];
        load_col<E>(A.sw.fr_inv, task, lane, inv);
        load_tab<E>(A.sw.fr_tab, task, lane, tab);
#pragma unroll
        for (int e = 0; e < E; ++e) r[e] = ((lane * E + e) < P.nz) ? r[e] : 0.0;
        tridiag_apply<E, double>(a, cc, inv, tab, r, lane);
    }
    if (A.stage == 1) {
#pragma unroll
        for (int e = 0; e < E; ++e) r[e] = x1[e] + r[e];
    }
    store_col<E, MP>(A.sw.xr_new, task, lane, r);
    if (!A.last) return;
    double yy[E], z2[E];
    load_col<E, MP>(A.y, task, lane, yy);
    load_col<E, MP>(A.z + 2 * A.nv, task, lane, z2);
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double yn = yy[e] + z2[e];
        const double sc = P.atol + fmax(fabs(yy[e]), fabs(yn)) * P.rtol;
        const double q = r[e] / sc;
        acc += q * q;
    }
    acc = wave_sum(acc);
    if (lane == 0) st_mp<MP>(A.part + task, acc);
}

template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK) k_err_fused(DevP P, ErrArgs A) {
    GUARD_RETURN(P.guard)
    TASK_PROLOGUE(P.ncol)
    err_fused_body<E, KIND, 0>(P, A, task, lane);
}

// filtered error estimate right-hand side  fun(t, y + error) + Z^T E / h  (radau.py:485-487)
template <int E, int KIND, int MP = 0>
__device__ __forceinline__ void err_rhs2_body(const DevP& P, const double* __restrict__ y, const double* __restrict__ err,
                                              const double* __restrict__ kvp, const double* __restrict__ z, size_t nv,
                                              double h, double* __restrict__ out, int task, int lane) {
    const int tr = task / P.ny, j = task - tr * P.ny;
    const int cs_col = (j > 0) ? task - 1 : task, cn_col = (j < P.ny - 1) ? task + 1 : task;
    ColCoef<E> cf;
    load_coef<E>(P, j, lane, cf);
    double c[E], cs[E], cn[E], t0[E], kv[E], ff[E];
    load_col<E, MP>(y, task, lane, c);
    load_col<E, MP>(err, task, lane, t0);
#pragma unroll
    for (int e = 0; e < E; ++e) c[e] = c[e] + t0[e];
    load_col<E, MP>(y, cs_col, lane, cs);
    load_col<E, MP>(err, cs_col, lane, t0);
#pragma unroll
    for (int e = 0; e < E; ++e) cs[e] = cs[e] + t0[e];
    load_col<E, MP>(y, cn_col, lane, cn);
    load_col<E, MP>(err, cn_col, lane, t0);
#pragma unroll
    for (int e = 0; e < E; ++e) cn[e] = cn[e] + t0[e];
    load_col<E, MP>(kvp, j, lane, kv);
    tend_col<E, KIND>(P, cf, c, cs, cn, kv, tr, lane, ff);
    if constexpr (KIND == 2) forced_sources<E>(P, kvp, j, lane, c, ff);
    if constexpr (KIND == 1) {
        double u1[E], u2[E], v1[E], v2[E];
        phos_load_others<E>(P, tr, j, lane, y, u1, u2);
        phos_load_others<E>(P, tr, j, lane, err, v1, v2);
        phos_add<E>(u1, u2, v1, v2);
        phos_sources<E>(P, tr, j, lane, c, u1, u2, cf.dzr, ff);
    }
    double z0[E], z1[E], z2[E];
    load_col<E, MP>(z, task, lane, z0);
    load_col<E, MP>(z + nv, task, lane, z1);
    load_col<E, MP>(z + 2 * nv, task, lane, z2);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double ze = ((z0[e] * cE[0] + z1[e] * cE[1]) + z2[e] * cE[2]) / h;
        ff[e] = ff[e] + ze;
    }
    store_col<E>(out, task, lane, ff);
}

template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK)
    k_err_rhs2(DevP P, const double* __restrict__ y, const double* __restrict__ err, const double* __restrict__ kvp,
               const double* __restrict__ z, size_t nv, double h, double* __restrict__ out) {
    TASK_PROLOGUE(P.ncol)
    err_rhs2_body<E, KIND, 0>(P, y, err, kvp, z, nv, h, out, task, lane);
}

// accepted step: y_new = y + Z2 and f_new = fun(t_new, y_new) in one pass (radau.py:509-521); the
// lateral neighbours' y_new are formed on the fly, each wave stores its own column
template <int E, int KIND, int MP = 0>
__device__ __forceinline__ void commit_tend_body(const DevP& P, const double* __restrict__ y, const double* __restrict__ z2,
                                                 const double* __restrict__ kvp, double* __restrict__ ynew,
                                                 double* __restrict__ f, int task, int lane) {
    const int tr = task / P.ny, j = task - tr * P.ny;
    const int cs_col = (j > 0) ? task - 1 : task, cn_col = (j < P.ny - 1) ? task + 1 : task;
    ColCoef<E> cf;
    load_coef<E>(P, j, lane, cf);
    double c[E], cs[E], cn[E], t0[E], kv[E], ff[E];
    load_col<E, MP>(y, task, lane, c);
    load_col<E, MP>(z2, task, lane, t0);
#pragma unroll
    for (int e = 0; e < E; ++e) c[e] = c[e] + t0[e];
    store_col<E, MP>(ynew, task, lane, c);
    load_col<E, MP>(y, cs_col, lane, cs);
    load_col<E, MP>(z2, cs_col, lane, t0);
#pragma unroll
    for (int e = 0; e < E; ++e) cs[e] = cs[e] + t0[e];
    load_col<E, MP>(y, cn_col, lane, cn);
    load_col<E, MP>(z2, cn_col, lane, t0);
#pragma unroll
    for (int e = 0; e < E; ++e) cn[e] = cn[e] + t0[e];
    load_col<E, MP>(kvp, j, lane, kv);
    tend_col<E, KIND>(P, cf, c, cs, cn, kv, tr, lane, ff);
    if constexpr (KIND == 2) forced_sources<E>(P, kvp, j, lane, c, ff);
    if constexpr (KIND == 1) {
        double u1[E], u2[E], v1[E], v2[E];
        phos_load_others<E>(P, tr, j, lane, y, u1, u2);
        phos_load_others<E>(P, tr, j, lane, z2, v1, v2);
        phos_add<E>(u1, u2, v1, v2);
        phos_sources<E>(P, tr, j, lane, c, u1, u2, cf.dzr, ff);
    }
    store_col<E>(f, task, lane, ff);
}

template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK)
    k_commit_tend(DevP P, const double* __restrict__ y, const double* __restrict__ z2, const double* __restrict__ kvp,
                  double* __restrict__ ynew, double* __restrict__ f) {
    TASK_PROLOGUE(P.ncol)
    commit_tend_body<E, KIND, 0>(P, y, z2, kvp, ynew, f, task, lane);
}

// The boundary between two steps in ONE launch: accepted step committed (y_new = y + Z2, f_new = fun(t_new, y_new)),
// Jacobian planes at t_new (optional), and the next attempt's set-up (vertical mixing planes at its three stage times,
// predicted stage values from the dense output of the step just taken).  Every piece reads only what the Newton
// iteration left behind or what its own wave writes: the commit and the prediction of a column are the same wave
// (the prediction reads the y_new that wave has just stored), the Jacobian of the modules served here does not depend
// on the state, and the new planes / stage values go to buffers nobody reads in this launch.
struct BoundaryArgs {
    const double *y, *z2, *kv_new;     // commit: state and third stage value of the step taken, plane at t_new
    double *ynew, *f;
    double *JL, *JU, *JS, *JN, *JC;
    int do_jac, nblk_vmix, nblk_jac;
    int jac_stage;                     // >= 0: Jacobian from the new plane of this stage (by the wave that computes it); then do_jac = 0
    int with_tend;                     // 0: y_new only (step replay: no error estimate will ask for f(t_new, y_new))
};

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
    hipLaunchKernelGGL(k_rows_sum, dim3((unsigned)nrows), dim3(NK2D_BLOCK), 0, c->stream, rows, c->ncol, out);
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
    DevP P = make_devp(c);
    P.guard = nullptr;
    NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_step_boundary<EE, KK>), dim3(B.nblk_vmix + B.nblk_jac + nk2d_grid(c->ncol)),
                                                         dim3(NK2D_BLOCK), 0, c->stream, P, V, B, A));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}

// sum((err / (atol + max(|y|, |y + Z2|) rtol))^2)  (radau.py:480-481)
template <int E, int MP = 0>
__device__ __forceinline__ void err_norm_body(const DevP& P, const double* __restrict__ y, const double* __restrict__ z2p,
                                              const double* __restrict__ err, double* __restrict__ part, int task, int lane) {
    double yy[E], z2[E], er[E];
    load_col<E, MP>(y, task, lane, yy);
    load_col<E, MP>(z2p, task, lane, z2);
    load_col<E, MP>(err, task, lane, er);
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double yn = yy[e] + z2[e];
        const double sc = P.atol + fmax(fabs(yy[e]), fabs(yn)) * P.rtol;
        const double a = er[e] / sc;
        acc += a * a;
    }
    acc = wave_sum(acc);
    if (lane == 0) st_mp<MP>(part + task, acc);
}

template <int E>
__global__ void k_err_norm(DevP P, const double* __restrict__ y, const double* __restrict__ z2p,
                           const double* __restrict__ err, double* __restrict__ part) {
    GUARD_RETURN(P.guard)
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
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_dense<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream,
                                              c->ncol, c->YOLD, c->ZP, c->nv, x, out));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}

// --- host wrappers used by the Radau driver --------------------------------------
static PredictArgs predict_args(nk2d_ctx* c, double x0, double x1, double x2) {
    PredictArgs A;
    A.y = c->Y; A.yold = c->YOLD; A.zp = c->ZP; A.z = c->Z; A.w = c->W;
    A.nv = c->nv;
    A.x0 = x0; A.x1 = x1; A.x2 = x2;
    return A;
}
int nk2d_r_predict(nk2d_ctx* c, double x0, double x1, double x2) {
    PredictArgs A = predict_args(c, x0, x1, x2);
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_predict<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream,
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
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_attempt_setup<EE>, dim3(nblk_vmix + nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0,
                                              c->stream, P, V, nblk_vmix, A, J));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
// algorithmic (unique) 8-byte words of one launch of the fused Newton iteration, P = nz*ny cells, N = tc*P values:
//   stage : read y, Z[3], W[3] (7N), kappa_v at 3 times + 4 static planes (7P),
//           write the 3 right-hand sides (3N) unless the update consumes them
//   sweep : Jacobian planes JL, JU (+JS, JN after the first sweep), pivot reciprocals
//           (real N + complex 2N), PCR tables (3 * 14/E * N), right-hand sides (3N, unless
//           just computed), previous iterate (3N, after the first sweep), new iterate (3N,
//           unless the update consumes it)
//   update: y (N, unless the stage read it), W read + write (6N), Z write (3N)
static double fused_words(const nk2d_ctx* c, bool do_stage, bool first, bool do_update, bool delta, bool do_factor) {
    const double Pc = (double)c->nz * c->ny, N = Pc * c->tc;
    double words = 0.0;
    if (do_stage) words += 7.0 * N + 7.0 * Pc + ((do_update || delta) ? 0.0 : 3.0 * N);
    const double fw = (c->factor_fp32 && !do_factor) ? 0.5 : 1.0;  // fp32 copies of the factorisation
    words += (first ? 2.0 : 4.0) * Pc + fw * (3.0 * N + 3.0 * 14.0 / c->E * N);  // factor read, or written when computed here
    if (do_factor) words += Pc;                                             // JC
    if (!do_stage && !delta) words += 3.0 * N;
    if (!first) words += 3.0 * N;
    if (!do_update) words += 3.0 * N;
    if (do_update) words += (do_stage ? 0.0 : N) + 9.0 * N;
    return words;
}

static void fill_fused_args(nk2d_ctx* c, FusedArgs& A, bool do_stage, bool first, bool do_update, double mreal,
                            double mcr, double mci, int src, bool delta) {
    A = {};
    A.st.y = c->Y; A.st.z = c->Z; A.st.w = c->W;
    A.st.zout = c->Z;
    A.st.kv[0] = c->KV[0]; A.st.kv[1] = c->KV[1]; A.st.kv[2] = c->KV[2];
    A.st.br = c->BR; A.st.bcr = c->BCR; A.st.bci = c->BCI;
    A.st.nv = c->nv; A.st.mreal = mreal; A.st.mcr = mcr; A.st.mci = mci;
    fill_factor_args(c, A.sw);
    A.sw.br = c->BR; A.sw.bcr = c->BCR; A.sw.bci = c->BCI;
    A.sw.xr_old = c->XR[src]; A.sw.xcr_old = c->XCR[src]; A.sw.xci_old = c->XCI[src];
    A.sw.xr_new = c->XR[1 - src]; A.sw.xcr_new = c->XCR[1 - src]; A.sw.xci_new = c->XCI[1 - src];
    A.sw.first = first ? 1 : 0;
    A.sw.cre = c->lu_cre; A.sw.ccr = c->lu_ccr; A.sw.cci = c->lu_cci;
    A.part = c->PART;
    A.do_stage = do_stage ? 1 : 0;
    A.do_update = do_update ? 1 : 0;
    A.delta = delta ? 1 : 0;
}

static int launch_fused(nk2d_ctx* c, const DevP& P, const FusedArgs& A, bool do_factor, bool do_stage) {
    if (c->team == 2) {    // a pair of waves per column (k_newton_pair)
        FinalArgs Fin = {};
        VmixArgs V = {};
        JacOut J = {};
        const dim3 grid(c->ncol), block(128);
        if (do_factor) {
            NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_pair<EE, KK, 1, 1, 0>), grid, block, 0, c->stream, P, A, Fin, V, J));
        } else if (do_stage) {
            NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_pair<EE, KK, 0, 1, 0>), grid, block, 0, c->stream, P, A, Fin, V, J));
        } else {
            NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_pair<EE, KK, 0, 0, 0>), grid, block, 0, c->stream, P, A, Fin, V, J));
        }
    } else if (c->team) {    // one workgroup of four waves per column (k_newton_team)
        if (do_factor) {
            NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_team<EE, KK, 1, 1>), dim3(nk2d_grid_xcd(c->ncol, 1)), dim3(NK2D_BLOCK), 0, c->stream, P, A));
        } else if (do_stage) {
            NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_team<EE, KK, 0, 1>), dim3(nk2d_grid_xcd(c->ncol, 1)), dim3(NK2D_BLOCK), 0, c->stream, P, A));
        } else {
            NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_team<EE, KK, 0, 0>), dim3(nk2d_grid_xcd(c->ncol, 1)), dim3(NK2D_BLOCK), 0, c->stream, P, A));
        }
    } else if (do_factor) {  // always a launch with the stage part
        NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_fused<EE, KK, 1, 1>), dim3(nk2d_grid_xcd(c->ncol, NK2D_WAVES_PER_BLOCK)), dim3(NK2D_BLOCK), 0, c->stream, P, A));
    } else if (do_stage || c->kind != 1) {
        NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_fused<EE, KK, 0, 1>), dim3(nk2d_grid_xcd(c->ncol, NK2D_WAVES_PER_BLOCK)), dim3(NK2D_BLOCK), 0, c->stream, P, A));
    } else {          // phosphorus, sweep-only launch: the lean instantiation
        NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_fused<EE, 1, 0, 0>), dim3(nk2d_grid_xcd(c->ncol, NK2D_WAVES_PER_BLOCK)), dim3(NK2D_BLOCK), 0, c->stream, P, A));
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
    nk2d_plane_job* job = c->plane_job;
    if (job && !job->done && !do_factor && c->team == 0 && c->kind != 1 && !c->xcd_map && P.guard == nullptr) {
        // the next attempt's planes behind this launch's columns (k_newton_fused_pl; see PreFactor)
        VmixArgs V;
        for (int i = 0; i < 3; ++i) {
            nk2d_host_interp(4, c->d.bld_tvals, c->d.bld_fvals, job->times[i], &V.frac[i]);
            V.out[i] = c->KVN[i];
        }
        V.frac[3] = 0.0; V.out[3] = nullptr;
        vmix_forcing_args(c, 3, job->times, V);
        V.bldmin = c->d.bldepth_min; V.y0 = c->d.vmix_log_shallow; V.y1 = c->d.vmix_log_deep;
        V.hw = c->d.vmix_half_width;
        JacOut J = {c->JB[0], c->JB[1], c->JB[2], c->JB[3], c->JB[4], job->jstage};
        const int nblk_cols = nk2d_grid(c->ncol);
        const dim3 grid(nblk_cols + nk2d_grid(c->ny * 3));
        if (c->kind == 2) {
            NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_fused_pl<EE, 2>), grid, dim3(NK2D_BLOCK), 0, c->stream, P, A, nblk_cols, V, J));
        } else {
            NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_fused_pl<EE, 0>), grid, dim3(NK2D_BLOCK), 0, c->stream, P, A, nblk_cols, V, J));
        }
        NK2D_CHECK(c, hipGetLastError());
        job->done = 1;
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
                        const double* times, double x0, double x1, double x2, int jac_stage, bool planes_done,
                        const double* next_shifts) {
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
    P.guard = nullptr;
    {
        const double words = fused_words(c, do_stage, first, true, delta, do_factor);
        // counted with the path's launches and bytes (end-to-end figures); not in the shape tallies of
        // k_newton_fused<E, KIND, 0, 1>: this is a kernel of its own, with the planes' work in it
        c->sweep_launches++;
        c->fused_bytes_all += 8.0 * words;
    }
    // planes_done: an earlier launch of this step already computed the next attempt's planes (k_newton_fused_pl); this
    // launch then carries that attempt's line factorisation instead, when its shifts are known and its Jacobian is new
    PreFactor F = {};
    F.mode = planes_done ? 1 : 0;
    const bool prefactor = planes_done && next_shifts != nullptr && jac_stage >= 0;
    if (prefactor) {
        F.mode = 2;
        F.sa.JL = c->JB[0]; F.sa.JU = c->JB[1]; F.sa.JS = c->JB[2]; F.sa.JN = c->JB[3]; F.sa.JC = c->JB[4];
        F.sa.cre = next_shifts[0]; F.sa.ccr = next_shifts[1]; F.sa.cci = next_shifts[2];
        F.sa.fr_inv = c->FB_INV; F.sa.fc_invr = c->FCB_INVR; F.sa.fc_invi = c->FCB_INVI;
        F.sa.fr_tab = c->FB_TAB; F.sa.fc_tabr = c->FCB_TABR; F.sa.fc_tabi = c->FCB_TABI;
        F.sa.f32 = 0;
        F.sa.nreal = c->ncol;
        F.sa.ntasks = 2 * c->ncol;
        const double Pc = (double)c->nz * c->ny, N = Pc * c->tc;
        c->fused_bytes_all += 8.0 * (3.0 * Pc + 3.0 * N + 3.0 * 14.0 / c->E * N);     // planes read, tables written
    }
    const dim3 grid(Fin.nblk_cols + (F.mode == 0 ? nk2d_grid(c->ny * 3) : (F.mode == 2 ? nk2d_grid(2 * c->ncol) : 0)));
#define NK2D_FINAL_LAUNCH(KK)                                                                                              \
    if (do_factor) {                                                                                                       \
        NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_final<EE, KK, 1, 1>), grid, dim3(NK2D_BLOCK), 0, c->stream, P, A, Fin, V, J, F)); \
    } else if (do_stage) {                                                                                                 \
        NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_final<EE, KK, 0, 1>), grid, dim3(NK2D_BLOCK), 0, c->stream, P, A, Fin, V, J, F)); \
    } else {                                                                                                               \
        NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_final<EE, KK, 0, 0>), grid, dim3(NK2D_BLOCK), 0, c->stream, P, A, Fin, V, J, F)); \
    }
    if (c->team == 2) {
        Fin.nblk_cols = c->ncol;
        const dim3 pgrid(c->ncol + (c->ny * 3 + 1) / 2), pblock(128);
#define NK2D_PAIR_FINAL(KK)                                                                                                \
        if (do_factor) {                                                                                                   \
            NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_pair<EE, KK, 1, 1, 1>), pgrid, pblock, 0, c->stream, P, A, Fin, V, J)); \
        } else if (do_stage) {                                                                                             \
            NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_pair<EE, KK, 0, 1, 1>), pgrid, pblock, 0, c->stream, P, A, Fin, V, J)); \
        } else {                                                                                                           \
            NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_pair<EE, KK, 0, 0, 1>), pgrid, pblock, 0, c->stream, P, A, Fin, V, J)); \
        }
        if (c->kind == 2) { NK2D_PAIR_FINAL(2) } else { NK2D_PAIR_FINAL(0) }
#undef NK2D_PAIR_FINAL
    } else if (c->kind == 2) { NK2D_FINAL_LAUNCH(2) } else { NK2D_FINAL_LAUNCH(0) }
#undef NK2D_FINAL_LAUNCH
    NK2D_CHECK(c, hipGetLastError());
    std::swap(c->Y, c->YOLD);
    if (do_stage) std::swap(c->Z, c->ZN);
    for (int i = 0; i < 3; ++i) std::swap(c->KV[i], c->KVN[i]);
    if (jac_stage >= 0) {
        std::swap(c->JL, c->JB[0]); std::swap(c->JU, c->JB[1]); std::swap(c->JS, c->JB[2]);
        std::swap(c->JN, c->JB[3]); std::swap(c->JC, c->JB[4]);
    }
    c->prefactored = 0;
    if (prefactor) {
        // the tables of the step that begins now: the integrator's next "LU" event with these shifts finds them in place
        std::swap(c->FR_INV, c->FB_INV); std::swap(c->FC_INVR, c->FCB_INVR); std::swap(c->FC_INVI, c->FCB_INVI);
        std::swap(c->FR_TAB, c->FB_TAB); std::swap(c->FC_TABR, c->FCB_TABR); std::swap(c->FC_TABI, c->FCB_TABI);
        c->prefactored = 1;
        c->pre_cre = F.sa.cre; c->pre_ccr = F.sa.ccr; c->pre_cci = F.sa.cci;
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
    NK2D_CHECK(c, hipMemcpyAsync(c->ZP, c->W, 3 * c->nv * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    FusedArgs A;
    fill_fused_args(c, A, do_stage, first, do_update, c->lu_cre, c->lu_ccr, c->lu_cci, 0, delta);
    A.st.w = c->ZP;
    A.st.zout = c->ZN;
    DevP P = make_devp(c);
    P.guard = nullptr;
    for (int i = 0; i < 3; ++i) NK2D_TRY(launch_fused(c, P, A, false, do_stage));   // warm-up
    NK2D_CHECK(c, hipEventRecord(c->timer_ev[0], c->stream));
    for (int i = 0; i < n; ++i) NK2D_TRY(launch_fused(c, P, A, false, do_stage));
    NK2D_CHECK(c, hipEventRecord(c->timer_ev[1], c->stream));
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
    if (c->prof_every <= 0 || c->prof_used + 2 > c->prof_ev.size()) return 0;
    // windows whose first launch also factorises (k_newton_fused<E, KIND, 1>, a different and heavier
    // kernel) are not timed: the windows measure k_newton_fused<E, KIND, 0> only
    if (c->factor_pending) return 0;
    if ((c->win_seq++ % c->prof_every) != 0) return 0;
    NK2D_CHECK(c, hipEventRecord(c->prof_ev[c->prof_used], c->stream));
    c->win_open = 1;
    c->win_launches = 0;
    c->win_bytes = 0.0;
    return 0;
}
int nk2d_prof_window_end(nk2d_ctx* c) {
    if (!c->win_open) return 0;
    c->win_open = 0;
    NK2D_CHECK(c, hipEventRecord(c->prof_ev[c->prof_used + 1], c->stream));
    c->prof_win_launches.push_back(c->win_launches);
    c->sweep_bytes += c->win_bytes;
    c->prof_used += 2;
    return 0;
}

int nk2d_r_err_rhs(nk2d_ctx* c, double h) {
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_err_rhs<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream,
                                               c->ncol, c->F, c->Z, c->nv, h, c->BR, c->cur_guard));
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
        NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_err_fused<EE, KK>), dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream, P, A));
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
    DevP P = make_devp(c);
    NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_err_rhs2<EE, KK>), dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream, P,
                                               c->Y, err, c->KV[3], c->Z, c->nv, h, c->BR));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
// y_new = Y + Z[2] -> YOLD (the spare buffer), F = fun(., y_new) with the plane kv
int nk2d_r_commit_tend(nk2d_ctx* c, const double* kv) {
    DevP P = make_devp(c);
    P.guard = nullptr;
    NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_commit_tend<EE, KK>), dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream, P,
                                               c->Y, c->Z + 2 * c->nv, kv, c->YOLD, c->F));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
int nk2d_r_err_norm(nk2d_ctx* c, const double* err) {
    DevP P = make_devp(c);
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_err_norm<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream, P,
                                               c->Y, c->Z + 2 * c->nv, err, c->part_on_host ? c->hPART : c->PART));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
int nk2d_r_wnorm(nk2d_ctx* c, const double* a, const double* b, double ca, double cb, const double* ys) {
    DevP P = make_devp(c);
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_wnorm<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream, P, a,
                                               b, ca, cb, ys, c->part_on_host ? c->hPART : c->PART));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
int nk2d_r_axpy(nk2d_ctx* c, const double* a, double s, const double* b, double* out) {
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_axpy<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream,
                                               c->ncol, a, s, b, out));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
int nk2d_r_final(nk2d_ctx* c, const double* y0, double* out) {
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_final<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream,
                                               c->ncol, c->ny, c->YOLD, c->ZP, c->nv, y0, c->MASK, out));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}

// =================================================================================================
// The whole forward year in ONE launch (nk2d_set_option "device_ctl" 3).
//
// The host-controlled integrator (nk2d_radau.hip) reads one scalar per simplified-Newton iteration and
// launches 60 000 small kernels per 416 x 416 year; at 26 x 26 ... 208 x 208 the year is nothing but
// launch gaps and host round trips.  Here every wave owns its (tracer, ypos) column for the WHOLE year:
// the phases of the Radau step (the same device functions the per-phase kernels call, in the same order)
// are separated by grid-wide barriers, and SciPy's controller (radau.py:399-539) runs redundantly in every
// wave -- all waves read the same norm partials, reduce them in the association of nk2d_part_sum and
// take identical decisions, so no decision is ever broadcast and the host is not involved until y(T).
//
// Visibility between workgroups follows the hand-off the guides validate for gfx950 (MI355X_MICROARCH.md,
// inter-workgroup visibility, table row 1): every array another workgroup may read is stored write-through
// and loaded L1-bypassing (MP = 1 accessors: relaxed agent-scope atomics = sc1); before a barrier every
// wave drains its stores (s_waitcnt vmcnt(0)), the workgroup joins, ONE lane adds to the arrival counter
// (agent scope) and polls it; the others wait at the workgroup barrier behind that lane.  Arrays only ever
// touched by their owning wave (W, right-hand sides, the line factorisation, F) stay plain.  Every spin is
// bounded; a timeout raises a grid-wide abort flag that every wave sees at its next barrier.
// The grid is launched cooperatively, so it is rejected -- not deadlocked -- when it is not fully resident.
// =================================================================================================
#define NK2D_SPIN_LIMIT 4000000

struct YearArgs {
    double *Y, *YOLD, *F, *Z, *ZP, *ZN, *W;
    double *BR, *BCR, *BCI, *XR[2], *XCR[2], *XCI[2], *TMP;
    double* KV[4];
    SweepArgs fac;             // Jacobian planes + factor pointers (the other members are set per phase)
    double* PART;              // [2][ncol]: norm partials, the two halves alternate from one reduction to the next
    double t0, t1, h_abs0, max_step, newton_tol, n_total, growth_cap;
    int jac_fresh, f32;
    int jac_stage;             // >= 0: Jacobian of an attempt from the plane of this stage time (option "jac_stage")
    double bld_t[4], bld_f[4], bldmin, vy0, vy1, hw;
    const int* m_tab;          // sweeps for the shift bucket k (host: nk2d_sweeps_for), n_tab entries
    int n_tab;
    double rho_c0, rho_dlog;
    unsigned* arrive;          // grid barrier arrival counter (zeroed by the host)
    int* abort_flag;
    double* out;               // [32]: status, t, counters, swap parities, bytes
    double* record;            // accepted steps [cap][NK2D_SCHED_WIDTH] or null
    double fingerprint;        // of the context (recorded with every step)
    long long record_cap;
    long long spin_ticks;      // longest wait at a grid barrier, in ticks of s_memrealtime (100 MHz)
    int fences;                // 1: agent-scope release / acquire fences around every grid barrier (option "year_fences")
};

// Arrival counter in NK2D_BAR_SHARDS shards, each on a 128-byte line of its own: an agent-scope atomic executes at
// the memory side and adds to ONE address serialise (MI355X_MICROARCH.md, global atomics: ~50 ns each) -- with 200
// workgroups on one counter the arrivals alone cost 10 us.  A workgroup adds to shard (blockIdx & 31); the polling
// wave reads all shards with one load instruction (lane i reads shard i) and sums them.
#define NK2D_BAR_SHARDS 32
#define NK2D_BAR_STRIDE 32   /* unsigned ints between shards = 128 bytes */

struct GridBarrier {
    unsigned* arrive;
    int* abort_flag;
    unsigned nwg, epoch;
    int* lds_ok;
    long long spin_ticks;
    int fences;
    int xcd = 0;     // 1: every workgroup of the barrier sits on one XCD -- ONE counter, adds executed in that XCD's L2
    int wg_id = 0;   // this workgroup's number among them (xcd = 0: blockIdx.x)
    __device__ __forceinline__ bool sync() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have left
        // validation mode: the textbook hand-off as well (every wave releases before the barrier and acquires after it), which
        // also covers an array the write-through / L1-bypassing accessors might have missed -- results must not change
        if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x < 64) {     // the first wave arrives for the workgroup and polls
            const int lane = threadIdx.x;
            const unsigned target = (epoch + 1u) * nwg;
            if (lane == 0) {
                if (xcd) __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_fetch_add(arrive + (size_t)(blockIdx.x % NK2D_BAR_SHARDS) * NK2D_BAR_STRIDE, 1u,
                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int good = 1;
            long long spins = 0;
            const long long t_begin = (long long)__builtin_amdgcn_s_memrealtime();
            for (;;) {
                unsigned v = 0u;
                if (lane < (xcd ? 1 : NK2D_BAR_SHARDS))
                    v = __hip_atomic_load(arrive + (size_t)lane * NK2D_BAR_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
                const unsigned total = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
                if (total >= target) break;
                const int ab = __builtin_amdgcn_readfirstlane(
                    (lane == 0) ? __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0);
                // bounded by TIME (a slow co-tenant must not fail a year that is merely waiting), and by a spin count as a
                // last resort should the clock not advance
                // (the clock is read every 64th poll; a limit of zero -- tests -- gives up at the first poll that has to wait)
                const bool late = ((++spins & 63) == 0 || spin_ticks == 0) &&
                                  (long long)__builtin_amdgcn_s_memrealtime() - t_begin > spin_ticks;
                if (late || spins > 4000LL * NK2D_SPIN_LIMIT || ab != 0) {
                    if (lane == 0) __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    good = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) *lds_ok = good;
        }
        __syncthreads();
        if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ++epoch;
        return *lds_ok != 0;
    }
};

// Synchronisation block of the one-launch years: arrival counters (32 shards on lines of their own), abort flag at 4096,
// tickets of the XCD flavour at 6144, and from 8192 one 128-byte line per column for NeighbourSync.
static size_t yr_sync_bytes(const nk2d_ctx* c) { return 8192 + (size_t)c->ncol * 128; }

// Where a workgroup is ONE column (the team flavour of k_frozen_persistent) the grid barrier asks for more than the data
// flow needs: a column reads what its two lateral neighbours (same tracer) wrote in the phase before, and nothing else of
// another workgroup.  So a column publishes the number of phases it has completed -- after every wave of the team has
// drained its write-through stores -- and waits until both neighbours have completed as many: it then runs at most one
// phase ahead of them, which is also what the buffers that alternate between phases (Z / ZN, the sweep iterates) and the
// ones rewritten in place two phases later need.  Point to point instead of all to all: no counter everybody adds to, no
// waiting for the slowest of all workgroups in every phase.  Same accessors, same bounded wait, same abort flag.
struct NeighbourSync {
    unsigned* flags;     // [ncol][32]: phases completed, one 128-byte line per column
    int* abort_flag;
    int me, left, right; // columns (left / right: -1 at the edge of the tracer's plane)
    unsigned phase;
    int* lds_ok;
    long long spin_ticks;
    int fences;
    __device__ __forceinline__ bool sync() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        ++phase;
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            if (lane == 0) __hip_atomic_store(flags + (size_t)me * 32, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int other = (lane == 0) ? left : ((lane == 1) ? right : -1);
            int good = 1;
            long long spins = 0;
            const long long t_begin = (long long)__builtin_amdgcn_s_memrealtime();
            for (;;) {
                unsigned v = phase;
                if (other >= 0) v = __hip_atomic_load(flags + (size_t)other * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((int)(v >= phase))) break;
                const int ab = __builtin_amdgcn_readfirstlane(
                    (lane == 0) ? __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0);
                const bool late = ((++spins & 63) == 0 || spin_ticks == 0) &&
                                  (long long)__builtin_amdgcn_s_memrealtime() - t_begin > spin_ticks;
                if (late || spins > 4000LL * NK2D_SPIN_LIMIT || ab != 0) {
                    if (lane == 0) __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    good = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) *lds_ok = good;
        }
        __syncthreads();
        if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        return *lds_ok != 0;
    }
};

// sum of the ncol per-column partials in the association of nk2d_part_sum / k_reduce (256 strided
// accumulators, then a binary tree), identical in every wave
__device__ __forceinline__ double year_part_sum(const double* part, int n, int lane) {
    double acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc[q] = 0.0;
        for (int i = lane + 64 * q; i < n; i += NK2D_BLOCK) acc[q] += ld_mp<1>(part + i);
    }
    acc[0] += acc[2];   // sh[t] += sh[t + 128]
    acc[1] += acc[3];
    double v = acc[0] + acc[1];   // sh[t] += sh[t + 64]
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return __shfl(v, 0, 64);
}

__device__ __forceinline__ double year_interp4(const double* xp, const double* fp, double x) {
    if (x > xp[3]) return fp[3];
    if (x < xp[0]) return fp[0];
    int j = 0;
    while (j + 1 < 4 && xp[j + 1] <= x) ++j;
    if (j == 3 || xp[j] == x) return fp[j];
    const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    return slope * (x - xp[j]) + fp[j];
}

__device__ __forceinline__ int year_sweeps_for(const YearArgs& A, double c_real) {
    if (A.n_tab <= 0) return 1;
    const double pos = log10(c_real / A.rho_c0) / A.rho_dlog;
    int k = (int)floor(pos);
    if (k < 0) return 400;
    if (k >= A.n_tab) k = A.n_tab - 1;
    return A.m_tab[k];
}

__device__ __forceinline__ double year_predict_factor(double h_abs, bool has_h_old, double h_abs_old, double err,
                                                      bool has_err_old, double err_old) {
    double mult = 1.0;
    if (has_err_old && has_h_old && err != 0.0) mult = h_abs / h_abs_old * pow(err_old / err, 0.25);
    return fmin(1.0, mult) * pow(err, -0.25);
}

// values every lane of every wave holds identically: tell the compiler (scalar registers, uniform branches)
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool uni_b(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
__device__ __forceinline__ double uni_d(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK) k_year_persistent(DevP P, YearArgs A) {
    __shared__ int lds_ok;
    const int lane = threadIdx.x & 63;
    const int wave = uni_i((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    const bool col_wave = wave < P.ncol;             // this wave owns column `wave` for the whole year
    GridBarrier bar{A.arrive, A.abort_flag, gridDim.x, 0u, &lds_ok, A.spin_ticks, A.fences};
    const double RC0 = 0.15505102572168222, RC1 = 0.6449489742783178, RC2 = 1.0;
    const double MU_REAL = 3.637834252744496, MU_CR = 2.6810828736277523, MU_CI = -3.050430199247411;
    const int NEWTON_MAXITER = 6;
    const size_t nv = (size_t)P.ncol * (E * 64);

    // buffers that swap roles: parities, the pointers are selected where they are used
    // the three stage buffers rotate: current iterate, previous step's (dense output), spare (single-launch iterations)
    int swapY = 0, swapKV = 0, zc = 0, zp = 1, zn = 2;
#define YR_ZBUF(i) (((i) == 0) ? A.Z : (((i) == 1) ? A.ZP : A.ZN))
#define YR_Y (swapY ? A.YOLD : A.Y)
#define YR_YOLD (swapY ? A.Y : A.YOLD)
#define YR_Z YR_ZBUF(zc)
#define YR_ZP YR_ZBUF(zp)
#define YR_ZN YR_ZBUF(zn)
#define YR_KV2 (swapKV ? A.KV[3] : A.KV[2])
#define YR_KV3 (swapKV ? A.KV[2] : A.KV[3])
    // controller state (identical in every wave)
    double t = A.t0, h_abs_s = A.h_abs0, h_abs_old_s = 0.0, err_old_s = 0.0;
    bool has_old_h = false, has_old_err = false, current_jac = true, have_lu = false, have_dense = false;
    double h_lu = 0.0, t_jac = A.t0, dense_t_old = 0.0, dense_h = 0.0;
    int m_real = 1, m_cplx = 1;
    bool factor_pending = false;
    bool pre_setup = false;      // the next attempt's planes and predicted stage values came with the commit phase
    double pre_h = 0.0;
    double lu_cre = 0.0, lu_ccr = 0.0, lu_cci = 0.0;
    int nfev = 0, njev = 0, nlu = 0, nsteps = 0, nrejected = 0, nnewton = 0, nsolve = 0, nsweeps = 0, nrec = 0;
    double words = 0.0;
    int status = 0;     // 0 ok, 1 barrier timeout, 2 non-finite step, 3 step too small
    // Norm partials alternate between two buffers: the waves still summing reduction n must not see the
    // partials of reduction n + 1, which a faster wave may already write (a single phase can lie between them)
    unsigned pev = 0;
#define YEAR_PART() (A.PART + (size_t)(pev & 1u) * P.ncol)
#define YEAR_SYNC() \
    if (!bar.sync()) { status = 1; goto finish; }
    const double Pc = (double)P.nz * P.ny, Ntot = Pc * P.tc;

    // Jacobian planes from the vertical mixing plane kv (tasks spread over all waves)
#define YEAR_JAC(kvp)                                                                                                   \
    for (int task = wave; task < P.ny; task += nwaves)                                                                  \
        jac_body<E, 1>(P, kvp, const_cast<double*>(A.fac.JL), const_cast<double*>(A.fac.JU), const_cast<double*>(A.fac.JS), \
                       const_cast<double*>(A.fac.JN), const_cast<double*>(A.fac.JC), nullptr, nullptr, task, lane);

    // one (stage time, ypos column) task of an attempt's planes; with option "jac_stage" the wave that computes the
    // column of that stage derives the Jacobian planes of the column from it
    const bool jac_at_stage = A.jac_stage >= 0;
    const double rc_jac = (A.jac_stage == 0) ? RC0 : ((A.jac_stage == 1) ? RC1 : RC2);
#define YEAR_PLANE(ti, fr, dst, j)                                                                                     \
    {                                                                                                                   \
        double kvc_[E];                                                                                                 \
        vmix_col_regs<E>(P, A.bldmin, A.vy0, A.vy1, A.hw, fr, j, lane, kvc_);                                           \
        store_col<E, 1>(dst, j, lane, kvc_);                                                                            \
        if ((ti) == A.jac_stage)                                                                                        \
            jac_core<E, 1>(P, kvc_, dst, const_cast<double*>(A.fac.JL), const_cast<double*>(A.fac.JU),                  \
                           const_cast<double*>(A.fac.JS), const_cast<double*>(A.fac.JN), const_cast<double*>(A.fac.JC), \
                           nullptr, nullptr, j, lane);                                                                  \
    }

    while (t < A.t1) {
        const double min_step = 10.0 * fabs(nextafter(t, INFINITY) - t);
        double h_abs, h_abs_old = 0.0, err_old = 0.0;
        bool has_h_old, has_err_old;
        if (h_abs_s > A.max_step) { h_abs = A.max_step; has_h_old = has_err_old = false; }
        else if (h_abs_s < min_step) { h_abs = min_step; has_h_old = has_err_old = false; }
        else { h_abs = h_abs_s; h_abs_old = h_abs_old_s; err_old = err_old_s; has_h_old = has_old_h; has_err_old = has_old_err; }
        if (A.jac_fresh && !current_jac && !jac_at_stage) {
            YEAR_JAC(YR_KV3)     // KV3 holds the plane at the current t
            YEAR_SYNC()
            t_jac = t; ++njev; current_jac = true; have_lu = false;
        }
        bool rejected = false, accepted = false, newton_failed = false;
        double h = 0.0, t_new = 0.0, err = 0.0, safety = 0.0, rate = 0.0;
        bool have_rate = false;
        int n_iter = 0;
        while (!accepted) {
            if (uni_b(!isfinite(h_abs))) { status = 2; goto finish; }
            if (uni_b(h_abs < min_step)) { status = 3; goto finish; }
            h = h_abs;
            t_new = t + h;
            if (t_new - A.t1 > 0) t_new = A.t1;
            h = uni_d(t_new - t);
            t_new = uni_d(t_new);
            h_abs = fabs(h);
            // stage planes at the three collocation times + predicted stage values (radau.py:445-448); the first
            // attempt of a step normally got them in the commit phase of the step before
            const bool have_setup = pre_setup && uni_b(h == pre_h);
            pre_setup = false;
            if (!have_setup) {
                const double f0 = uni_d(year_interp4(A.bld_t, A.bld_f, t + (h * RC0)));
                const double f1 = uni_d(year_interp4(A.bld_t, A.bld_f, t + (h * RC1)));
                const double f2 = uni_d(year_interp4(A.bld_t, A.bld_f, t + (h * RC2)));
                for (int task = wave; task < 3 * P.ny; task += nwaves) {
                    const int ti = task / P.ny, j = task - ti * P.ny;
                    YEAR_PLANE(ti, (ti == 0) ? f0 : ((ti == 1) ? f1 : f2), (ti == 0) ? A.KV[0] : ((ti == 1) ? A.KV[1] : YR_KV2), j)
                }
                if (col_wave) {
                    if (have_dense) {
                        PredictArgs PA;
                        PA.y = YR_Y; PA.yold = YR_YOLD; PA.zp = YR_ZP; PA.z = YR_Z; PA.w = A.W; PA.nv = nv;
                        PA.x0 = ((t + h * RC0) - dense_t_old) / dense_h;
                        PA.x1 = ((t + h * RC1) - dense_t_old) / dense_h;
                        PA.x2 = ((t + h * RC2) - dense_t_old) / dense_h;
                        predict_body<E, 1>(PA, wave, lane);
                    } else {
                        double zero[E];
#pragma unroll
                        for (int e = 0; e < E; ++e) zero[e] = 0.0;
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            store_col<E, 1>(YR_Z + i * nv, wave, lane, zero);
                            store_col<E>(A.W + i * nv, wave, lane, zero);
                        }
                    }
                }
                YEAR_SYNC()
            }
            if (jac_at_stage) { t_jac = t + (h * rc_jac); ++njev; current_jac = true; have_lu = false; }
            bool converged = false;
            while (!converged) {
                if (!have_lu) {
                    h_lu = h; have_lu = true;
                    m_real = uni_i(year_sweeps_for(A, MU_REAL / h));
                    m_cplx = uni_i(year_sweeps_for(A, MU_CR / h));
                    nlu += 2;
                    lu_cre = MU_REAL / h; lu_ccr = MU_CR / h; lu_cci = MU_CI / h;
                    factor_pending = true;
                }
                // simplified Newton iterations (radau.py:48-136)
                const double mreal = MU_REAL / h, mcr = MU_CR / h, mci = MU_CI / h;
                const int m = (m_real > m_cplx) ? m_real : m_cplx;
                double dW_norm_old = 0.0;
                bool has_old = false;
                have_rate = false; rate = 0.0;
                converged = false;
                int k = 0;
                for (k = 0; k < NEWTON_MAXITER; ++k) {
                    int src = 0;
                    for (int it = 0; it < m; ++it) {
                        const bool do_stage = it == 0, first = it == 0, do_update = it == m - 1, delta = m == 2;
                        const bool do_factor = factor_pending && it == 0;
                        FusedArgs FA = {};
                        FA.st.y = YR_Y; FA.st.z = YR_Z; FA.st.w = A.W;
                        // a single-sweep solve is ONE phase: its update writes the spare buffer (the neighbours still
                        // read the old stage values in this phase), then the buffers swap
                        FA.st.zout = (do_stage && do_update) ? YR_ZN : YR_Z;
                        FA.st.kv[0] = A.KV[0]; FA.st.kv[1] = A.KV[1]; FA.st.kv[2] = YR_KV2;
                        FA.st.br = A.BR; FA.st.bcr = A.BCR; FA.st.bci = A.BCI;
                        FA.st.nv = nv; FA.st.mreal = mreal; FA.st.mcr = mcr; FA.st.mci = mci;
                        FA.sw = A.fac;
                        FA.sw.br = A.BR; FA.sw.bcr = A.BCR; FA.sw.bci = A.BCI;
                        FA.sw.xr_old = src ? A.XR[1] : A.XR[0]; FA.sw.xcr_old = src ? A.XCR[1] : A.XCR[0];
                        FA.sw.xci_old = src ? A.XCI[1] : A.XCI[0];
                        FA.sw.xr_new = src ? A.XR[0] : A.XR[1]; FA.sw.xcr_new = src ? A.XCR[0] : A.XCR[1];
                        FA.sw.xci_new = src ? A.XCI[0] : A.XCI[1];
                        FA.sw.first = first ? 1 : 0;
                        FA.sw.cre = lu_cre; FA.sw.ccr = lu_ccr; FA.sw.cci = lu_cci;
                        FA.sw.f32 = 0;
                        FA.part = YEAR_PART();
                        FA.do_stage = do_stage ? 1 : 0; FA.do_update = do_update ? 1 : 0; FA.delta = delta ? 1 : 0;
                        if (col_wave) {
                            if (do_factor) newton_fused_body<E, KIND, 1, 1, 1>(P, FA, wave, lane);
                            else newton_fused_body<E, KIND, 0, 1, 1>(P, FA, wave, lane);
                        }
                        {   // algorithmic bytes, as nk2d_r_newton_fused counts them
                            double wd = 0.0;
                            if (do_stage) wd += 7.0 * Ntot + 7.0 * Pc + ((do_update || delta) ? 0.0 : 3.0 * Ntot);
                            wd += (first ? 2.0 : 4.0) * Pc + (3.0 * Ntot + 3.0 * 14.0 / E * Ntot);
                            if (do_factor) wd += Pc;
                            if (!do_stage && !delta) wd += 3.0 * Ntot;
                            if (!first) wd += 3.0 * Ntot;
                            if (!do_update) wd += 3.0 * Ntot;
                            if (do_update) wd += (do_stage ? 0.0 : Ntot) + 9.0 * Ntot;
                            words += wd;
                        }
                        ++nsweeps;
                        if (it == 0) factor_pending = false;
                        src = 1 - src;
                        if (do_stage && do_update) { const int tmp = zc; zc = zn; zn = tmp; }
                        YEAR_SYNC()
                    }
                    nsolve += 2; nfev += 3; ++nnewton;
                    const double sum = uni_d(year_part_sum(YEAR_PART(), P.ncol, lane));
                    ++pev;
                    const double dW_norm = sqrt(sum) / sqrt(3.0 * A.n_total);
                    if (uni_b(!(dW_norm == dW_norm))) break;
                    if (has_old) { rate = uni_d(dW_norm / dW_norm_old); have_rate = true; }
                    if (have_rate && uni_b(rate >= 1.0 || pow(rate, (double)(NEWTON_MAXITER - k)) / (1.0 - rate) * dW_norm > A.newton_tol)) break;
                    if (uni_b(dW_norm == 0.0 || (have_rate && rate / (1.0 - rate) * dW_norm < A.newton_tol))) { converged = true; break; }
                    dW_norm_old = dW_norm;
                    has_old = true;
                }
                n_iter = (k < NEWTON_MAXITER) ? k + 1 : NEWTON_MAXITER;
                if (!converged) {
                    if (current_jac) break;
                    // stale Jacobian: refresh it and repeat the iteration from the predicted stage values
                    // (radau.py:462-470: solve_collocation_system starts from Z0 again)
                    YEAR_JAC(YR_KV3)
                    if (col_wave) {
                        if (have_dense) {
                            PredictArgs PA;
                            PA.y = YR_Y; PA.yold = YR_YOLD; PA.zp = YR_ZP; PA.z = YR_Z; PA.w = A.W; PA.nv = nv;
                            PA.x0 = ((t + h * RC0) - dense_t_old) / dense_h;
                            PA.x1 = ((t + h * RC1) - dense_t_old) / dense_h;
                            PA.x2 = ((t + h * RC2) - dense_t_old) / dense_h;
                            predict_body<E, 1>(PA, wave, lane);
                        } else {
                            double zero[E];
#pragma unroll
                            for (int e = 0; e < E; ++e) zero[e] = 0.0;
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                store_col<E, 1>(YR_Z + i * nv, wave, lane, zero);
                                store_col<E>(A.W + i * nv, wave, lane, zero);
                            }
                        }
                    }
                    YEAR_SYNC()
                    t_jac = t; ++njev; current_jac = true; have_lu = false;
                }
            }
            if (!converged) {
                h_abs = uni_d(h_abs * 0.5);
                have_lu = false;
                newton_failed = true;
                continue;
            }
            // error estimate (radau.py:477-487)
            int buf = 0;
            {
                if (m_real <= 2) {
                    int src = 0;
                    for (int it = 0; it < m_real; ++it) {
                        ErrArgs EA = {};
                        EA.sw = A.fac;
                        EA.f = A.F; EA.z = YR_Z; EA.y = YR_Y; EA.nv = nv; EA.h = h; EA.part = YEAR_PART();
                        EA.sw.xr_old = src ? A.XR[1] : A.XR[0]; EA.sw.xr_new = src ? A.XR[0] : A.XR[1];
                        EA.stage = it; EA.last = (it == m_real - 1) ? 1 : 0;
                        if (col_wave) err_fused_body<E, KIND, 1>(P, EA, wave, lane);
                        src = 1 - src;
                        ++nsweeps;
                        YEAR_SYNC()
                    }
                    buf = src;
                } else {
                    if (col_wave) err_rhs_body<E, 1>(A.F, YR_Z, nv, h, A.BR, wave, lane);
                    // the first sweep reads only its own column's right-hand side: no barrier before it
                    int src = 0;
                    const int tr = wave / P.ny, j = wave - tr * P.ny;
                    for (int it = 0; it < m_real; ++it) {
                        SweepArgs SA = A.fac;
                        SA.br = A.BR; SA.nreal = P.ncol; SA.ntasks = P.ncol;
                        SA.xr_old = src ? A.XR[1] : A.XR[0]; SA.xr_new = src ? A.XR[0] : A.XR[1]; SA.first = (it == 0) ? 1 : 0;
                        if (col_wave) sweep_body<E, KIND, 1>(P, SA, j * P.tc + tr, lane);
                        src = 1 - src;
                        ++nsweeps;
                        YEAR_SYNC()
                    }
                    buf = src;
                    if (col_wave) err_norm_body<E, 1>(P, YR_Y, YR_Z + 2 * nv, buf ? A.XR[1] : A.XR[0], YEAR_PART(), wave, lane);
                    YEAR_SYNC()
                }
                ++nsolve;
            }
            {
                const double sum = uni_d(year_part_sum(YEAR_PART(), P.ncol, lane));
                ++pev;
                err = sqrt(sum) / sqrt(A.n_total);
            }
            safety = 0.9 * (2 * NEWTON_MAXITER + 1) / (2 * NEWTON_MAXITER + n_iter);
            if (rejected && uni_b(err > 1)) {
                // filtered estimate: error <- solve(fun(t, y + error) + Z^T E / h)  (radau.py:485-487)
                if (col_wave) {
                    double tmp[E];
                    load_col<E, 1>(buf ? A.XR[1] : A.XR[0], wave, lane, tmp);
                    store_col<E, 1>(A.TMP, wave, lane, tmp);
                }
                YEAR_SYNC()
                if (col_wave) err_rhs2_body<E, KIND, 1>(P, YR_Y, A.TMP, YR_KV3, YR_Z, nv, h, A.BR, wave, lane);
                ++nfev;
                int src = 0;
                const int tr = wave / P.ny, j = wave - tr * P.ny;
                for (int it = 0; it < m_real; ++it) {
                    SweepArgs SA = A.fac;
                    SA.br = A.BR; SA.nreal = P.ncol; SA.ntasks = P.ncol;
                    SA.xr_old = src ? A.XR[1] : A.XR[0]; SA.xr_new = src ? A.XR[0] : A.XR[1]; SA.first = (it == 0) ? 1 : 0;
                    if (col_wave) sweep_body<E, KIND, 1>(P, SA, j * P.tc + tr, lane);
                    src = 1 - src;
                    ++nsweeps;
                    YEAR_SYNC()
                }
                buf = src;
                ++nsolve;
                if (col_wave) err_norm_body<E, 1>(P, YR_Y, YR_Z + 2 * nv, buf ? A.XR[1] : A.XR[0], YEAR_PART(), wave, lane);
                YEAR_SYNC()
                const double sum = uni_d(year_part_sum(YEAR_PART(), P.ncol, lane));
                ++pev;
                err = sqrt(sum) / sqrt(A.n_total);
            }
            if (uni_b(err > 1)) {
                const double factor = year_predict_factor(h_abs, has_h_old, h_abs_old, err, has_err_old, err_old);
                h_abs = uni_d(h_abs * fmax(0.2, safety * factor));
                have_lu = false;
                rejected = true;
                ++nrejected;
            } else {
                accepted = true;
            }
        }
        const bool recompute_jac = uni_b(n_iter > 2 && have_rate && rate > 1e-3);
        double factor = year_predict_factor(h_abs, has_h_old, h_abs_old, err, has_err_old, err_old);
        factor = fmin(10.0, safety * factor);
        if (newton_failed && A.growth_cap > 0.0) factor = fmin(factor, A.growth_cap);
        factor = uni_d(factor);
        const double h_lu_used = h_lu;
        if (!recompute_jac && uni_b(factor < 1.2)) factor = 1;
        else have_lu = false;
        if (A.record && nrec < A.record_cap && wave == 0 && lane == 0) {
            double* r = A.record + (size_t)nrec * NK2D_SCHED_WIDTH;
            r[0] = t; r[1] = t_new; r[2] = h; r[3] = (double)n_iter; r[4] = t_jac; r[5] = h_lu_used;
            r[6] = err; r[7] = A.fingerprint;
        }
        ++nrec;
        // y_new, f_new = fun(t_new, y_new)
        const double h_abs_next = uni_d(h_abs * factor);
        const bool jac_due = recompute_jac || (A.jac_fresh != 0);
        bool fusedb = uni_b(t + h == t_new) && uni_b(t_new < A.t1);
        double h2 = 0.0;
        if (fusedb) {
            const double min_step2 = 10.0 * fabs(nextafter(t_new, INFINITY) - t_new);
            double h_abs2 = h_abs_next;
            if (h_abs2 > A.max_step) h_abs2 = A.max_step;
            else if (h_abs2 < min_step2) h_abs2 = min_step2;
            double t_new2 = t_new + h_abs2;
            if (t_new2 - A.t1 > 0) t_new2 = A.t1;
            h2 = uni_d(t_new2 - t_new);
            fusedb = uni_b(isfinite(h2) && h2 > 0.0);
        }
        if (fusedb) {
            // ONE phase for the whole boundary: commit, Jacobian at t_new where due, and the next attempt's planes and
            // predicted stage values (every piece reads what the Newton iteration left or what its own wave writes)
            const double f0 = uni_d(year_interp4(A.bld_t, A.bld_f, t_new + (h2 * RC0)));
            const double f1 = uni_d(year_interp4(A.bld_t, A.bld_f, t_new + (h2 * RC1)));
            const double f2 = uni_d(year_interp4(A.bld_t, A.bld_f, t_new + (h2 * RC2)));
            for (int task = wave; task < 3 * P.ny; task += nwaves) {
                const int ti = task / P.ny, j = task - ti * P.ny;
                YEAR_PLANE(ti, (ti == 0) ? f0 : ((ti == 1) ? f1 : f2), (ti == 0) ? A.KV[0] : ((ti == 1) ? A.KV[1] : YR_KV3), j)
            }
            if (jac_due && !jac_at_stage) { YEAR_JAC(YR_KV2) }      // the third stage plane is the plane at t_new
            if (col_wave) {
                commit_tend_body<E, KIND, 1>(P, YR_Y, YR_Z + 2 * nv, YR_KV2, YR_YOLD, A.F, wave, lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's y_new is in memory before it reads it back
                PredictArgs PA;
                PA.y = YR_YOLD; PA.yold = YR_Y; PA.zp = YR_Z; PA.z = YR_ZP; PA.w = A.W; PA.nv = nv;
                PA.x0 = ((t_new + h2 * RC0) - t) / (t_new - t);
                PA.x1 = ((t_new + h2 * RC1) - t) / (t_new - t);
                PA.x2 = ((t_new + h2 * RC2) - t) / (t_new - t);
                predict_body<E, 1>(PA, wave, lane);
            }
            swapKV ^= 1;
            swapY ^= 1;
            { const int tmp = zc; zc = zp; zp = tmp; }
            have_dense = true; dense_t_old = t; dense_h = uni_d(t_new - t);
            t = t_new;
            ++nsteps; ++nfev;
            YEAR_SYNC()
            pre_setup = true; pre_h = h2;
            if (jac_at_stage) {
                current_jac = false;     // came with the planes; booked when the attempt starts
            } else if (jac_due) {
                t_jac = t; ++njev; current_jac = true;
                if (!recompute_jac) have_lu = false;
            } else {
                current_jac = false;
            }
        } else {
            if (uni_b(t + h == t_new)) {
                swapKV ^= 1;        // the third stage plane is the plane at t_new
            } else {
                const double fr = uni_d(year_interp4(A.bld_t, A.bld_f, t_new));
                for (int j = wave; j < P.ny; j += nwaves) vmix_col<E, 1>(P, A.bldmin, A.vy0, A.vy1, A.hw, fr, YR_KV3, j, lane);
                YEAR_SYNC()
            }
            if (col_wave) commit_tend_body<E, KIND, 1>(P, YR_Y, YR_Z + 2 * nv, YR_KV3, YR_YOLD, A.F, wave, lane);
            swapY ^= 1;
            { const int tmp = zc; zc = zp; zp = tmp; }
            have_dense = true; dense_t_old = t; dense_h = uni_d(t_new - t);
            t = t_new;
            ++nsteps; ++nfev;
            YEAR_SYNC()
            if (recompute_jac) {
                YEAR_JAC(YR_KV3)
                YEAR_SYNC()
                t_jac = t; ++njev; current_jac = true;
            } else {
                current_jac = false;
            }
        }
        h_abs_old_s = h_abs_s; has_old_h = true;
        err_old_s = err; has_old_err = true;
        h_abs_s = h_abs_next;
    }
finish:
    if (wave == 0 && lane == 0) {
        double* o = A.out;
        o[0] = (double)status; o[1] = t; o[2] = (double)nfev; o[3] = (double)njev; o[4] = (double)nlu;
        o[5] = (double)nsteps; o[6] = (double)nrejected; o[7] = (double)nnewton; o[8] = (double)nsolve;
        o[9] = (double)nsweeps; o[10] = (double)nrec; o[11] = (double)swapY; o[12] = (double)(zc + 4 * zp + 16 * zn);
        o[13] = (double)swapKV; o[14] = 8.0 * words; o[15] = (double)bar.epoch; o[16] = t_jac;
    }
#undef YEAR_SYNC
#undef YEAR_JAC
#undef YEAR_PLANE
#undef YR_Y
#undef YR_YOLD
#undef YR_Z
#undef YR_ZP
#undef YR_ZN
#undef YR_ZBUF
#undef YR_KV2
#undef YR_KV3
#undef YEAR_PART
}

// Cooperative launches of one process go through ONE queue of the HIP runtime, created on first use: contexts driven from
// several host threads (the tracer modules of a ModelState run their years in a thread pool) enqueue on it one at a time --
// two threads inside hipLaunchCooperativeKernel at once left the runtime with a queue it crashed on when the process ended
// (rocr::AMD::AqlQueue::~AqlQueue under hsa_shut_down; tools/probe_exit2.py).  Held for the enqueue only.
static std::mutex& coop_launch_mutex() {
    static std::mutex m;
    return m;
}

// host side: run the stepping loop of a forward year in the persistent kernel.  The caller has done SciPy's
// prologue (f0 in F, initial step size, Jacobian at t0 with the plane of t0 in KV[3]).  Returns 1 when the
// launch is not possible (grid not fully resident): the caller then steps under host control.
int nk2d_year_persistent(nk2d_ctx* c, double h_abs0, double newton_tol, double max_step, double n_total,
                         double* record, int64_t record_cap, int64_t* record_n) {
    if (c->kind != 0) return 1;
    const int nblk = nk2d_grid(c->ncol);
    if (!c->YR_OUT) {
        NK2D_CHECK(c, hipMalloc((void**)&c->YR_PART, sizeof(double) * 2 * c->ncol));
        NK2D_CHECK(c, hipMalloc((void**)&c->YR_OUT, sizeof(double) * 32));
        NK2D_CHECK(c, hipMalloc((void**)&c->YR_SYNC, yr_sync_bytes(c)));
        NK2D_CHECK(c, hipMalloc((void**)&c->YR_MTAB, sizeof(int) * std::max<size_t>(c->rho_tab.size(), 1)));
        NK2D_CHECK(c, hipHostMalloc((void**)&c->hYR_OUT, sizeof(double) * 32));
        NK2D_CHECK(c, hipEventCreate(&c->yr_ev[0]));
        NK2D_CHECK(c, hipEventCreate(&c->yr_ev[1]));
        c->yr_lin_tol = -1.0;
        c->yr_rec_cap = 0;
        c->YR_REC = nullptr;
    }
    const int min_sweeps = (c->min_sweeps > 1 && nk2d_has_lateral(c)) ? 2 : 1;
    if (c->yr_lin_tol != c->d.lin_tol) {
        // sweeps per shift bucket with the host's arithmetic (nk2d_sweeps_for), looked up on the device
        std::vector<int> mtab(c->rho_tab.size());
        for (size_t k = 0; k < mtab.size(); ++k)
            mtab[k] = std::max(nk2d_sweeps_for(c, c->rho_c0 * std::pow(10.0, ((double)k + 0.5) * c->rho_dlog)), min_sweeps);
        if (!mtab.empty())
            NK2D_CHECK(c, hipMemcpy(c->YR_MTAB, mtab.data(), sizeof(int) * mtab.size(), hipMemcpyHostToDevice));
        c->yr_lin_tol = c->d.lin_tol;
    }
    if (record && record_cap > c->yr_rec_cap) {
        if (c->YR_REC) NK2D_CHECK(c, hipFree(c->YR_REC));
        NK2D_CHECK(c, hipMalloc((void**)&c->YR_REC, sizeof(double) * NK2D_SCHED_WIDTH * record_cap));
        c->yr_rec_cap = record_cap;
    }
    NK2D_CHECK(c, hipMemsetAsync(c->YR_SYNC, 0, yr_sync_bytes(c), c->stream));
    YearArgs A = {};
    A.Y = c->Y; A.YOLD = c->YOLD; A.F = c->F; A.Z = c->Z; A.ZP = c->ZP; A.ZN = c->ZN; A.W = c->W;
    A.BR = c->BR; A.BCR = c->BCR; A.BCI = c->BCI;
    for (int i = 0; i < 2; ++i) { A.XR[i] = c->XR[i]; A.XCR[i] = c->XCR[i]; A.XCI[i] = c->XCI[i]; }
    A.TMP = c->TMP;
    for (int i = 0; i < 4; ++i) A.KV[i] = c->KV[i];
    fill_factor_args(c, A.fac);
    A.fac.f32 = 0;
    A.PART = c->YR_PART;
    A.t0 = c->d.t0; A.t1 = c->d.t1; A.h_abs0 = h_abs0; A.max_step = max_step; A.newton_tol = newton_tol;
    A.n_total = n_total; A.growth_cap = c->growth_cap; A.jac_fresh = c->jac_fresh; A.jac_stage = c->jac_fresh ? c->jac_stage : -1;
    for (int i = 0; i < 4; ++i) { A.bld_t[i] = c->d.bld_tvals[i]; A.bld_f[i] = c->d.bld_fvals[i]; }
    A.bldmin = c->d.bldepth_min; A.vy0 = c->d.vmix_log_shallow; A.vy1 = c->d.vmix_log_deep; A.hw = c->d.vmix_half_width;
    A.m_tab = c->YR_MTAB; A.n_tab = (int)c->rho_tab.size(); A.rho_c0 = c->rho_c0; A.rho_dlog = c->rho_dlog;
    A.arrive = (unsigned*)c->YR_SYNC; A.abort_flag = (int*)((char*)c->YR_SYNC + 4096);
    A.out = c->YR_OUT;
    A.record = record ? c->YR_REC : nullptr;
    A.record_cap = record ? record_cap : 0;
    A.fingerprint = nk2d_fingerprint(c);
    A.spin_ticks = (long long)(c->barrier_timeout_ms * 1.0e5);
    A.fences = c->year_fences;
    DevP P = make_devp(c);
    P.guard = nullptr;
    void* args[2] = {&P, &A};
    hipError_t rc = hipErrorInvalidValue;
    NK2D_CHECK(c, hipEventRecord(c->yr_ev[0], c->stream));
    {
        std::lock_guard<std::mutex> coop(coop_launch_mutex());
        NK2D_DISPATCH_E(c->E, rc = hipLaunchCooperativeKernel((const void*)k_year_persistent<EE, 0>, dim3(nblk), dim3(NK2D_BLOCK),
                                                               args, 0, c->stream));
    }
    if (rc == hipErrorCooperativeLaunchTooLarge) { (void)hipGetLastError(); return 1; }
    NK2D_CHECK(c, rc);
    NK2D_CHECK(c, hipEventRecord(c->yr_ev[1], c->stream));
    NK2D_CHECK(c, hipMemcpyAsync(c->hYR_OUT, c->YR_OUT, sizeof(double) * 32, hipMemcpyDeviceToHost, c->stream));
    NK2D_CHECK(c, hipStreamSynchronize(c->stream));
    const double* o = c->hYR_OUT;
    const int status = (int)o[0];
    // the buffers swapped roles on the device an odd or even number of times
    if ((int)o[11]) std::swap(c->Y, c->YOLD);
    {   // the three stage buffers in the roles the device left them in
        double* bufs[3] = {c->Z, c->ZP, c->ZN};
        const int code = (int)o[12];
        c->Z = bufs[code & 3]; c->ZP = bufs[(code >> 2) & 3]; c->ZN = bufs[(code >> 4) & 3];
    }
    if ((int)o[13]) std::swap(c->KV[2], c->KV[3]);
    c->st.nfev += (int64_t)o[2]; c->st.njev += (int64_t)o[3]; c->st.nlu += (int64_t)o[4];
    c->st.nsteps += (int64_t)o[5]; c->st.nrejected += (int64_t)o[6]; c->st.nnewton += (int64_t)o[7];
    c->st.nsolve += (int64_t)o[8]; c->st.nsweeps += (int64_t)o[9]; c->st.nlaunch += 1;
    float ms = 0.f;
    NK2D_CHECK(c, hipEventElapsedTime(&ms, c->yr_ev[0], c->yr_ev[1]));
    if (c->prof_every > 0) {
        // profile window = the whole-year kernel: one "launch", its algorithmic bytes, its duration
        c->prof_ms_sum += ms; c->prof_windows += 1; c->prof_cnt += 1;
        c->sweep_launches += 1; c->sweep_bytes += o[14]; c->fused_bytes_all += o[14];
    }
    // a grid barrier timed out (a co-tenant held the chip, say): the input is intact (the caller re-copies it), the year
    // reruns under host control -- counted, not failed
    if (status == 1) return 2;
    if (status == 2) return nk2d_fail(c, "Radau: step size is not finite (non-finite state or tendency)", -3);
    if (status == 3) return nk2d_fail(c, "Radau: required step size is less than spacing between numbers", -3);
    const int64_t nrec = (int64_t)o[10];
    if (record && nrec > 0) {
        const int64_t ncopy = std::min<int64_t>(nrec, record_cap);
        NK2D_CHECK(c, hipMemcpy(record, c->YR_REC, sizeof(double) * NK2D_SCHED_WIDTH * ncopy, hipMemcpyDeviceToHost));
    }
    if (record_n) *record_n = nrec;
    if (record && nrec > record_cap && record != c->own_rec.data())
        return nk2d_fail(c, "nk2d_comp_fcn: schedule record buffer too small", -4);
    return 0;
}


// =================================================================================================
// The frozen year of a small grid in ONE launch, on a schedule cache.
//
// A frozen year (nk2d_comp_fcn_frozen: the perturbed year of a finite-difference product) decides nothing, and for
// the modules whose Jacobian is a function of time alone everything but the state is known from the schedule:
// the mixing planes of every step's stage times, its Jacobian planes, its line factorisation.  Up to 208 x 208 that is
// at most a few GB per schedule, computed ONCE per schedule (= once per Newton iteration) by two batched launches over
// (step, column) -- k_cache_planes, k_cache_factor -- and read by every year of the Krylov solve.  What is left of a
// year are its simplified-Newton iterations, one phase each: k_frozen_persistent runs them all in one cooperative
// launch, a wave per column, the phases separated by the grid barrier of k_year_persistent, with the launch-per-phase
// path's own device functions (newton_fused_body; the last iteration of a step ends it: FINAL) -- bit-identical to it.
// At 26 x 26 a launch-per-phase year is 2 200 launches of 7.6 us; a phase here costs its barrier plus a microsecond.
// =================================================================================================
struct CacheRow {
    VmixArgs v;          // slots 0..2: the stage times of the row (out: its planes in the cache); slot 3: its Jacobian time
    double cre, ccr, cci;    // shifts of its line factorisation (h_lu)
};

struct CachePtrs {
    double *KV, *J;                                     // [n][3][kv_len], [n][5][np]
    double *fr_inv, *fc_invr, *fc_invi;                 // [n][nv]
    double *fr_tab, *fc_tabr, *fc_tabi;                 // [n][ncol * NK2D_TAB * 64]
    size_t kv_len, np, nv, ntab;
};

// planes and Jacobian of every row: task = (row, slot 0..3, ypos column)
template <int E>
__global__ void __launch_bounds__(NK2D_BLOCK) k_cache_planes(DevP P, const CacheRow* __restrict__ rows, CachePtrs C, int n) {
    const int lane = threadIdx.x & 63;
    const long long task = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long per_row = 4LL * P.ny;
    if (task >= per_row * n) return;
    const int i = (int)(task / per_row), rem = (int)(task - (long long)i * per_row);
    const int slot = rem / P.ny, j = rem - slot * P.ny;
    const CacheRow& R = rows[i];
    if (slot < 3) {
        double kv[E];
        vmix_body_kv<E>(P, R.v, slot * P.ny + j, lane, kv);
    } else {
        double kv[E], up[E], dn[E], so[E], no[E], ce[E];
        vmix_col_regs<E>(P, R.v.bldmin, R.v.y0, R.v.y1, R.v.hw, R.v.frac[3], j, lane, kv);
        jac_cols<E>(P, kv, j, lane, up, dn, so, no, ce);
        double* J = C.J + (size_t)i * 5 * C.np;
        store_col<E>(J, j, lane, up);
        store_col<E>(J + C.np, j, lane, dn);
        store_col<E>(J + 2 * C.np, j, lane, so);
        store_col<E>(J + 3 * C.np, j, lane, no);
        store_col<E>(J + 4 * C.np, j, lane, ce);
    }
}

// line factorisation of every row: task = (row, (system, tracer), ypos column) -- the work of k_factor per row
template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK) k_cache_factor(DevP P, const CacheRow* __restrict__ rows, CachePtrs C, int n) {
    const int lane = threadIdx.x & 63;
    const long long task = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long per_row = 2LL * P.ncol;
    if (task >= per_row * n) return;
    const int i = (int)(task / per_row), rem = (int)(task - (long long)i * per_row);
    const CacheRow& R = rows[i];
    SweepArgs A = {};
    const double* J = C.J + (size_t)i * 5 * C.np;
    A.JL = J; A.JU = J + C.np; A.JS = J + 2 * C.np; A.JN = J + 3 * C.np; A.JC = J + 4 * C.np;
    A.fr_inv = C.fr_inv + (size_t)i * C.nv; A.fc_invr = C.fc_invr + (size_t)i * C.nv; A.fc_invi = C.fc_invi + (size_t)i * C.nv;
    A.fr_tab = C.fr_tab + (size_t)i * C.ntab; A.fc_tabr = C.fc_tabr + (size_t)i * C.ntab; A.fc_tabi = C.fc_tabi + (size_t)i * C.ntab;
    A.f32 = 0;
    A.cre = R.cre; A.ccr = R.ccr; A.cci = R.cci;
    A.nreal = P.ncol; A.ntasks = 2 * P.ncol;
    factor_body<E, KIND>(P, A, rem, lane);
}

// The single-phase Newton iteration (stage + one sweep + update) of newton_fused_body for the one-launch year of a small
// grid, with EVERY operand requested before the first is used.  A wave issues in order: in the generic body the Jacobian
// planes and the factorisation are asked for behind the stage arithmetic, W again behind the solves -- four round trips
// to memory in a row, 4 of the 6 us a phase takes at one level per lane.  Here there is one.  The arithmetic is the
// generic body's, operation for operation (the same inline functions, the same expressions in the same order), so the
// results are its results bit for bit; the re-loads of y and W before the update read what the stage part read.
// KIND 0, no factorisation in the phase, double precision tables.
template <int E, int MP, int FINAL>
__device__ __forceinline__ void newton_single_body(const DevP& P, const FusedArgs& A, int task, int lane, const FinalArgs* fin = nullptr) {
    const int tr = task / P.ny, j = task - tr * P.ny;
    const int cs_col = (j > 0) ? task - 1 : task, cn_col = (j < P.ny - 1) ? task + 1 : task;
    // ---- every load
    ColCoef<E> cf;
    load_coef<E>(P, j, lane, cf);
    double y0[E], ys[E], yn[E], zc[3][E], zs[3][E], zn[3][E], kvs[3][E], w0[E], w1[E], w2[E], jl[E], ju[E];
    load_col<E, MP>(A.st.y, task, lane, y0);
    load_col<E, MP>(A.st.y, cs_col, lane, ys);
    load_col<E, MP>(A.st.y, cn_col, lane, yn);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        load_col<E, MP>(A.st.z + i * A.st.nv, task, lane, zc[i]);
        load_col<E, MP>(A.st.z + i * A.st.nv, cs_col, lane, zs[i]);
        load_col<E, MP>(A.st.z + i * A.st.nv, cn_col, lane, zn[i]);
        load_col<E>(A.st.kv[i], j, lane, kvs[i]);              // the schedule cache: constant during the launch
    }
    load_col<E>(A.st.w, task, lane, w0);
    load_col<E>(A.st.w + A.st.nv, task, lane, w1);
    load_col<E>(A.st.w + 2 * A.st.nv, task, lane, w2);
    load_col<E>(A.sw.JL, j, lane, jl);
    load_col<E>(A.sw.JU, j, lane, ju);
    double inv_r[E], tab_r[NK2D_TAB], t0[E], t1[E], tr0[NK2D_TAB], ti0[NK2D_TAB];
    load_col<E>(A.sw.fr_inv, task, lane, inv_r);
    load_tab<E>(A.sw.fr_tab, task, lane, tab_r);
    load_col<E>(A.sw.fc_invr, task, lane, t0);
    load_col<E>(A.sw.fc_invi, task, lane, t1);
    load_tab<E>(A.sw.fc_tabr, task, lane, tr0);
    load_tab<E>(A.sw.fc_tabi, task, lane, ti0);
    // ---- stage tendencies and transformed residuals
    double fr[E], fcr[E], fci[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { fr[e] = 0.0; fcr[e] = 0.0; fci[e] = 0.0; }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double c[E], cs[E], cn[E], f[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { c[e] = y0[e] + zc[i][e]; cs[e] = ys[e] + zs[i][e]; cn[e] = yn[e] + zn[i][e]; }
        tend_col<E, 0>(P, cf, c, cs, cn, kvs[i], tr, lane, f);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            fr[e] = fr[e] + f[e] * cTI[0][i];
            fcr[e] = fcr[e] + f[e] * cTI[1][i];
            fci[e] = fci[e] + f[e] * cTI[2][i];
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        fr[e] = fr[e] - A.st.mreal * w0[e];
        fcr[e] = fcr[e] - (A.st.mcr * w1[e] - A.st.mci * w2[e]);
        fci[e] = fci[e] - (A.st.mcr * w2[e] + A.st.mci * w1[e]);
    }
    // ---- the two line solves of the column (first sweep: no lateral terms)
    double a[E], cc[E];
    line_offdiag<E, 0>(P, tr, lane, jl, ju, a, cc);
#pragma unroll
    for (int e = 0; e < E; ++e) fr[e] = ((lane * E + e) < P.nz) ? fr[e] : 0.0;
    tridiag_apply<E, double>(a, cc, inv_r, tab_r, fr, lane);
    {
        cplx r[E], inv[E], tab[NK2D_TAB];
#pragma unroll
        for (int e = 0; e < E; ++e) inv[e] = c_make(t0[e], t1[e]);
#pragma unroll
        for (int i = 0; i < NK2D_TAB; ++i) tab[i] = c_make(tr0[i], ti0[i]);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool valid = (lane * E + e) < P.nz;
            r[e] = c_make(valid ? fcr[e] : 0.0, valid ? fci[e] : 0.0);
        }
        tridiag_apply<E, cplx>(a, cc, inv, tab, r, lane);
#pragma unroll
        for (int e = 0; e < E; ++e) { fcr[e] = r[e].re; fci[e] = r[e].im; }
    }
    // ---- dW = (fr, fcr, fci): norm partial, W += dW, Z = T W
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double sc = P.atol + fabs(y0[e]) * P.rtol;
        const double d0 = fr[e] / sc, d1 = fcr[e] / sc, d2 = fci[e] / sc;
        acc += (d0 * d0 + d1 * d1) + d2 * d2;
        w0[e] = w0[e] + fr[e];
        w1[e] = w1[e] + fcr[e];
        w2[e] = w2[e] + fci[e];
    }
    acc = wave_sum(acc);
    if (lane == 0) st_mp<0>(A.part + task, acc);
    double* wout = const_cast<double*>(A.st.w);
    if constexpr (FINAL) {
        double z0[E], z1[E], z2[E], ynw[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            z0[e] = (cT[0][0] * w0[e] + cT[0][1] * w1[e]) + cT[0][2] * w2[e];
            z1[e] = (cT[1][0] * w0[e] + cT[1][1] * w1[e]) + cT[1][2] * w2[e];
            z2[e] = (cT[2][0] * w0[e] + cT[2][1] * w1[e]) + cT[2][2] * w2[e];
            ynw[e] = y0[e] + z2[e];
        }
        store_col<E, MP>(fin->ynew, task, lane, ynw);
        const double xs[3] = {fin->x0, fin->x1, fin->x2};
        double o[3][E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            double q[3];
#pragma unroll
            for (int cidx = 0; cidx < 3; ++cidx) q[cidx] = (z0[e] * cP[0][cidx] + z1[e] * cP[1][cidx]) + z2[e] * cP[2][cidx];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double p1 = xs[i], p2 = p1 * xs[i], p3 = p2 * xs[i];
                double v = (q[0] * p1 + q[1] * p2) + q[2] * p3;
                v = v + y0[e];
                o[i][e] = v - ynw[e];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) store_col<E, MP>(fin->znext + i * A.st.nv, task, lane, o[i]);
        double wv[E];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int e = 0; e < E; ++e) wv[e] = (cTI[r][0] * o[0][e] + cTI[r][1] * o[1][e]) + cTI[r][2] * o[2][e];
            store_col<E>(wout + r * A.st.nv, task, lane, wv);
        }
        return;
    }
    store_col<E>(wout, task, lane, w0);
    store_col<E>(wout + A.st.nv, task, lane, w1);
    store_col<E>(wout + 2 * A.st.nv, task, lane, w2);
    double zz[E];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int e = 0; e < E; ++e) zz[e] = (cT[r][0] * w0[e] + cT[r][1] * w1[e]) + cT[r][2] * w2[e];
        store_col<E, MP>(A.st.zout + r * A.st.nv, task, lane, zz);
    }
}

struct FrozenRow {
    double mreal, mcr, mci;      // MU / h of the row
    double x0, x1, x2;           // dense-output abscissae of the NEXT row's stage times (the step-ending launch)
    int n_iter, m;               // simplified-Newton iterations, sweeps per solve
    double h;                    // step size
    int err;                     // 1: SciPy's error estimate of this step is evaluated too (its partials to row 3 i + 2 of STEP_PART)
};

struct FrozenArgs {
    double *Y, *YOLD, *Z, *ZN, *W, *F;
    double *BR, *BCR, *BCI, *XR[2], *XCR[2], *XCI[2];
    double* PART;                // scratch partials [ncol]
    double* STEP_PART;           // rows of ncol: 3 per step (last iteration, the one before, error estimate -- unused here)
    const FrozenRow* rows;
    CachePtrs C;
    int n;
    unsigned* arrive;
    int* abort_flag;
    double* out;                 // [32]: status, rows done, parities
    long long spin_ticks;
    int fences;
    unsigned* tickets;           // XCD flavour: the workgroups that find themselves on XCD 0 take a number here
    int nwg;                     // ... until this many have one
};

// XCD = 1: launched plainly with eight times the workgroups it needs (and some); a workgroup reads the XCD it landed on
// (HW_REG_XCC_ID), those on XCD 0 take a ticket, the first nwg of them are the year's workgroups, everybody else exits.
// All exchanges then stay in ONE L2: plain stores + L1-bypassing loads (MP = 2), ONE arrival counter with L2-executed adds
// -- a barrier costs 1.0-1.5 us instead of 2.1 us and a neighbour's column comes from L2 instead of the fabric
// (tools/proto_xcd_barrier.hip, profiles/r03_xcd_barrier.log).  HIP promises no placement: if XCD 0 does not get its nwg
// workgroups the barrier times out, the abort flag is raised and the caller runs the cooperative flavour (XCD = 0).
// f = fun(t, y) of the column (the plane kvp is the mixing plane at t): the tendency at a step start, for the error estimate
template <int E, int KIND, int MP>
__device__ __forceinline__ void tend_at_body(const DevP& P, const double* __restrict__ y, const double* __restrict__ kvp,
                                             double* __restrict__ f, int task, int lane) {
    const int tr = task / P.ny, j = task - tr * P.ny;
    const int cs_col = (j > 0) ? task - 1 : task, cn_col = (j < P.ny - 1) ? task + 1 : task;
    ColCoef<E> cf;
    load_coef<E>(P, j, lane, cf);
    double c[E], cs[E], cn[E], kv[E], ff[E];
    load_col<E, MP>(y, task, lane, c);
    load_col<E, MP>(y, cs_col, lane, cs);
    load_col<E, MP>(y, cn_col, lane, cn);
    load_col<E, MP>(kvp, j, lane, kv);
    tend_col<E, KIND>(P, cf, c, cs, cn, kv, tr, lane, ff);
    if constexpr (KIND == 2) forced_sources<E>(P, kvp, j, lane, c, ff);
    store_col<E, MP>(f, task, lane, ff);
}

// what the step-ending launch does behind the update (FINAL in newton_fused_body), from the stage values in memory: for
// the steps whose last Newton iteration is an ordinary one because their error estimate sits in between
template <int E, int MP>
__device__ __forceinline__ void step_tail_body(const double* __restrict__ y, const double* __restrict__ z, size_t nv,
                                               const FinalArgs& fin, double* __restrict__ wout, int task, int lane) {
    double yy[E], z0[E], z1[E], z2[E], yn[E];
    load_col<E, MP>(y, task, lane, yy);
    load_col<E, MP>(z, task, lane, z0);
    load_col<E, MP>(z + nv, task, lane, z1);
    load_col<E, MP>(z + 2 * nv, task, lane, z2);
#pragma unroll
    for (int e = 0; e < E; ++e) yn[e] = yy[e] + z2[e];
    store_col<E, MP>(fin.ynew, task, lane, yn);
    const double xs[3] = {fin.x0, fin.x1, fin.x2};
    double o[3][E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        double q[3];
#pragma unroll
        for (int cidx = 0; cidx < 3; ++cidx) q[cidx] = (z0[e] * cP[0][cidx] + z1[e] * cP[1][cidx]) + z2[e] * cP[2][cidx];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double p1 = xs[i], p2 = p1 * xs[i], p3 = p2 * xs[i];
            double v = (q[0] * p1 + q[1] * p2) + q[2] * p3;
            v = v + yy[e];
            o[i][e] = v - yn[e];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) store_col<E, MP>(fin.znext + i * nv, task, lane, o[i]);
    double wv[E];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int e = 0; e < E; ++e) wv[e] = (cTI[r][0] * o[0][e] + cTI[r][1] * o[1][e]) + cTI[r][2] * o[2][e];
        store_col<E, MP>(wout + r * nv, task, lane, wv);
    }
}

// TEAM = 1: a workgroup is ONE column, its four waves the team of newton_team_body (a stage tendency each on three of them,
// the complex system on the fourth, exchanges through LDS): the phase of a small grid is the dependent arithmetic of one
// column's Newton iteration, and the team cuts that chain (three tendencies one after the other, then the real and the
// complex solve one after the other -> one tendency, then both solves side by side).  Same arithmetic, same bits.
template <int E, int KIND, int XCD, int TEAM = 0, int NB = 0>
__global__ void __launch_bounds__(NK2D_BLOCK) k_frozen_persistent(DevP P, FrozenArgs A) {
    __shared__ int lds_ok;
    __shared__ int lds_id;
    __shared__ double team_lds[TEAM ? sizeof(TeamLds<E, 3>) / sizeof(double) : 1];
    constexpr int MPX = XCD ? 2 : 1;
    const int lane = threadIdx.x & 63;
    int wg = (int)blockIdx.x;
    if constexpr (XCD) {
        if (threadIdx.x == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            int id = -1;
            if ((xcc & 7u) == 0u) {
                const unsigned t = __hip_atomic_fetch_add(A.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t < (unsigned)A.nwg) id = (int)t;
            }
            lds_id = id;
        }
        __syncthreads();
        wg = lds_id;
        if (wg < 0) return;
    }
    const int tw = uni_i((int)(threadIdx.x >> 6));                      // TEAM: the wave's place in its team
    const int wave = TEAM ? uni_i(wg) : uni_i(wg * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6));   // the column
    const bool col_wave = wave < P.ncol;
    GridBarrier bar{A.arrive, A.abort_flag, XCD ? (unsigned)A.nwg : gridDim.x, 0u, &lds_ok, A.spin_ticks, A.fences, XCD, wg};
    // NB: neighbour-to-neighbour hand-over instead of the grid barrier.  The unit is the workgroup: one column (teams), or
    // the columns of its waves -- then the workgroup to the left matters if its first column has a left neighbour, the one to
    // the right if its last column has a right neighbour (a tracer boundary inside the workgroup needs nothing)
    int nb_left, nb_right;
    if constexpr (TEAM) {
        const int nb_j = wave % P.ny;
        nb_left = (nb_j > 0) ? wg - 1 : -1;
        nb_right = (nb_j < P.ny - 1) ? wg + 1 : -1;
    } else {
        const int wpb = (int)(blockDim.x >> 6);
        const int c0 = wg * wpb, cl = min(c0 + wpb - 1, P.ncol - 1);
        nb_left = (c0 % P.ny > 0) ? wg - 1 : -1;
        nb_right = (cl % P.ny < P.ny - 1) ? wg + 1 : -1;
    }
    NeighbourSync nbs{(unsigned*)((char*)A.arrive + 8192), A.abort_flag, wg, nb_left, nb_right, 0u, &lds_ok, A.spin_ticks, A.fences};
    const size_t nv = A.C.nv;
    int swapY = 0, swapZ = 0, status = 0, done = 0;
#define FZ_Y (swapY ? A.YOLD : A.Y)
#define FZ_YOLD (swapY ? A.Y : A.YOLD)
#define FZ_Z (swapZ ? A.ZN : A.Z)
#define FZ_ZN (swapZ ? A.Z : A.ZN)
#define FZ_SYNC() \
    if (!(NB ? nbs.sync() : bar.sync())) { status = 1; goto finish; }
    // first attempt of the year: Z0 = 0, W0 = 0 (radau.py:445-446)
    if (col_wave && (!TEAM || tw == 0)) {
        double zero[E];
#pragma unroll
        for (int e = 0; e < E; ++e) zero[e] = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            store_col<E, MPX>(FZ_Z + i * nv, wave, lane, zero);
            store_col<E, (TEAM ? MPX : 0)>(A.W + i * nv, wave, lane, zero);
        }
    }
    FZ_SYNC()
    for (int i = 0; i < A.n; ++i) {
        const FrozenRow R = A.rows[i];
        const int n_iter = uni_i(R.n_iter), m = uni_i(R.m);
        const bool last_row = i == A.n - 1;
        const double* kvb = A.C.KV + (size_t)i * 3 * A.C.kv_len;
        const double* J = A.C.J + (size_t)i * 5 * A.C.np;
        // SciPy's error estimate on this step too (every "frozen_err_check"-th: the host compares it with what the recorded
        // step was accepted with).  Three phases of their own, one wave per column: the tendency at the step start before the
        // Newton iterations, the estimate behind the last of them -- which is then an ordinary iteration --, the end of the step
        const bool with_err = uni_i(R.err) != 0 && !last_row && i > 0;
        if (with_err) {
            if (col_wave && (!TEAM || tw == 0))
                tend_at_body<E, KIND, MPX>(P, FZ_Y, A.C.KV + ((size_t)(i - 1) * 3 + 2) * A.C.kv_len, A.F, wave, lane);
            FZ_SYNC()
        }
        for (int k = 0; k < n_iter; ++k) {
            int src = 0;
            for (int it = 0; it < m; ++it) {
                const bool do_stage = it == 0, first = it == 0, do_update = it == m - 1, delta = m == 2;
                const bool is_final = do_update && k == n_iter - 1 && !last_row && !with_err;
                FusedArgs FA = {};
                FA.st.y = FZ_Y; FA.st.z = FZ_Z; FA.st.w = A.W;
                FA.st.zout = (do_stage && do_update) ? FZ_ZN : FZ_Z;
                FA.st.kv[0] = kvb; FA.st.kv[1] = kvb + A.C.kv_len; FA.st.kv[2] = kvb + 2 * A.C.kv_len;
                FA.st.br = A.BR; FA.st.bcr = A.BCR; FA.st.bci = A.BCI;
                FA.st.nv = nv; FA.st.mreal = R.mreal; FA.st.mcr = R.mcr; FA.st.mci = R.mci;
                FA.sw.JL = J; FA.sw.JU = J + A.C.np; FA.sw.JS = J + 2 * A.C.np; FA.sw.JN = J + 3 * A.C.np; FA.sw.JC = J + 4 * A.C.np;
                FA.sw.fr_inv = A.C.fr_inv + (size_t)i * nv; FA.sw.fc_invr = A.C.fc_invr + (size_t)i * nv;
                FA.sw.fc_invi = A.C.fc_invi + (size_t)i * nv;
                FA.sw.fr_tab = A.C.fr_tab + (size_t)i * A.C.ntab; FA.sw.fc_tabr = A.C.fc_tabr + (size_t)i * A.C.ntab;
                FA.sw.fc_tabi = A.C.fc_tabi + (size_t)i * A.C.ntab;
                FA.sw.f32 = 0;
                FA.sw.br = A.BR; FA.sw.bcr = A.BCR; FA.sw.bci = A.BCI;
                FA.sw.xr_old = src ? A.XR[1] : A.XR[0]; FA.sw.xcr_old = src ? A.XCR[1] : A.XCR[0];
                FA.sw.xci_old = src ? A.XCI[1] : A.XCI[0];
                FA.sw.xr_new = src ? A.XR[0] : A.XR[1]; FA.sw.xcr_new = src ? A.XCR[0] : A.XCR[1];
                FA.sw.xci_new = src ? A.XCI[0] : A.XCI[1];
                FA.sw.first = first ? 1 : 0;
                FA.part = (k == n_iter - 1) ? A.STEP_PART + (size_t)(3 * i) * P.ncol
                                            : ((k == n_iter - 2) ? A.STEP_PART + (size_t)(3 * i + 1) * P.ncol : A.PART);
                FA.do_stage = do_stage ? 1 : 0; FA.do_update = do_update ? 1 : 0; FA.delta = delta ? 1 : 0;
                if (is_final) {
                    FinalArgs Fin;
                    Fin.ynew = FZ_YOLD;
                    Fin.znext = do_stage ? FZ_ZN : FZ_Z;
                    Fin.x0 = R.x0; Fin.x1 = R.x1; Fin.x2 = R.x2;
                    Fin.nblk_cols = 0;
                    if constexpr (TEAM) {
                        if (col_wave)
                            newton_team_body<E, KIND, 0, 1, 4, 1, MPX>(P, FA, *reinterpret_cast<TeamLds<E, 3>*>(team_lds), wave, tw, lane, &Fin);
                    } else if (col_wave) {
                        bool taken = false;
                        if constexpr (KIND == 0 && E <= 2) {
                            if (m == 1) { newton_single_body<E, MPX, 1>(P, FA, wave, lane, &Fin); taken = true; }
                        }
                        if (!taken) newton_fused_body<E, KIND, 0, 1, MPX, 1>(P, FA, wave, lane, &Fin);
                    }
                    swapY ^= 1;
                    if (do_stage) swapZ ^= 1;
                } else {
                    if constexpr (TEAM) {
                        if (col_wave)
                            newton_team_body<E, KIND, 0, 1, 4, 0, MPX>(P, FA, *reinterpret_cast<TeamLds<E, 3>*>(team_lds), wave, tw, lane, nullptr);
                    } else if (col_wave) {
                        bool taken = false;
                        if constexpr (KIND == 0 && E <= 2) {
                            if (m == 1) { newton_single_body<E, MPX, 0>(P, FA, wave, lane); taken = true; }
                        }
                        if (!taken) newton_fused_body<E, KIND, 0, 1, MPX, 0>(P, FA, wave, lane);
                    }
                    if (do_stage && do_update) swapZ ^= 1;
                }
                src = 1 - src;
                FZ_SYNC()
            }
        }
        if (with_err) {
            if (col_wave && (!TEAM || tw == 0)) {
                ErrArgs EA = {};
                EA.sw.JL = J; EA.sw.JU = J + A.C.np; EA.sw.JS = J + 2 * A.C.np; EA.sw.JN = J + 3 * A.C.np; EA.sw.JC = J + 4 * A.C.np;
                EA.sw.fr_inv = A.C.fr_inv + (size_t)i * nv;
                EA.sw.fr_tab = A.C.fr_tab + (size_t)i * A.C.ntab;
                EA.sw.xr_old = A.XR[0]; EA.sw.xr_new = A.XR[1];
                EA.f = A.F; EA.z = FZ_Z; EA.y = FZ_Y; EA.nv = nv; EA.h = R.h;
                EA.part = A.STEP_PART + (size_t)(3 * i + 2) * P.ncol;
                EA.stage = 0; EA.last = 1;
                err_fused_body<E, KIND, MPX>(P, EA, wave, lane);
            }
            FZ_SYNC()
            if (col_wave && (!TEAM || tw == 0)) {
                FinalArgs Fin;
                Fin.ynew = FZ_YOLD;
                Fin.znext = FZ_ZN;
                Fin.x0 = R.x0; Fin.x1 = R.x1; Fin.x2 = R.x2;
                Fin.nblk_cols = 0;
                step_tail_body<E, MPX>(FZ_Y, FZ_Z, nv, Fin, A.W, wave, lane);
            }
            swapY ^= 1;
            swapZ ^= 1;
            FZ_SYNC()
        }
        done = i + 1;
    }
finish:
    if constexpr (NB != 0) {
        // no barrier behind the last phase: every workgroup reports a failure of its own (the host cleared `out`), the first
        // the rest -- a workgroup that gave up raised the abort flag, its neighbours give up on it in turn
        if (status != 0 && threadIdx.x == 0) A.out[0] = (double)status;
        if (wg == 0 && threadIdx.x == 0) {
            A.out[1] = (double)done; A.out[2] = (double)swapY; A.out[3] = (double)swapZ; A.out[4] = (double)nbs.phase;
        }
    } else if (wave == 0 && lane == 0 && (!TEAM || tw == 0)) {
        A.out[0] = (double)status; A.out[1] = (double)done; A.out[2] = (double)swapY; A.out[3] = (double)swapZ;
        A.out[4] = (double)bar.epoch;
    }
#undef FZ_SYNC
#undef FZ_Y
#undef FZ_YOLD
#undef FZ_Z
#undef FZ_ZN
}

// the cache of everything a schedule fixes besides the state; rebuilt when another schedule comes
struct nk2d_frozen_cache {
    uint64_t key = 0;
    int64_t n = 0;
    CachePtrs C = {};
    double* slab = nullptr;            // the one allocation the table pointers of C point into
    CacheRow* rows_dev = nullptr;      // [n]
    FrozenRow* frows_dev = nullptr;    // [n]
    size_t cap_rows = 0;
    std::vector<FrozenRow> frows;
    // a LARGE slab is allocated by a thread of its own (hipMalloc of 120 GB takes 0.03 - 3 s depending on what the process
    // holds on the host and the device): the years of the meantime run launch by launch
    std::thread alloc_thread;
    std::atomic<int> alloc_state{0};   // 0 nothing under way, 1 under way, 2 done (alloc_* valid), 3 failed
    double* alloc_slab = nullptr;
    CacheRow* alloc_rows = nullptr;
    FrozenRow* alloc_frows = nullptr;
    size_t alloc_cap = 0;
};

static uint64_t sched_key(const double* sched, int64_t n) {
    uint64_t h = 14695981039346656037ull;
    const unsigned char* p = (const unsigned char*)sched;
    const size_t nb = sizeof(double) * (size_t)n * NK2D_SCHED_WIDTH;
    for (size_t i = 0; i < nb; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h ? h : 1;
}

// 1 while a thread is allocating the slab of this context's schedule cache
int nk2d_frozen_cache_pending(const nk2d_ctx* c) {
    const nk2d_frozen_cache* fc = (const nk2d_frozen_cache*)c->frozen_cache;
    return (fc && fc->alloc_state.load() == 1) ? 1 : 0;
}

void nk2d_frozen_cache_free(nk2d_ctx* c) {
    nk2d_frozen_cache* fc = (nk2d_frozen_cache*)c->frozen_cache;
    if (!fc) return;
    if (fc->alloc_thread.joinable()) fc->alloc_thread.join();
    if (fc->alloc_state.load() == 2) {
        if (fc->alloc_slab) (void)hipFree(fc->alloc_slab);
        if (fc->alloc_rows) (void)hipFree(fc->alloc_rows);
        if (fc->alloc_frows) (void)hipFree(fc->alloc_frows);
    }
    if (fc->slab) (void)hipFree(fc->slab);
    if (fc->rows_dev) (void)hipFree(fc->rows_dev);
    if (fc->frows_dev) (void)hipFree(fc->frows_dev);
    delete fc;
    c->frozen_cache = nullptr;
}

// 0: the year ran in one launch (buffers in their roles after the last-but-one row's end; the last row's Newton iterations
//    done, its commit left to the caller);  1: not for this context / schedule (the launch-per-phase path runs);
// 2: a grid barrier timed out (the same);  < 0: error
// the instantiation for (levels per lane, module kind, flavour); the team flavour exists for one and two levels per lane
template <int E, int KIND, int XCD, int TEAM>
static hipError_t launch_frozen_one(nk2d_ctx* c, bool coop, dim3 grid, DevP& P, FrozenArgs& A) {
    if (coop) {
        void* args[2] = {&P, &A};
        if constexpr (!XCD) {
            if (c->frozen_nbsync) {
                // a wave per column with the neighbour hand-over: option "frozen_wpb" waves (= columns) to a workgroup -- the
                // waves of a workgroup move in lock step, its neighbours are the workgroups to the left and right
                const int wpb = TEAM ? NK2D_WAVES_PER_BLOCK : std::max(1, std::min(NK2D_WAVES_PER_BLOCK, c->frozen_wpb));
                const dim3 g = TEAM ? grid : dim3((unsigned)((c->ncol + wpb - 1) / wpb));
                return hipLaunchCooperativeKernel((const void*)k_frozen_persistent<E, KIND, XCD, TEAM, 1>, g, dim3(64 * wpb), args, 0, c->stream);
            }
        }
        return hipLaunchCooperativeKernel((const void*)k_frozen_persistent<E, KIND, XCD, TEAM, 0>, grid, dim3(NK2D_BLOCK), args, 0, c->stream);
    }
    hipLaunchKernelGGL((k_frozen_persistent<E, KIND, XCD, TEAM, 0>), grid, dim3(NK2D_BLOCK), 0, c->stream, P, A);
    return hipGetLastError();
}
template <int KIND, int XCD, int TEAM>
static hipError_t launch_frozen_e(nk2d_ctx* c, bool coop, dim3 grid, DevP& P, FrozenArgs& A) {
    switch (c->E) {
        case 1: return launch_frozen_one<1, KIND, XCD, TEAM>(c, coop, grid, P, A);
        case 2: return launch_frozen_one<2, KIND, XCD, TEAM>(c, coop, grid, P, A);
        // three and four levels per lane: a wave per column, cooperative flavour (all module kinds)
        case 3: if constexpr (!TEAM && !XCD) return launch_frozen_one<3, KIND, 0, 0>(c, coop, grid, P, A); else break;
        case 4: if constexpr (!TEAM && !XCD) return launch_frozen_one<4, KIND, 0, 0>(c, coop, grid, P, A); else break;
        // five to eight levels per lane (up to 512 levels): a wave per column, cooperative flavour, linear sources
        case 5: if constexpr (!TEAM && !XCD && KIND == 0) return launch_frozen_one<5, 0, 0, 0>(c, coop, grid, P, A); else break;
        case 6: if constexpr (!TEAM && !XCD && KIND == 0) return launch_frozen_one<6, 0, 0, 0>(c, coop, grid, P, A); else break;
        case 7: if constexpr (!TEAM && !XCD && KIND == 0) return launch_frozen_one<7, 0, 0, 0>(c, coop, grid, P, A); else break;
        case 8: if constexpr (!TEAM && !XCD && KIND == 0) return launch_frozen_one<8, 0, 0, 0>(c, coop, grid, P, A); else break;
        default: break;
    }
    return hipErrorInvalidValue;
}
static hipError_t launch_frozen(nk2d_ctx* c, bool xcd, bool team, bool coop, dim3 grid, DevP& P, FrozenArgs& A) {
    const bool forced = c->kind == 2;
    if (xcd) {
        if (team) return forced ? launch_frozen_e<2, 1, 1>(c, coop, grid, P, A) : launch_frozen_e<0, 1, 1>(c, coop, grid, P, A);
        return forced ? launch_frozen_e<2, 1, 0>(c, coop, grid, P, A) : launch_frozen_e<0, 1, 0>(c, coop, grid, P, A);
    }
    if (team) return forced ? launch_frozen_e<2, 0, 1>(c, coop, grid, P, A) : launch_frozen_e<0, 0, 1>(c, coop, grid, P, A);
    return forced ? launch_frozen_e<2, 0, 0>(c, coop, grid, P, A) : launch_frozen_e<0, 0, 0>(c, coop, grid, P, A);
}

int nk2d_frozen_persistent(nk2d_ctx* c, const double* sched, int64_t n, std::vector<char>* err_rows) {
    const bool needs_state = c->kind == 1 || (c->kind == 2 && c->d.sms_nrec > 0 && c->d.sink_thres > 0.0);
    if (!c->frozen_persistent || needs_state || c->hist_n != 0 || c->norm_hook || n < 1) return 1;
    // (instantiated for one to four levels per lane, and for five to eight with linear sources)
    if (c->E > c->frozen_persistent_max_e || c->E > 8 || (c->E > 4 && c->kind != 0)) return 1;
    // every row but the last must hand over to the next one (t_new == next t, whole step taken): the step-ending launch
    // predicts the next attempt from this step's collocation polynomial
    for (int64_t i = 0; i + 1 < n; ++i) {
        const double* r = sched + i * NK2D_SCHED_WIDTH;
        if (!(r[0] + r[2] == r[1] && r[NK2D_SCHED_WIDTH] == r[1] && r[NK2D_SCHED_WIDTH + 2] > 0.0 && (int)r[3] >= 1)) return 1;
    }
    if ((int)sched[(n - 1) * NK2D_SCHED_WIDTH + 3] < 1) return 1;
    const size_t ntab = (size_t)c->ncol * NK2D_TAB * 64;
    const double bytes = 8.0 * (double)n * (3.0 * c->kv_len + 5.0 * c->np + 3.0 * c->nv + 3.0 * ntab);
    if (bytes > c->frozen_cache_max_gb * 1.0e9) return 1;
    {   // a slab a thread was asked for: not there yet (launch by launch), there (adopt it), or refused (never again)
        nk2d_frozen_cache* fc = (nk2d_frozen_cache*)c->frozen_cache;
        const int st = fc ? fc->alloc_state.load() : 0;
        if (st == 1) return 1;
        if (st == 2 || st == 3) {
            if (fc->alloc_thread.joinable()) fc->alloc_thread.join();
            fc->alloc_state.store(0);
            if (st == 3) { c->frozen_persistent = 0; return 1; }
            fc->slab = fc->alloc_slab; fc->rows_dev = fc->alloc_rows; fc->frows_dev = fc->alloc_frows; fc->cap_rows = fc->alloc_cap;
            fc->alloc_slab = nullptr; fc->alloc_rows = nullptr; fc->alloc_frows = nullptr;
            fc->C.KV = nullptr;     // (the tables' places are set below)
        }
    }
    {   // ... and never more than what the device has to spare right now (other contexts of the process, other tenants)
        nk2d_frozen_cache* have = (nk2d_frozen_cache*)c->frozen_cache;
        if (!have || have->cap_rows < (size_t)n) {
            size_t free_b = 0, total_b = 0;
            NK2D_CHECK(c, hipMemGetInfo(&free_b, &total_b));
            const double held = have ? 8.0 * (double)have->cap_rows * (3.0 * c->kv_len + 5.0 * c->np + 3.0 * c->nv + 3.0 * ntab) : 0.0;
            if (1.02 * bytes > 0.85 * ((double)free_b + held)) return 1;
        }
    }
    nk2d_frozen_cache* fc = (nk2d_frozen_cache*)c->frozen_cache;
    if (!fc) { fc = new nk2d_frozen_cache(); c->frozen_cache = fc; }
    // (the rows also say which steps carry an error estimate)
    const uint64_t key = sched_key(sched, n) ^ (uint64_t)nk2d_fingerprint(c) ^ ((uint64_t)(c->frozen_err_check + 1) * 0x9E3779B97F4A7C15ull);
    if (fc->key != key || fc->n != n) {
        // option "frozen_cache_after": that many years of a schedule run launch by launch before its cache is built.  Default:
        // 0 for caches below 8 GB, 3 above.  Building a 100 GB cache takes 26 ms where a one-launch year saves 40
        // (tools/probe_cache_build.py) -- but its FIRST allocation has been seen to take 0.8 s inside a Newton run, and a Newton
        // iteration with two or three Krylov iterations has nothing to pay that back with; a long Krylov solve has.
        const int after = c->frozen_cache_after >= 0 ? c->frozen_cache_after : (bytes > 8.0e9 ? 3 : 0);
        if (c->frozen_seen_key != key) { c->frozen_seen_key = key; c->frozen_seen_years = 0; }
        if (c->frozen_seen_years++ < after) return 1;
    }
    DevP P = make_devp(c);
    P.guard = nullptr;
    if (fc->key != key || fc->n != n) {
        // ---- (re)build the cache for this schedule
        if (fc->cap_rows < (size_t)n) {
            // ONE allocation for the whole cache, with room for the longer schedules of later Newton iterates: giving 100 GB
            // back and asking for them again costs seconds (measured inside a Newton run: 4.4 s), the first request 0.03 - 0.8 s
            NK2D_CHECK(c, hipStreamSynchronize(c->stream));
            const size_t per_row = 3 * c->kv_len + 5 * c->np + 3 * c->nv + 3 * ntab;
            // (what this cache holds now is given back first -- by the thread, where a thread allocates)
            double* old_slab = fc->slab;
            CacheRow* old_rows = fc->rows_dev;
            FrozenRow* old_frows = fc->frows_dev;
            const double held_b = 8.0 * (double)fc->cap_rows * (double)per_row;
            fc->slab = nullptr; fc->rows_dev = nullptr; fc->frows_dev = nullptr; fc->cap_rows = 0;
            fc->key = 0; fc->n = 0;
            size_t cap = (size_t)n + (size_t)n / 6 + 16;
            {
                size_t free_b = 0, total_b = 0;
                NK2D_CHECK(c, hipMemGetInfo(&free_b, &total_b));
                const size_t fit = (size_t)(0.9 * ((double)free_b + held_b) / (8.0 * (double)per_row));
                cap = std::max((size_t)n, std::min(cap, fit));
            }
            const size_t slab_bytes = sizeof(double) * cap * per_row;
            const bool in_thread = slab_bytes > (size_t)8e9 && c->frozen_alloc_async;
            if (!in_thread) {
                if (old_slab) (void)hipFree(old_slab);
                if (old_rows) (void)hipFree(old_rows);
                if (old_frows) (void)hipFree(old_frows);
            }
            if (in_thread) {
                // (the new slab and tables come from the thread: launch by launch until they are there)
                if (fc->alloc_thread.joinable()) fc->alloc_thread.join();
                fc->alloc_state.store(1);
                const int dev = c->dev;
                fc->alloc_thread = std::thread([fc, dev, slab_bytes, cap, old_slab, old_rows, old_frows]() {
                    bool ok = hipSetDevice(dev) == hipSuccess;
                    if (old_slab) (void)hipFree(old_slab);
                    if (old_rows) (void)hipFree(old_rows);
                    if (old_frows) (void)hipFree(old_frows);
                    fc->alloc_slab = nullptr; fc->alloc_rows = nullptr; fc->alloc_frows = nullptr;
                    ok = ok && hipMalloc((void**)&fc->alloc_slab, slab_bytes) == hipSuccess;
                    ok = ok && hipMalloc((void**)&fc->alloc_rows, sizeof(CacheRow) * cap) == hipSuccess;
                    ok = ok && hipMalloc((void**)&fc->alloc_frows, sizeof(FrozenRow) * cap) == hipSuccess;
                    if (!ok) {
                        (void)hipGetLastError();
                        if (fc->alloc_slab) (void)hipFree(fc->alloc_slab);
                        if (fc->alloc_rows) (void)hipFree(fc->alloc_rows);
                        if (fc->alloc_frows) (void)hipFree(fc->alloc_frows);
                    }
                    fc->alloc_cap = cap;
                    fc->alloc_state.store(ok ? 2 : 3);
                });
                return 1;
            }
            NK2D_CHECK(c, hipMalloc((void**)&fc->slab, slab_bytes));
            NK2D_CHECK(c, hipMalloc((void**)&fc->rows_dev, sizeof(CacheRow) * cap));
            NK2D_CHECK(c, hipMalloc((void**)&fc->frows_dev, sizeof(FrozenRow) * cap));
            fc->cap_rows = cap;
        }
        if (fc->C.KV != fc->slab || fc->C.nv != c->nv) {   // (a new slab: the tables' places in it)
            const size_t cap = fc->cap_rows;
            double* p = fc->slab;
            fc->C.KV = p; p += cap * 3 * c->kv_len;
            fc->C.J = p; p += cap * 5 * c->np;
            fc->C.fr_inv = p; p += cap * c->nv;
            fc->C.fc_invr = p; p += cap * c->nv;
            fc->C.fc_invi = p; p += cap * c->nv;
            fc->C.fr_tab = p; p += cap * ntab;
            fc->C.fc_tabr = p; p += cap * ntab;
            fc->C.fc_tabi = p;
        }
        fc->C.kv_len = c->kv_len; fc->C.np = c->np; fc->C.nv = c->nv; fc->C.ntab = ntab;
        const double RCs[3] = {0.15505102572168222, 0.6449489742783178, 1.0};
        const double MU_REAL = 3.637834252744496, MU_CR = 2.6810828736277523, MU_CI = -3.050430199247411;
        std::vector<CacheRow> rows((size_t)n);
        fc->frows.assign((size_t)n, FrozenRow());
        for (int64_t i = 0; i < n; ++i) {
            const double* r = sched + i * NK2D_SCHED_WIDTH;
            const double t = r[0], t_new = r[1], h = r[2], t_jac = r[4], h_lu = r[5];
            CacheRow& R = rows[(size_t)i];
            double times[4];
            for (int k = 0; k < 3; ++k) times[k] = t + (h * RCs[k]);
            times[3] = t_jac;
            for (int k = 0; k < 4; ++k) {
                nk2d_host_interp(4, c->d.bld_tvals, c->d.bld_fvals, times[k], &R.v.frac[k]);
                R.v.out[k] = (k < 3) ? fc->C.KV + ((size_t)i * 3 + k) * c->kv_len : nullptr;
            }
            vmix_forcing_args(c, 3, times, R.v);
            R.v.bldmin = c->d.bldepth_min; R.v.y0 = c->d.vmix_log_shallow; R.v.y1 = c->d.vmix_log_deep; R.v.hw = c->d.vmix_half_width;
            R.cre = MU_REAL / h_lu; R.ccr = MU_CR / h_lu; R.cci = MU_CI / h_lu;
            FrozenRow& F = fc->frows[(size_t)i];
            F.mreal = MU_REAL / h; F.mcr = MU_CR / h; F.mci = MU_CI / h;
            F.n_iter = (int)r[3];
            int m_real = nk2d_sweeps_for(c, MU_REAL / h_lu), m_cplx = nk2d_sweeps_for(c, MU_CR / h_lu);
            if (c->min_sweeps > 1 && nk2d_has_lateral(c)) { m_real = std::max(m_real, 2); m_cplx = std::max(m_cplx, 2); }
            F.m = std::max(m_real, m_cplx);
            F.h = h;
            // (single-sweep solves only: the estimate is then the column's own; two-sweep rows go unsampled)
            F.err = (c->frozen_err_check > 0 && i > 0 && i + 1 < n && (i % c->frozen_err_check) == 0 && F.m == 1 && F.n_iter >= 1) ? 1 : 0;
            F.x0 = F.x1 = F.x2 = 1.0;
            if (i + 1 < n) {
                const double h2 = r[NK2D_SCHED_WIDTH + 2];
                F.x0 = ((t_new + h2 * RCs[0]) - t) / (t_new - t);
                F.x1 = ((t_new + h2 * RCs[1]) - t) / (t_new - t);
                F.x2 = ((t_new + h2 * RCs[2]) - t) / (t_new - t);
            }
        }
        NK2D_CHECK(c, hipMemcpyAsync(fc->rows_dev, rows.data(), sizeof(CacheRow) * (size_t)n, hipMemcpyHostToDevice, c->stream));
        NK2D_CHECK(c, hipMemcpyAsync(fc->frows_dev, fc->frows.data(), sizeof(FrozenRow) * (size_t)n, hipMemcpyHostToDevice, c->stream));
        NK2D_CHECK(c, hipStreamSynchronize(c->stream));    // `rows` leaves scope
        {
            const long long tasks = 4LL * c->ny * n;
            NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_cache_planes<EE>, dim3((unsigned)((tasks + 3) / 4)), dim3(NK2D_BLOCK), 0, c->stream,
                                                      P, fc->rows_dev, fc->C, (int)n));
            NK2D_CHECK(c, hipGetLastError());
            const long long ftasks = 2LL * c->ncol * n;
            NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_cache_factor<EE, KK>), dim3((unsigned)((ftasks + 3) / 4)), dim3(NK2D_BLOCK), 0,
                                                               c->stream, P, fc->rows_dev, fc->C, (int)n));
            NK2D_CHECK(c, hipGetLastError());
            c->st.nlaunch += 2;
        }
        fc->key = key;
        fc->n = n;
        c->frozen_cache_builds++;
    }
    // ---- the year
    if (!c->YR_OUT) {
        NK2D_CHECK(c, hipMalloc((void**)&c->YR_PART, sizeof(double) * 2 * c->ncol));
        NK2D_CHECK(c, hipMalloc((void**)&c->YR_OUT, sizeof(double) * 32));
        NK2D_CHECK(c, hipMalloc((void**)&c->YR_SYNC, yr_sync_bytes(c)));
        NK2D_CHECK(c, hipMalloc((void**)&c->YR_MTAB, sizeof(int) * std::max<size_t>(c->rho_tab.size(), 1)));
        NK2D_CHECK(c, hipHostMalloc((void**)&c->hYR_OUT, sizeof(double) * 32));
        NK2D_CHECK(c, hipEventCreate(&c->yr_ev[0]));
        NK2D_CHECK(c, hipEventCreate(&c->yr_ev[1]));
        c->yr_lin_tol = -1.0;
        c->yr_rec_cap = 0;
        c->YR_REC = nullptr;
    }
    NK2D_CHECK(c, hipMemsetAsync(c->YR_SYNC, 0, yr_sync_bytes(c), c->stream));
    FrozenArgs A = {};
    A.Y = c->Y; A.YOLD = c->YOLD; A.Z = c->Z; A.ZN = c->ZN; A.W = c->W; A.F = c->F;
    A.BR = c->BR; A.BCR = c->BCR; A.BCI = c->BCI;
    for (int i = 0; i < 2; ++i) { A.XR[i] = c->XR[i]; A.XCR[i] = c->XCR[i]; A.XCI[i] = c->XCI[i]; }
    A.PART = c->YR_PART;
    A.STEP_PART = c->STEP_PART;
    A.rows = fc->frows_dev;
    A.C = fc->C;
    A.n = (int)n;
    A.arrive = (unsigned*)c->YR_SYNC; A.abort_flag = (int*)((char*)c->YR_SYNC + 4096);
    A.out = c->YR_OUT;
    A.spin_ticks = (long long)(c->barrier_timeout_ms * 1.0e5);
    A.fences = c->year_fences;
    // option "frozen_team": a workgroup per column (four waves: newton_team_body) instead of a wave per column.  Measured
    // (tools/probe_frozen_persistent.py, profiles/r03_frozen_team.log): teams want a CU each -- on one XCD, two to four
    // workgroups to a CU, they lose more than they gain (26^2: 13.3 ms, 16.1 ms when LDS padding forces exactly two per CU) --
    // so they run in the cooperative flavour, where they beat the wave-per-column year on one XCD at every size: 26^2 11.4
    // against 11.8 - 12.6 ms, 30^2 12.6 / 14.2, 40^2 15.5 / 16.1, 52^2 18.5 / 20.1, 104^2 36.5 / 40.9.
    const bool team = c->frozen_team && c->E <= 2;
    const int nblk = team ? c->ncol : nk2d_grid(c->ncol);
    A.tickets = (unsigned*)((char*)c->YR_SYNC + 6144);
    A.nwg = nblk;
    const double* o = c->hYR_OUT;
    bool ran = false, timed = false;
    // ---- all of the year's workgroups on ONE XCD (option "frozen_xcd"; at most what an XCD's 32 CUs hold at once)
    // (one workgroup per CU is what the kernel's registers admit at two levels per lane: an XCD holds 32 of them at once)
    // (up to two levels per lane: beyond, the cooperative flavour with the neighbour hand-over is the faster one -- 250 x 48: 91.5 ms
    // on one XCD against 76 ms launch by launch)
    if (c->frozen_xcd && !c->frozen_xcd_failed && !team && nblk <= 28 && c->E <= 2) {
        // a workgroup that does not get its partners gives up after 20 ms (the year itself takes less than that per phase)
        A.spin_ticks = std::min<long long>(A.spin_ticks, 2000000LL);
        NK2D_CHECK(c, hipMemsetAsync(c->YR_OUT, 0, sizeof(double) * 32, c->stream));
        const dim3 grid(8 * nblk + 64);
        NK2D_CHECK(c, hipEventRecord(c->yr_ev[0], c->stream));
        NK2D_CHECK(c, launch_frozen(c, /*xcd*/ true, team, /*coop*/ false, grid, P, A));
        NK2D_CHECK(c, hipGetLastError());
        NK2D_CHECK(c, hipEventRecord(c->yr_ev[1], c->stream));
        NK2D_CHECK(c, hipMemcpyAsync(c->hYR_OUT, c->YR_OUT, sizeof(double) * 32, hipMemcpyDeviceToHost, c->stream));
        NK2D_CHECK(c, hipStreamSynchronize(c->stream));
        ran = (int)o[0] == 0 && (int64_t)o[1] == n;
        timed = true;
        if (!ran) {
            // XCD 0 did not get its workgroups (placement is not promised): not again on this context; the state is
            // where the year started only if nothing ran -- hand the year back to the caller, who restarts it
            c->frozen_xcd_failed = 1;
            return 2;
        }
        c->frozen_xcd_years++;
    }
    if (!ran) {
        A.spin_ticks = (long long)(c->barrier_timeout_ms * 1.0e5);
        hipError_t rc = hipErrorInvalidValue;
        NK2D_CHECK(c, hipMemsetAsync(c->YR_OUT, 0, sizeof(double) * 32, c->stream));
        NK2D_CHECK(c, hipEventRecord(c->yr_ev[0], c->stream));
        {
            std::lock_guard<std::mutex> coop(coop_launch_mutex());
            rc = launch_frozen(c, /*xcd*/ false, team, /*coop*/ true, dim3(nblk), P, A);
        }
        if (rc == hipErrorCooperativeLaunchTooLarge) { (void)hipGetLastError(); return 1; }
        NK2D_CHECK(c, rc);
        NK2D_CHECK(c, hipEventRecord(c->yr_ev[1], c->stream));
        timed = true;
        NK2D_CHECK(c, hipMemcpyAsync(c->hYR_OUT, c->YR_OUT, sizeof(double) * 32, hipMemcpyDeviceToHost, c->stream));
        NK2D_CHECK(c, hipStreamSynchronize(c->stream));
        if ((int)o[0] != 0 || (int64_t)o[1] != n) return 2;
    }
    if (team) c->frozen_team_years++;
    if (timed) {   // the launch itself, between two events on the context's stream (bench.py's roofline of the one-launch year)
        float ms = 0.f;
        NK2D_CHECK(c, hipEventElapsedTime(&ms, c->yr_ev[0], c->yr_ev[1]));
        c->frozen_launch_us += (int64_t)(1000.0 * (double)ms);
    }
    if ((int)o[2]) std::swap(c->Y, c->YOLD);
    if ((int)o[3]) std::swap(c->Z, c->ZN);
    // counters of the year, as the launch-per-phase path books them, and the algorithmic bytes of its phases (the formula of
    // the launches they replace; nothing of the schedule cache's one-off construction)
    for (int64_t i = 0; i < n; ++i) {
        const FrozenRow& F = fc->frows[(size_t)i];
        c->st.nnewton += F.n_iter; c->st.nfev += 3 * (int64_t)F.n_iter; c->st.nsolve += 2 * (int64_t)F.n_iter;
        c->st.nsweeps += (int64_t)F.n_iter * F.m;
        double words = 0.0;
        for (int it = 0; it < F.m; ++it) words += fused_words(c, it == 0, it == 0, it == F.m - 1, F.m == 2, false);
        c->fused_bytes_all += 8.0 * words * F.n_iter;
    }
    if (err_rows) {
        err_rows->assign((size_t)n, 0);
        for (int64_t i = 0; i < n; ++i)
            if (fc->frows[(size_t)i].err) { (*err_rows)[(size_t)i] = 1; c->st.nerr_checked++; c->st.nfev++; c->st.nsolve++; }
    }
    c->sweep_launches += 1;
    c->st.nsteps += n - 1;      // the last row's commit is the caller's
    c->st.njev += n; c->st.nlu += 2 * n;
    c->st.nlaunch += 1;
    return 0;
}
