// nk2d_kernels.hip -- model kernels of the py_driver_2d hot path for gfx950:
// layout conversion, vertical-mixing coefficient, fused advection/mixing tendency,
// Jacobian planes, line-relaxation sweeps of the shifted systems and the elementwise
// pieces of the Radau IIA step.  One wavefront owns one (tracer, ypos) column, see
// nk2d_common.h.  Compiled with -ffp-contract=off: the tendency and coefficient
// kernels keep the reference's operation order (nk_ooc/py_driver_2d/advection.py:51-76,
// horiz_mix.py:50-71, vert_mix.py:24-87, iage.py:22-41); fused multiply-adds are
// written out explicitly only inside the tridiagonal solves.
#include "nk2d_common.h"

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
    if (x > xp[n - 1]) { *out = fp[n - 1]; return 0; }
    if (x < xp[0]) { *out = fp[0]; return 0; }
    int j = 0;
    while (j + 1 < n && xp[j + 1] <= x) ++j;  // xp[j] <= x < xp[j+1]
    if (j == n - 1 || xp[j] == x) { *out = fp[j]; return 0; }
    const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    *out = slope * (x - xp[j]) + fp[j];
    return 0;
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
    int hi = 0;
    while (hi < n && xs[hi] < x) ++hi;   // searchsorted(xs, x), side = "left"
    if (hi < 1) hi = 1;
    if (hi > n - 1) hi = n - 1;
    *lo = hi - 1;
    *dx = x - xs[hi - 1];
    *den = xs[hi] - xs[hi - 1];
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

template <int E>
__device__ __forceinline__ void vmix_body(const DevP& P, const VmixArgs& A, int task, int lane) {
    const int ti = task / P.ny, j = task - ti * P.ny;
    const double frac = A.frac[ti];
    const double bld = A.bldmin + (P.BLDMAX[j] - A.bldmin) * frac;
    const double x0 = bld - A.hw, x1 = bld + A.hw;
    const double y0 = A.y0, y1 = A.y1;
    const double slope = (y1 - y0) / (x1 - x0);
    double zm0[E], zm1[E], dm[E], dmr[E], wb[E], kv[E];
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
    store_col<E>(A.out[ti], j, lane, kv);
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
template <int E>
__global__ void k_jac(DevP P, const double* __restrict__ kvp, double* __restrict__ JL, double* __restrict__ JU,
                      double* __restrict__ JS, double* __restrict__ JN, double* __restrict__ JC,
                      const double* __restrict__ ylin, double* __restrict__ UPR) {
    TASK_PROLOGUE(P.ny)
    const int j = task;
    ColCoef<E> cf;
    load_coef<E>(P, j, lane, cf);
    double kv[E], kvprev[E], up[E], dn[E], so[E], no[E], ce[E];
    load_col<E>(kvp, j, lane, kv);
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
    store_col<E>(JL, j, lane, up);
    store_col<E>(JU, j, lane, dn);
    store_col<E>(JS, j, lane, so);
    store_col<E>(JN, j, lane, no);
    store_col<E>(JC, j, lane, ce);
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
__device__ __forceinline__ void line_diag(const DevP& P, const double* __restrict__ JC, int tr, int j, int lane,
                                          double shift_re, double (&dre)[E]) {
    double jc[E], upr[E], dzr[E];
    load_col<E>(JC, j, lane, jc);
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
__global__ void __launch_bounds__(NK2D_BLOCK) k_factor(DevP P, SweepArgs A) {
    TASK_PROLOGUE(A.ntasks)
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
__global__ void __launch_bounds__(NK2D_BLOCK) k_sweep(DevP P, SweepArgs A) {
    GUARD_RETURN(P.guard)
    TASK_PROLOGUE(A.ntasks)
    const int nvar = A.ntasks / P.ny, j = task / nvar, var = task - j * nvar;
    const bool is_c = var >= A.nreal / P.ny;
    const int tr = is_c ? var - A.nreal / P.ny : var;
    const int col = tr * P.ny + j;
    double jl[E], ju[E], a[E], cc[E];
    load_col<E>(A.JL, j, lane, jl);
    load_col<E>(A.JU, j, lane, ju);
    line_offdiag<E, KIND>(P, tr, lane, jl, ju, a, cc);
    const int cs_col = (j > 0) ? col - 1 : col, cn_col = (j < P.ny - 1) ? col + 1 : col;
    double js[E], jn[E], upr[E];
    if (!A.first) {
        load_col<E>(A.JS, j, lane, js);
        load_col<E>(A.JN, j, lane, jn);
        if constexpr (KIND == 1) load_col<E>(P.UPR, j, lane, upr);
    }
    if (!is_c) {
        double r[E], inv[E], tab[NK2D_TAB];
        load_col<E>(A.br, col, lane, r);
        load_col<E>(A.fr_inv, col, lane, inv);
        load_tab<E>(A.fr_tab, col, lane, tab);
        if (!A.first) {
            double xs[E], xn[E];
            load_col<E>(A.xr_old, cs_col, lane, xs);
            load_col<E>(A.xr_old, cn_col, lane, xn);
#pragma unroll
            for (int e = 0; e < E; ++e) r[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], r[e]));
            if constexpr (KIND == 1) phos_couple<E>(P, tr, j, lane, A.xr_old, upr, r);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) r[e] = ((lane * E + e) < P.nz) ? r[e] : 0.0;
        tridiag_apply<E, double>(a, cc, inv, tab, r, lane);
        store_col<E>(A.xr_new, col, lane, r);
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
            load_col<E>(A.xcr_old, cs_col, lane, xs);
            load_col<E>(A.xcr_old, cn_col, lane, xn);
#pragma unroll
            for (int e = 0; e < E; ++e) rr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], rr[e]));
            load_col<E>(A.xci_old, cs_col, lane, xs);
            load_col<E>(A.xci_old, cn_col, lane, xn);
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
        store_col<E>(A.xcr_new, col, lane, rr);
        store_col<E>(A.xci_new, col, lane, ri);
    }
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
    // at least two sweeps whenever there is lateral coupling: with one, the stage part and the update
    // would share a launch, and the update of one column would race with the stage reads of its
    // neighbours (a single launch is only used when the columns do not couple at all, rho = 0)
    if (m < 2) m = 2;
    if (m > 400) m = 400;
    return m;
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
    double sh[NK2D_BLOCK];
    for (int t = 0; t < NK2D_BLOCK; ++t) {
        double acc = 0.0;
        for (int i = t; i < ntasks; i += NK2D_BLOCK) acc += part[i];
        sh[t] = acc;
    }
    for (int o = NK2D_BLOCK / 2; o > 0; o >>= 1)
        for (int t = 0; t < o; ++t) sh[t] += sh[t + o];
    *out = sh[0];
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

template <int E>
__device__ __forceinline__ void predict_body(const PredictArgs& A, int task, int lane) {
    const double* __restrict__ y = A.y;
    const double* __restrict__ yold = A.yold;
    const double* __restrict__ zp = A.zp;
    double* __restrict__ z = A.z;
    double* __restrict__ w = A.w;
    const size_t nv = A.nv;
    const double x0 = A.x0, x1 = A.x1, x2 = A.x2;
    double yy[E], yo[E], z0[E], z1[E], z2[E];
    load_col<E>(y, task, lane, yy);
    load_col<E>(yold, task, lane, yo);
    load_col<E>(zp, task, lane, z0);
    load_col<E>(zp + nv, task, lane, z1);
    load_col<E>(zp + 2 * nv, task, lane, z2);
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
    for (int i = 0; i < 3; ++i) store_col<E>(z + i * nv, task, lane, o[i]);
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
template <int E>
__global__ void k_attempt_setup(DevP P, VmixArgs V, int nblk_vmix, PredictArgs A) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    if ((int)blockIdx.x < nblk_vmix) {
        const int task = blockIdx.x * wpb + wave;
        if (task < P.ny * 3) vmix_body<E>(P, V, task, lane);
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
template <int E, int KIND, int FACTOR, int STAGE>
__global__ void __launch_bounds__(NK2D_BLOCK) k_newton_fused(DevP P, FusedArgs A) {
    GUARD_RETURN(P.guard)
    TASK_PROLOGUE(P.ncol)
    const int tr = task / P.ny, j = task - tr * P.ny;
    const int cs_col = (j > 0) ? task - 1 : task, cn_col = (j < P.ny - 1) ? task + 1 : task;
    double fr[E], fcr[E], fci[E];
    if (STAGE && A.do_stage) {
        ColCoef<E> cf;
        load_coef<E>(P, j, lane, cf);
        double y0[E], ys[E], yn[E];
        load_col<E>(A.st.y, task, lane, y0);
        load_col<E>(A.st.y, cs_col, lane, ys);
        load_col<E>(A.st.y, cn_col, lane, yn);
#pragma unroll
        for (int e = 0; e < E; ++e) { fr[e] = 0.0; fcr[e] = 0.0; fci[e] = 0.0; }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double c[E], cs[E], cn[E], kv[E], f[E];
            load_col<E>(A.st.z + i * A.st.nv, task, lane, c);
            load_col<E>(A.st.z + i * A.st.nv, cs_col, lane, cs);
            load_col<E>(A.st.z + i * A.st.nv, cn_col, lane, cn);
            load_col<E>(A.st.kv[i], j, lane, kv);
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
        load_col<E>(A.sw.JL, j, lane, jl);
        load_col<E>(A.sw.JU, j, lane, ju);
        line_offdiag<E, KIND>(P, tr, lane, jl, ju, a, cc);
    }
    if (!A.sw.first) {
        double js[E], jn[E], xs[E], xn[E];
        load_col<E>(A.sw.JS, j, lane, js);
        load_col<E>(A.sw.JN, j, lane, jn);
        load_col<E>(A.sw.xr_old, cs_col, lane, xs);
        load_col<E>(A.sw.xr_old, cn_col, lane, xn);
#pragma unroll
        for (int e = 0; e < E; ++e) fr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fr[e]));
        load_col<E>(A.sw.xcr_old, cs_col, lane, xs);
        load_col<E>(A.sw.xcr_old, cn_col, lane, xn);
#pragma unroll
        for (int e = 0; e < E; ++e) fcr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fcr[e]));
        load_col<E>(A.sw.xci_old, cs_col, lane, xs);
        load_col<E>(A.sw.xci_old, cn_col, lane, xn);
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
            line_diag<E, KIND>(P, A.sw.JC, tr, j, lane, A.sw.cre, dre);
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
            line_diag<E, KIND>(P, A.sw.JC, tr, j, lane, A.sw.ccr, dre);
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
        load_col<E>(A.sw.xr_old, task, lane, x1);
#pragma unroll
        for (int e = 0; e < E; ++e) fr[e] = x1[e] + fr[e];
        load_col<E>(A.sw.xcr_old, task, lane, x1);
#pragma unroll
        for (int e = 0; e < E; ++e) fcr[e] = x1[e] + fcr[e];
        load_col<E>(A.sw.xci_old, task, lane, x1);
#pragma unroll
        for (int e = 0; e < E; ++e) fci[e] = x1[e] + fci[e];
    }
    if (!A.do_update) {
        store_col<E>(A.sw.xr_new, task, lane, fr);
        store_col<E>(A.sw.xcr_new, task, lane, fcr);
        store_col<E>(A.sw.xci_new, task, lane, fci);
        return;
    }
    // dW = (fr, fcr, fci): norm partial, W += dW, Z = T W
    double yy[E], w0[E], w1[E], w2[E];
    load_col<E>(A.st.y, task, lane, yy);
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
    if (lane == 0) A.part[task] = acc;
    double* wout = const_cast<double*>(A.st.w);
    double* zout = const_cast<double*>(A.st.z);
    store_col<E>(wout, task, lane, w0);
    store_col<E>(wout + A.st.nv, task, lane, w1);
    store_col<E>(wout + 2 * A.st.nv, task, lane, w2);
    double zz[E];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int e = 0; e < E; ++e) zz[e] = (cT[r][0] * w0[e] + cT[r][1] * w1[e]) + cT[r][2] * w2[e];
        store_col<E>(zout + r * A.st.nv, task, lane, zz);
    }
}

// error estimate right-hand side  f + Z^T E / h   (radau.py:478-479)
template <int E>
__global__ void k_err_rhs(int ncol, const double* __restrict__ f, const double* __restrict__ z, size_t nv, double h,
                          double* __restrict__ out, const int* __restrict__ guard) {
    GUARD_RETURN(guard)
    TASK_PROLOGUE(ncol)
    double ff[E], z0[E], z1[E], z2[E];
    load_col<E>(f, task, lane, ff);
    load_col<E>(z, task, lane, z0);
    load_col<E>(z + nv, task, lane, z1);
    load_col<E>(z + 2 * nv, task, lane, z2);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double ze = ((z0[e] * cE[0] + z1[e] * cE[1]) + z2[e] * cE[2]) / h;
        ff[e] = ff[e] + ze;
    }
    store_col<E>(out, task, lane, ff);
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

template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK) k_err_fused(DevP P, ErrArgs A) {
    GUARD_RETURN(P.guard)
    TASK_PROLOGUE(P.ncol)
    const int tr = task / P.ny, j = task - tr * P.ny;
    double r[E], x1[E], a[E], cc[E];
    {
        double jl[E], ju[E];
        load_col<E>(A.sw.JL, j, lane, jl);
        load_col<E>(A.sw.JU, j, lane, ju);
        line_offdiag<E, KIND>(P, tr, lane, jl, ju, a, cc);
    }
    if (A.stage == 0) {
        double z0[E], z1[E], z2[E];
        load_col<E>(A.f, task, lane, r);
        load_col<E>(A.z, task, lane, z0);
        load_col<E>(A.z + A.nv, task, lane, z1);
        load_col<E>(A.z + 2 * A.nv, task, lane, z2);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const double ze = ((z0[e] * cE[0] + z1[e] * cE[1]) + z2[e] * cE[2]) / A.h;
            r[e] = r[e] + ze;
        }
    } else {
        const int cs_col = (j > 0) ? task - 1 : task, cn_col = (j < P.ny - 1) ? task + 1 : task;
        double js[E], jn[E], xs[E], xn[E];
        load_col<E>(A.sw.JS, j, lane, js);
        load_col<E>(A.sw.JN, j, lane, jn);
        load_col<E>(A.sw.xr_old, cs_col, lane, xs);
        load_col<E>(A.sw.xr_old, cn_col, lane, xn);
        load_col<E>(A.sw.xr_old, task, lane, x1);
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
    store_col<E>(A.sw.xr_new, task, lane, r);
    if (!A.last) return;
    double yy[E], z2[E];
    load_col<E>(A.y, task, lane, yy);
    load_col<E>(A.z + 2 * A.nv, task, lane, z2);
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double yn = yy[e] + z2[e];
        const double sc = P.atol + fmax(fabs(yy[e]), fabs(yn)) * P.rtol;
        const double q = r[e] / sc;
        acc += q * q;
    }
    acc = wave_sum(acc);
    if (lane == 0) A.part[task] = acc;
}

// filtered error estimate right-hand side  fun(t, y + error) + Z^T E / h  (radau.py:485-487)
template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK)
    k_err_rhs2(DevP P, const double* __restrict__ y, const double* __restrict__ err, const double* __restrict__ kvp,
               const double* __restrict__ z, size_t nv, double h, double* __restrict__ out) {
    TASK_PROLOGUE(P.ncol)
    const int tr = task / P.ny, j = task - tr * P.ny;
    const int cs_col = (j > 0) ? task - 1 : task, cn_col = (j < P.ny - 1) ? task + 1 : task;
    ColCoef<E> cf;
    load_coef<E>(P, j, lane, cf);
    double c[E], cs[E], cn[E], t0[E], kv[E], ff[E];
    load_col<E>(y, task, lane, c);
    load_col<E>(err, task, lane, t0);
#pragma unroll
    for (int e = 0; e < E; ++e) c[e] = c[e] + t0[e];
    load_col<E>(y, cs_col, lane, cs);
    load_col<E>(err, cs_col, lane, t0);
#pragma unroll
    for (int e = 0; e < E; ++e) cs[e] = cs[e] + t0[e];
    load_col<E>(y, cn_col, lane, cn);
    load_col<E>(err, cn_col, lane, t0);
#pragma unroll
    for (int e = 0; e < E; ++e) cn[e] = cn[e] + t0[e];
    load_col<E>(kvp, j, lane, kv);
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
    load_col<E>(z, task, lane, z0);
    load_col<E>(z + nv, task, lane, z1);
    load_col<E>(z + 2 * nv, task, lane, z2);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double ze = ((z0[e] * cE[0] + z1[e] * cE[1]) + z2[e] * cE[2]) / h;
        ff[e] = ff[e] + ze;
    }
    store_col<E>(out, task, lane, ff);
}

// accepted step: y_new = y + Z2 and f_new = fun(t_new, y_new) in one pass (radau.py:509-521); the
// lateral neighbours' y_new are formed on the fly, each wave stores its own column
template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK)
    k_commit_tend(DevP P, const double* __restrict__ y, const double* __restrict__ z2, const double* __restrict__ kvp,
                  double* __restrict__ ynew, double* __restrict__ f) {
    TASK_PROLOGUE(P.ncol)
    const int tr = task / P.ny, j = task - tr * P.ny;
    const int cs_col = (j > 0) ? task - 1 : task, cn_col = (j < P.ny - 1) ? task + 1 : task;
    ColCoef<E> cf;
    load_coef<E>(P, j, lane, cf);
    double c[E], cs[E], cn[E], t0[E], kv[E], ff[E];
    load_col<E>(y, task, lane, c);
    load_col<E>(z2, task, lane, t0);
#pragma unroll
    for (int e = 0; e < E; ++e) c[e] = c[e] + t0[e];
    store_col<E>(ynew, task, lane, c);
    load_col<E>(y, cs_col, lane, cs);
    load_col<E>(z2, cs_col, lane, t0);
#pragma unroll
    for (int e = 0; e < E; ++e) cs[e] = cs[e] + t0[e];
    load_col<E>(y, cn_col, lane, cn);
    load_col<E>(z2, cn_col, lane, t0);
#pragma unroll
    for (int e = 0; e < E; ++e) cn[e] = cn[e] + t0[e];
    load_col<E>(kvp, j, lane, kv);
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

// sum((err / (atol + max(|y|, |y + Z2|) rtol))^2)  (radau.py:480-481)
template <int E>
__global__ void k_err_norm(DevP P, const double* __restrict__ y, const double* __restrict__ z2p,
                           const double* __restrict__ err, double* __restrict__ part) {
    GUARD_RETURN(P.guard)
    TASK_PROLOGUE(P.ncol)
    double yy[E], z2[E], er[E];
    load_col<E>(y, task, lane, yy);
    load_col<E>(z2p, task, lane, z2);
    load_col<E>(err, task, lane, er);
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double yn = yy[e] + z2[e];
        const double sc = P.atol + fmax(fabs(yy[e]), fabs(yn)) * P.rtol;
        const double a = er[e] / sc;
        acc += a * a;
    }
    acc = wave_sum(acc);
    if (lane == 0) part[task] = acc;
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
int nk2d_r_attempt_setup(nk2d_ctx* c, const double* times, double* const* out, double x0, double x1, double x2) {
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
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_attempt_setup<EE>, dim3(nblk_vmix + nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0,
                                              c->stream, P, V, nblk_vmix, A));
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    return 0;
}
// one launch of the fused Newton iteration; src = ping-pong buffer with the previous
// sweep's iterate, the new iterate goes to 1-src unless do_update consumes it
int nk2d_r_newton_fused(nk2d_ctx* c, bool do_stage, bool first, bool do_update, double mreal, double mcr,
                        double mci, int src, bool delta) {
    FusedArgs A = {};
    A.st.y = c->Y; A.st.z = c->Z; A.st.w = c->W;
    A.st.kv[0] = c->KV[0]; A.st.kv[1] = c->KV[1]; A.st.kv[2] = c->KV[2];
    A.st.br = c->BR; A.st.bcr = c->BCR; A.st.bci = c->BCI;
    A.st.nv = c->nv; A.st.mreal = mreal; A.st.mcr = mcr; A.st.mci = mci;
    fill_factor_args(c, A.sw);
    A.sw.br = c->BR; A.sw.bcr = c->BCR; A.sw.bci = c->BCI;
    A.sw.xr_old = c->XR[src]; A.sw.xcr_old = c->XCR[src]; A.sw.xci_old = c->XCI[src];
    A.sw.xr_new = c->XR[1 - src]; A.sw.xcr_new = c->XCR[1 - src]; A.sw.xci_new = c->XCI[1 - src];
    A.sw.first = first ? 1 : 0;
    A.part = c->part_on_host ? c->hPART : c->PART;
    A.do_stage = do_stage ? 1 : 0;
    A.do_update = do_update ? 1 : 0;
    A.delta = delta ? 1 : 0;
    // pivots / PCR tables of a new (h, J): computed inside the first launch that uses them
    bool do_factor = c->factor_pending != 0;
    if (do_factor && !do_stage) {  // not expected: the first launch after an "LU" event evaluates the stages
        NK2D_TRY(nk2d_k_factor(c, true, true, c->lu_cre, c->lu_ccr, c->lu_cci));
        do_factor = false;
    }
    A.sw.cre = c->lu_cre; A.sw.ccr = c->lu_ccr; A.sw.cci = c->lu_cci;
    c->factor_pending = 0;
    DevP P = make_devp(c);
    {
        // algorithmic (unique) bytes of this launch, P = nz*ny cells, N = tc*P values:
        //   stage : read y, Z[3], W[3] (7N), kappa_v at 3 times + 4 static planes (7P),
        //           write the 3 right-hand sides (3N) unless the update consumes them
        //   sweep : Jacobian planes JL, JU (+JS, JN after the first sweep), pivot reciprocals
        //           (real N + complex 2N), PCR tables (3 * 14/E * N), right-hand sides (3N, unless
        //           just computed), previous iterate (3N, after the first sweep), new iterate (3N,
        //           unless the update consumes it)
        //   update: y (N, unless the stage read it), W read + write (6N), Z write (3N)
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
        c->sweep_launches++;
        c->fused_bytes_all += 8.0 * words;
        if (c->win_open) {
            c->win_launches++;
            c->win_bytes += 8.0 * words;
        }
    }
    if (do_factor) {  // always a launch with the stage part
        NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_fused<EE, KK, 1, 1>), dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream, P, A));
    } else if (do_stage || c->kind != 1) {
        NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_newton_fused<EE, KK, 0, 1>), dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream, P, A));
    } else {          // phosphorus, sweep-only launch: the lean instantiation
        NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL((k_newton_fused<EE, 1, 0, 0>), dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, c->stream, P, A));
    }
    NK2D_CHECK(c, hipGetLastError());
    c->st.nlaunch++;
    c->st.nsweeps++;
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
