// nk2d_bodies.h -- the device functions of the py_driver_2d hot path for gfx950, shared by every translation unit that
// launches them: the per-phase kernels (nk2d_kernels.hip), the one-launch years (nk2d_year.hip, nk2d_frozen.hip) and the
// command stream (nk2d_stream.hip).  One wavefront owns one (tracer, ypos) column, see nk2d_common.h.  Every kernel is a
// thin wrapper around the bodies here, so that all flavours of a phase compute the same bits.  Compiled with
// -ffp-contract=off: the tendency and coefficient functions keep the reference's operation order
// (nk_ooc/py_driver_2d/advection.py:51-76, horiz_mix.py:50-71, vert_mix.py:24-87, iage.py:22-41); fused multiply-adds are
// written out explicitly only inside the tridiagonal solves.
#pragma once
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "nk2d_common.h"
#include "nk2d_hostmath.h"


struct DevP {
    int nz, ny, tc, ncol;
    const double *VV, *KH, *WT, *WB, *DZR, *ZM0, *ZM1, *DM, *DMR, *DYR, *BLDMAX;
    double surf[NK2D_MAX_TRACERS], starget[NK2D_MAX_TRACERS], decay[NK2D_MAX_TRACERS], csrc;
    double atol, rtol;
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

static inline DevP make_devp(const nk2d_ctx* c) {
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
static __constant__ double cTI[3][3] = {
    {4.17871859155190428, 0.32768282076106237, 0.52337644549944951},
    {-4.17871859155190428, -0.32768282076106237, 0.47662355450055044},
    {0.50287263494578682, -2.57192694985560522, 0.59603920482822492}};
static __constant__ double cT[3][3] = {
    {0.09443876248897524, -0.14125529502095421, 0.03002919410514742},
    {0.25021312296533332, 0.20412935229379994, -0.38294211275726192},
    {1.0, 1.0, 0.0}};
static __constant__ double cP[3][3] = {
    {10.048809399827414, -25.62959144707664, 15.580782047249224},
    {-1.382142733160748, 10.296258113743303, -8.914115380582556},
    {0.3333333333333333, -2.6666666666666665, 3.3333333333333335}};
static __constant__ double cE[3] = {-10.048809399827414, 1.382142733160748, -0.3333333333333333};

#define TASK_PROLOGUE(ntasks)                                              \
    const int lane = threadIdx.x & 63;                                     \
    const int task = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); \
    if (task >= (ntasks)) return;


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
static inline void forcing_bracket(int n, const double* xs, double x, int* lo, double* dx, double* den) {
    nk2d_hm_bracket(n, xs, x, lo, dx, den);
}
static inline void vmix_forcing_args(const nk2d_ctx* c, int nt, const double* times, VmixArgs& A) {
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

// The same from LDS: a wave of a resident kernel owns its column for a whole year and the coefficients are static -- 7 E + 1
// values per lane, 25 KB per wave at seven levels per lane, fetched from L2 again in every phase otherwise (an eighth of a
// phase's bytes, a fifth of its load instructions, all of them ahead of the first tendency)
#define NK2D_COEF_LDS_DOUBLES(E) ((7 * (E) + 1) * 64)
template <int E>
__device__ __forceinline__ void store_coef_lds(double* s, int lane, const ColCoef<E>& cf) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        s[(0 * E + e) * 64 + lane] = cf.vS[e]; s[(1 * E + e) * 64 + lane] = cf.vN[e]; s[(2 * E + e) * 64 + lane] = cf.khS[e];
        s[(3 * E + e) * 64 + lane] = cf.khN[e]; s[(4 * E + e) * 64 + lane] = cf.wT[e]; s[(5 * E + e) * 64 + lane] = cf.wB[e];
        s[(6 * E + e) * 64 + lane] = cf.dzr[e];
    }
    s[7 * E * 64 + lane] = cf.dyr;
}
template <int E>
__device__ __forceinline__ void load_coef_lds(const double* s, int lane, ColCoef<E>& cf) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        cf.vS[e] = s[(0 * E + e) * 64 + lane]; cf.vN[e] = s[(1 * E + e) * 64 + lane]; cf.khS[e] = s[(2 * E + e) * 64 + lane];
        cf.khN[e] = s[(3 * E + e) * 64 + lane]; cf.wT[e] = s[(4 * E + e) * 64 + lane]; cf.wB[e] = s[(5 * E + e) * 64 + lane];
        cf.dzr[e] = s[(6 * E + e) * 64 + lane];
    }
    cf.dyr = s[7 * E * 64 + lane];
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

static inline void fill_factor_args(const nk2d_ctx* c, SweepArgs& A) {
    A.JL = c->JL; A.JU = c->JU; A.JS = c->JS; A.JN = c->JN; A.JC = c->JC;
    A.fr_inv = c->FR_INV; A.fc_invr = c->FC_INVR; A.fc_invi = c->FC_INVI;
    A.fr_tab = c->FR_TAB; A.fc_tabr = c->FC_TABR; A.fc_tabi = c->FC_TABI;
    A.fr_inv32 = c->FR32_INV; A.fc_invr32 = c->FC32_INVR; A.fc_invi32 = c->FC32_INVI;
    A.fr_tab32 = c->FR32_TAB; A.fc_tabr32 = c->FC32_TABR; A.fc_tabi32 = c->FC32_TABI;
    A.f32 = c->factor_fp32;
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

// start of a step attempt in one launch: the vertical mixing planes at the three stage times
// (first blocks) and the predicted stage values (remaining blocks) are independent of each other
struct JacOut {
    double *JL, *JU, *JS, *JN, *JC;
    int stage;   // >= 0: the waves computing the plane of this stage time also derive the Jacobian planes from it
};

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
// W of a column in LDS ([3][E][64] doubles behind the coefficients): W is the wave's own, read and rewritten by every Newton
// iteration and by nothing else -- a fifth of an iteration's bytes that need not leave the compute unit while a resident kernel runs
template <int E>
__device__ __forceinline__ void w_lds_get(const double* s, int r, int lane, double (&v)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = s[(r * E + e) * 64 + lane];
}
template <int E>
__device__ __forceinline__ void w_lds_put(double* s, int r, int lane, const double (&v)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) s[(r * E + e) * 64 + lane] = v[e];
}

// What a column's wave finds in LDS inside a resident kernel (CL, a bit mask):
//   bit 0  coef: the static coefficients of the column (load_coef_lds)
//   bit 1  w   : W of the column lives there (w_lds_get / _put)
//   bit 2  step: what is constant over the Newton iterations of a STEP and the same for every tracer of the ypos column -- the
//                three mixing columns of the stage times, then the vertical Jacobian diagonals JL, JU ([5][E][64] doubles)
//   bit 3  piv : the pivot reciprocals of the column's real system ([E][64])
struct LdsSrc {
    const double* coef;
    double* w;
    const double* step;
    const double* piv;
};
template <int E>
__device__ __forceinline__ void col_lds_get(const double* s, int r, int lane, double (&v)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = s[(r * E + e) * 64 + lane];
}
template <int E, int KIND, int FACTOR, int STAGE, int MP = 0, int FINAL = 0, int CL = 0>
__device__ __forceinline__ void newton_fused_body(const DevP& P, const FusedArgs& A, int task, int lane,
                                                  const FinalArgs* fin = nullptr, const LdsSrc* lds = nullptr) {
    const double* coef_lds = (CL & 1) ? lds->coef : nullptr;
    double* w_lds = (CL & 2) ? lds->w : nullptr;
    (void)coef_lds; (void)w_lds;
    const int tr = task / P.ny, j = task - tr * P.ny;
    const int cs_col = (j > 0) ? task - 1 : task, cn_col = (j < P.ny - 1) ? task + 1 : task;
    double fr[E], fcr[E], fci[E];
    if (STAGE && A.do_stage) {
        ColCoef<E> cf;
        if constexpr (CL & 1) load_coef_lds<E>(coef_lds, lane, cf);       // (the wave's own column: stored there at kernel entry)
        else load_coef<E>(P, j, lane, cf);
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
            if constexpr (CL & 4) col_lds_get<E>(lds->step, i, lane, kv);
            else load_col<E, MP>(A.st.kv[i], j, lane, kv);
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
        if constexpr (CL & 2) {
            w_lds_get<E>(w_lds, 0, lane, w0); w_lds_get<E>(w_lds, 1, lane, w1); w_lds_get<E>(w_lds, 2, lane, w2);
        } else {
            load_col<E>(A.st.w, task, lane, w0);
            load_col<E>(A.st.w + A.st.nv, task, lane, w1);
            load_col<E>(A.st.w + 2 * A.st.nv, task, lane, w2);
        }
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
        if constexpr (CL & 4) {
            col_lds_get<E>(lds->step, 3, lane, jl);
            col_lds_get<E>(lds->step, 4, lane, ju);
        } else {
            load_col<E, MP>(A.sw.JL, j, lane, jl);
            load_col<E, MP>(A.sw.JU, j, lane, ju);
        }
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
            if constexpr (CL & 8) col_lds_get<E>(lds->piv, 0, lane, inv);
            else load_col<E>(A.sw.fr_inv, task, lane, inv);
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
    if constexpr (CL & 2) {
        w_lds_get<E>(w_lds, 0, lane, w0); w_lds_get<E>(w_lds, 1, lane, w1); w_lds_get<E>(w_lds, 2, lane, w2);
    } else {
        load_col<E>(A.st.w, task, lane, w0);
        load_col<E>(A.st.w + A.st.nv, task, lane, w1);
        load_col<E>(A.st.w + 2 * A.st.nv, task, lane, w2);
    }
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
            if constexpr (CL & 2) w_lds_put<E>(w_lds, r, lane, wv);
            else store_col<E>(wout + r * A.st.nv, task, lane, wv);
        }
        return;
    }
    double* zout = A.st.zout;
    if constexpr (CL & 2) {
        w_lds_put<E>(w_lds, 0, lane, w0); w_lds_put<E>(w_lds, 1, lane, w1); w_lds_put<E>(w_lds, 2, lane, w2);
    } else {
        store_col<E>(wout, task, lane, w0);
        store_col<E>(wout + A.st.nv, task, lane, w1);
        store_col<E>(wout + 2 * A.st.nv, task, lane, w2);
    }
    double zz[E];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int e = 0; e < E; ++e) zz[e] = (cT[r][0] * w0[e] + cT[r][1] * w1[e]) + cT[r][2] * w2[e];
        store_col<E, MP>(zout + r * A.st.nv, task, lane, zz);
    }
}


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

// --- host wrappers used by the Radau driver --------------------------------------
static inline PredictArgs predict_args(nk2d_ctx* c, double x0, double x1, double x2) {
    PredictArgs A;
    A.y = c->Y; A.yold = c->YOLD; A.zp = c->ZP; A.z = c->Z; A.w = c->W;
    A.nv = c->nv;
    A.x0 = x0; A.x1 = x1; A.x2 = x2;
    return A;
}
// algorithmic (unique) 8-byte words of one launch of the fused Newton iteration, P = nz*ny cells, N = tc*P values:
//   stage : read y, Z[3], W[3] (7N), kappa_v at 3 times + 4 static planes (7P),
//           write the 3 right-hand sides (3N) unless the update consumes them
//   sweep : Jacobian planes JL, JU (+JS, JN after the first sweep), pivot reciprocals
//           (real N + complex 2N), PCR tables (3 * 14/E * N), right-hand sides (3N, unless
//           just computed), previous iterate (3N, after the first sweep), new iterate (3N,
//           unless the update consumes it)
//   update: y (N, unless the stage read it), W read + write (6N), Z write (3N)
static inline double fused_words(const nk2d_ctx* c, bool do_stage, bool first, bool do_update, bool delta, bool do_factor) {
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

static inline void fill_fused_args(nk2d_ctx* c, FusedArgs& A, bool do_stage, bool first, bool do_update, double mreal,
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

// =================================================================================================
// Resident kernels (the one-launch frozen year, nk2d_frozen.hip; the command stream, nk2d_stream.hip): a wave owns its
// (tracer, ypos) column for as long as the kernel runs, the phases are the device functions the per-phase kernels call.
//
// Visibility between workgroups follows the hand-off the guides validate for gfx950 (MI355X_MICROARCH.md, inter-workgroup
// visibility, table row 1): every array another workgroup may read is stored write-through and loaded L1-bypassing (MP = 1
// accessors: relaxed agent-scope atomics = sc1); before a hand-over every wave drains its stores (s_waitcnt vmcnt(0)), the
// workgroup joins, ONE lane publishes and polls; the others wait at the workgroup barrier behind that lane.  Arrays only ever
// touched by their owning wave (right-hand sides, the line factorisation, F) stay plain -- or live in LDS.  Every spin is
// bounded by time; a timeout raises an abort flag every workgroup sees at its next hand-over.
// =================================================================================================
#define NK2D_SPIN_LIMIT 4000000

// Synchronisation block of the one-launch frozen year: abort flag at 4096, from 8192 one 128-byte line per workgroup
// for NeighbourSync.
static inline size_t yr_sync_bytes(const nk2d_ctx* c) { return 8192 + (size_t)c->ncol * 128; }

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

// values every lane of every wave holds identically: tell the compiler (scalar registers, uniform branches)
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool uni_b(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
__device__ __forceinline__ double uni_d(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}



// =================================================================================================
// The frozen year of a small grid in ONE launch, on a schedule cache.
//
// A frozen year (nk2d_comp_fcn_frozen: the perturbed year of a finite-difference product) decides nothing, and for
// the modules whose Jacobian is a function of time alone everything but the state is known from the schedule:
// the mixing planes of every step's stage times, its Jacobian planes, its line factorisation.  Up to 208 x 208 that is
// at most a few GB per schedule, computed ONCE per schedule (= once per Newton iteration) by two batched launches over
// (step, column) -- k_cache_planes, k_cache_factor -- and read by every year of the Krylov solve (at 416 x 416: 102 GB, the
// cache is what the 288 GB are for).  What is left of a year are its simplified-Newton iterations, one phase each:
// k_frozen_persistent runs them all in one launch whose workgroups are all resident, a wave or a four-wave team per
// column, workgroups handing over to their lateral neighbours between phases, with the launch-per-phase path's own device
// functions (newton_fused_body; the last iteration of a step ends it: FINAL) -- bit-identical to it.
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
    int coef_lds;                // option "frozen_coef_lds" (bits of LdsSrc): what a wave finds in LDS; bits 2, 3 need `by_column`
    int by_column;               // 1: a workgroup is ONE ypos column with all its tracers (a wave each) instead of adjacent columns of one tracer
};

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

