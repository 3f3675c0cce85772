// proto_two_wave.hip -- stand-alone experiment for the next round (not part of libnk2d.so):
// does the middle sweep of the line relaxation (real + complex system of every column:
// factor tables, right-hand sides and lateral neighbours read, two tridiagonal solves, new
// iterate written) run faster with TWO waves per column?
//
//   variant A: the product layout -- one wave per column, E = 7 levels per lane (nz = 416),
//              block-end system solved by PCR with wave shuffles (tridiag_apply of nk2d_common.h)
//   variant B: 128 "lanes" (two waves of one workgroup) per column, E = 4 levels per lane,
//              PCR over the 128 block ends through LDS
//
// Both variants solve the same 832 x 2 systems; the program checks that they agree and prints the
// time per launch of each.  Build and run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I include tools/proto_two_wave.hip -o /tmp/proto && /tmp/proto
#include "../newton-krylov_ooc_amd/csrc/nk2d_common.h"

#include <cmath>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } \
    } while (0)

constexpr int NZ = 416, NY = 416, TC = 2, NCOL = TC * NY;
constexpr int EA = 7;            // variant A: levels per lane
constexpr int EB = 4, LB = 128;  // variant B: levels per lane, lanes per column
constexpr int TAB2 = 16;         // K1[7], K2[7], IB, G

// ---------------------------------------------------------------------------------
// variant A kernels (product primitives)
// ---------------------------------------------------------------------------------
struct ArgsA {
    const double *a, *c, *dre;                 // [NY] planes, packed A layout (shared by the tracers)
    double dim;                                // imaginary part of the complex diagonal
    double *inv_r, *tab_r, *inv_cr, *inv_ci, *tab_cr, *tab_ci;
    const double *js, *jn;                     // lateral couplings, [NY] planes
    const double *br, *bcr, *bci, *xr, *xcr, *xci;
    double *yr, *ycr, *yci;
};

__global__ void __launch_bounds__(256) kA_factor(ArgsA A) {
    const int lane = threadIdx.x & 63, task = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= NCOL) return;
    const int j = task % NY;
    double a[EA], c[EA], d[EA];
    load_col<EA>(A.a, j, lane, a);
    load_col<EA>(A.c, j, lane, c);
    load_col<EA>(A.dre, j, lane, d);
    double inv[EA], tab[NK2D_TAB];
    tridiag_factor<EA, double>(a, c, d, inv, tab, lane);
    store_col<EA>(A.inv_r, task, lane, inv);
    for (int i = 0; i < NK2D_TAB; ++i) A.tab_r[((size_t)task * NK2D_TAB + i) * 64 + lane] = tab[i];
    cplx dc[EA], invc[EA], tabc[NK2D_TAB];
    for (int e = 0; e < EA; ++e) dc[e] = c_make(d[e], ((lane * EA + e) < NZ) ? A.dim : 0.0);
    tridiag_factor<EA, cplx>(a, c, dc, invc, tabc, lane);
    double re[EA], im[EA];
    for (int e = 0; e < EA; ++e) { re[e] = invc[e].re; im[e] = invc[e].im; }
    store_col<EA>(A.inv_cr, task, lane, re);
    store_col<EA>(A.inv_ci, task, lane, im);
    for (int i = 0; i < NK2D_TAB; ++i) {
        A.tab_cr[((size_t)task * NK2D_TAB + i) * 64 + lane] = tabc[i].re;
        A.tab_ci[((size_t)task * NK2D_TAB + i) * 64 + lane] = tabc[i].im;
    }
}

__global__ void __launch_bounds__(256) kA_sweep(ArgsA A) {
    const int lane = threadIdx.x & 63, task = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= NCOL) return;
    const int j = task % NY;
    const int cs = (j > 0) ? task - 1 : task, cn = (j < NY - 1) ? task + 1 : task;
    double a[EA], c[EA], js[EA], jn[EA], fr[EA], fcr[EA], fci[EA], xs[EA], xn[EA];
    load_col<EA>(A.a, j, lane, a);
    load_col<EA>(A.c, j, lane, c);
    load_col<EA>(A.js, j, lane, js);
    load_col<EA>(A.jn, j, lane, jn);
    load_col<EA>(A.br, task, lane, fr);
    load_col<EA>(A.bcr, task, lane, fcr);
    load_col<EA>(A.bci, task, lane, fci);
    load_col<EA>(A.xr, cs, lane, xs); load_col<EA>(A.xr, cn, lane, xn);
    for (int e = 0; e < EA; ++e) fr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fr[e]));
    load_col<EA>(A.xcr, cs, lane, xs); load_col<EA>(A.xcr, cn, lane, xn);
    for (int e = 0; e < EA; ++e) fcr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fcr[e]));
    load_col<EA>(A.xci, cs, lane, xs); load_col<EA>(A.xci, cn, lane, xn);
    for (int e = 0; e < EA; ++e) fci[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fci[e]));
    {
        double inv[EA], tab[NK2D_TAB];
        load_col<EA>(A.inv_r, task, lane, inv);
        for (int i = 0; i < NK2D_TAB; ++i) tab[i] = A.tab_r[((size_t)task * NK2D_TAB + i) * 64 + lane];
        for (int e = 0; e < EA; ++e) fr[e] = ((lane * EA + e) < NZ) ? fr[e] : 0.0;
        tridiag_apply<EA, double>(a, c, inv, tab, fr, lane);
    }
    {
        cplx r[EA], inv[EA], tab[NK2D_TAB];
        double t0[EA], t1[EA];
        load_col<EA>(A.inv_cr, task, lane, t0);
        load_col<EA>(A.inv_ci, task, lane, t1);
        for (int e = 0; e < EA; ++e) {
            const bool valid = (lane * EA + e) < NZ;
            inv[e] = c_make(t0[e], t1[e]);
            r[e] = c_make(valid ? fcr[e] : 0.0, valid ? fci[e] : 0.0);
        }
        for (int i = 0; i < NK2D_TAB; ++i)
            tab[i] = c_make(A.tab_cr[((size_t)task * NK2D_TAB + i) * 64 + lane], A.tab_ci[((size_t)task * NK2D_TAB + i) * 64 + lane]);
        tridiag_apply<EA, cplx>(a, c, inv, tab, r, lane);
        for (int e = 0; e < EA; ++e) { fcr[e] = r[e].re; fci[e] = r[e].im; }
    }
    store_col<EA>(A.yr, task, lane, fr);
    store_col<EA>(A.ycr, task, lane, fcr);
    store_col<EA>(A.yci, task, lane, fci);
}

// variant A0: the loads and stores of variant A with no tridiagonal solve (memory floor of the launch)
__global__ void __launch_bounds__(256) kA0_sweep(ArgsA A) {
    const int lane = threadIdx.x & 63, task = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= NCOL) return;
    const int j = task % NY;
    const int cs = (j > 0) ? task - 1 : task, cn = (j < NY - 1) ? task + 1 : task;
    double a[EA], c[EA], js[EA], jn[EA], fr[EA], fcr[EA], fci[EA], xs[EA], xn[EA], t0[EA];
    load_col<EA>(A.a, j, lane, a);
    load_col<EA>(A.c, j, lane, c);
    load_col<EA>(A.js, j, lane, js);
    load_col<EA>(A.jn, j, lane, jn);
    load_col<EA>(A.br, task, lane, fr);
    load_col<EA>(A.bcr, task, lane, fcr);
    load_col<EA>(A.bci, task, lane, fci);
    load_col<EA>(A.xr, cs, lane, xs); load_col<EA>(A.xr, cn, lane, xn);
    for (int e = 0; e < EA; ++e) fr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fr[e])) + a[e];
    load_col<EA>(A.xcr, cs, lane, xs); load_col<EA>(A.xcr, cn, lane, xn);
    for (int e = 0; e < EA; ++e) fcr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fcr[e])) + c[e];
    load_col<EA>(A.xci, cs, lane, xs); load_col<EA>(A.xci, cn, lane, xn);
    for (int e = 0; e < EA; ++e) fci[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fci[e]));
    load_col<EA>(A.inv_r, task, lane, t0);
    for (int e = 0; e < EA; ++e) fr[e] += t0[e];
    load_col<EA>(A.inv_cr, task, lane, t0);
    for (int e = 0; e < EA; ++e) fcr[e] += t0[e];
    load_col<EA>(A.inv_ci, task, lane, t0);
    for (int e = 0; e < EA; ++e) fci[e] += t0[e];
    double acc = 0.0;
    for (int i = 0; i < NK2D_TAB; ++i)
        acc += A.tab_r[((size_t)task * NK2D_TAB + i) * 64 + lane] + A.tab_cr[((size_t)task * NK2D_TAB + i) * 64 + lane] +
               A.tab_ci[((size_t)task * NK2D_TAB + i) * 64 + lane];
    fr[0] += acc;
    store_col<EA>(A.yr, task, lane, fr);
    store_col<EA>(A.ycr, task, lane, fcr);
    store_col<EA>(A.yci, task, lane, fci);
}

__global__ void k_empty(int) {}

// variant A0w: the same traffic as A0 with 16-byte loads / stores per lane (levels stored in pairs:
// slot ((col * 4 + e / 2) * 64 + lane) * 2 + (e & 1), one padding level per lane for E = 7).  The data are
// not re-laid out for this timing-only variant: the byte count per column is what matters.
__device__ __forceinline__ void load_w(const double* base, size_t col, int lane, double (&o)[8]) {
    const double2* p = reinterpret_cast<const double2*>(base + col * (size_t)(EA * 64)) + lane;
    // 3.5 double2 per lane cover the 7 x 64 doubles of a column: read 3 full pairs and one half pair
    for (int q = 0; q < 3; ++q) { double2 v = p[q * 64]; o[2 * q] = v.x; o[2 * q + 1] = v.y; }
    o[6] = base[col * (size_t)(EA * 64) + 6 * 64 + lane];
    o[7] = 0.0;
}
__device__ __forceinline__ void store_w(double* base, size_t col, int lane, const double (&v)[8]) {
    double2* p = reinterpret_cast<double2*>(base + col * (size_t)(EA * 64)) + lane;
    for (int q = 0; q < 3; ++q) p[q * 64] = make_double2(v[2 * q], v[2 * q + 1]);
    base[col * (size_t)(EA * 64) + 6 * 64 + lane] = v[6];
}
__global__ void __launch_bounds__(256) kA0w_sweep(ArgsA A) {
    const int lane = threadIdx.x & 63, task = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= NCOL) return;
    const int j = task % NY;
    const int cs = (j > 0) ? task - 1 : task, cn = (j < NY - 1) ? task + 1 : task;
    double a[8], c[8], js[8], jn[8], fr[8], fcr[8], fci[8], xs[8], xn[8], t0[8];
    load_w(A.a, j, lane, a);
    load_w(A.c, j, lane, c);
    load_w(A.js, j, lane, js);
    load_w(A.jn, j, lane, jn);
    load_w(A.br, task, lane, fr);
    load_w(A.bcr, task, lane, fcr);
    load_w(A.bci, task, lane, fci);
    load_w(A.xr, cs, lane, xs); load_w(A.xr, cn, lane, xn);
    for (int e = 0; e < 7; ++e) fr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fr[e])) + a[e];
    load_w(A.xcr, cs, lane, xs); load_w(A.xcr, cn, lane, xn);
    for (int e = 0; e < 7; ++e) fcr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fcr[e])) + c[e];
    load_w(A.xci, cs, lane, xs); load_w(A.xci, cn, lane, xn);
    for (int e = 0; e < 7; ++e) fci[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fci[e]));
    load_w(A.inv_r, task, lane, t0);
    for (int e = 0; e < 7; ++e) fr[e] += t0[e];
    load_w(A.inv_cr, task, lane, t0);
    for (int e = 0; e < 7; ++e) fcr[e] += t0[e];
    load_w(A.inv_ci, task, lane, t0);
    for (int e = 0; e < 7; ++e) fci[e] += t0[e];
    double acc = 0.0;
    const double2* tr = reinterpret_cast<const double2*>(A.tab_r + (size_t)task * NK2D_TAB * 64) + lane;
    const double2* tcr = reinterpret_cast<const double2*>(A.tab_cr + (size_t)task * NK2D_TAB * 64) + lane;
    const double2* tci = reinterpret_cast<const double2*>(A.tab_ci + (size_t)task * NK2D_TAB * 64) + lane;
    for (int i = 0; i < NK2D_TAB / 2; ++i) {
        double2 u = tr[i * 64], v = tcr[i * 64], w = tci[i * 64];
        acc += (u.x + u.y) + (v.x + v.y) + (w.x + w.y);
    }
    fr[0] += acc;
    store_w(A.yr, task, lane, fr);
    store_w(A.ycr, task, lane, fcr);
    store_w(A.yci, task, lane, fci);
}

// variant A2: variant A with the real and the complex solve advanced side by side (two independent
// dependency chains in every loop body) instead of one after the other
__global__ void __launch_bounds__(256) kA2_sweep(ArgsA A) {
    const int lane = threadIdx.x & 63, task = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= NCOL) return;
    const int j = task % NY;
    const int cs = (j > 0) ? task - 1 : task, cn = (j < NY - 1) ? task + 1 : task;
    double a[EA], c[EA], js[EA], jn[EA], fr[EA], fcr[EA], fci[EA], xs[EA], xn[EA];
    load_col<EA>(A.a, j, lane, a);
    load_col<EA>(A.c, j, lane, c);
    load_col<EA>(A.js, j, lane, js);
    load_col<EA>(A.jn, j, lane, jn);
    load_col<EA>(A.br, task, lane, fr);
    load_col<EA>(A.bcr, task, lane, fcr);
    load_col<EA>(A.bci, task, lane, fci);
    load_col<EA>(A.xr, cs, lane, xs); load_col<EA>(A.xr, cn, lane, xn);
    for (int e = 0; e < EA; ++e) fr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fr[e]));
    load_col<EA>(A.xcr, cs, lane, xs); load_col<EA>(A.xcr, cn, lane, xn);
    for (int e = 0; e < EA; ++e) fcr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fcr[e]));
    load_col<EA>(A.xci, cs, lane, xs); load_col<EA>(A.xci, cn, lane, xn);
    for (int e = 0; e < EA; ++e) fci[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fci[e]));
    double inv[EA], t0[EA], t1[EA], tab[NK2D_TAB];
    cplx invc[EA], tabc[NK2D_TAB], r[EA];
    load_col<EA>(A.inv_r, task, lane, inv);
    load_col<EA>(A.inv_cr, task, lane, t0);
    load_col<EA>(A.inv_ci, task, lane, t1);
    for (int i = 0; i < NK2D_TAB; ++i) {
        tab[i] = A.tab_r[((size_t)task * NK2D_TAB + i) * 64 + lane];
        tabc[i] = c_make(A.tab_cr[((size_t)task * NK2D_TAB + i) * 64 + lane], A.tab_ci[((size_t)task * NK2D_TAB + i) * 64 + lane]);
    }
    for (int e = 0; e < EA; ++e) {
        const bool valid = (lane * EA + e) < NZ;
        fr[e] = valid ? fr[e] : 0.0;
        invc[e] = c_make(t0[e], t1[e]);
        r[e] = c_make(valid ? fcr[e] : 0.0, valid ? fci[e] : 0.0);
    }
    double al[EA], be[EA];
    cplx alc[EA], bec[EA];
    al[0] = a[0];
    alc[0] = c_make(a[0], 0.0);
    for (int i = 1; i < EA; ++i) {
        double m = inv[i - 1] * a[i];
        cplx mc = t_mulr(invc[i - 1], a[i]);
        fr[i] = __builtin_fma(-m, fr[i - 1], fr[i]);
        r[i] = t_nfma(r[i], mc, r[i - 1]);
        al[i] = -(m * al[i - 1]);
        alc[i] = t_neg(t_mul(mc, alc[i - 1]));
    }
    be[EA - 1] = 0.0;
    be[EA - 2] = c[EA - 2];
    bec[EA - 1] = c_make(0.0, 0.0);
    bec[EA - 2] = c_make(c[EA - 2], 0.0);
    for (int i = EA - 3; i >= 0; --i) {
        double m = inv[i + 1] * c[i];
        cplx mc = t_mulr(invc[i + 1], c[i]);
        fr[i] = __builtin_fma(-m, fr[i + 1], fr[i]);
        r[i] = t_nfma(r[i], mc, r[i + 1]);
        al[i] = __builtin_fma(-m, al[i + 1], al[i]);
        alc[i] = t_nfma(alc[i], mc, alc[i + 1]);
        be[i] = -(m * be[i + 1]);
        bec[i] = t_neg(t_mul(mc, bec[i + 1]));
    }
    double R = __builtin_fma(-tab[13], __shfl_down(fr[0], 1, 64), fr[EA - 1]);
    cplx Rc = t_nfma(r[EA - 1], tabc[13], shfl_down_t(r[0], 1));
    int lv = 0;
    for (int s = 1; s < 64; s <<= 1, ++lv) {
        const double Rm = __shfl_up(R, s, 64), Rp = __shfl_down(R, s, 64);
        const cplx Rmc = shfl_up_t(Rc, s), Rpc = shfl_down_t(Rc, s);
        R = __builtin_fma(-Rp, tab[6 + lv], __builtin_fma(-Rm, tab[lv], R));
        Rc = t_nfma(t_nfma(Rc, Rmc, tabc[lv]), Rpc, tabc[6 + lv]);
    }
    const double xl = R * tab[12];
    const cplx xlc = t_mul(Rc, tabc[12]);
    double xp = __shfl_up(xl, 1, 64);
    cplx xpc = shfl_up_t(xlc, 1);
    if (lane == 0) { xp = 0.0; xpc = c_make(0.0, 0.0); }
    for (int i = 0; i < EA - 1; ++i) {
        fr[i] = inv[i] * __builtin_fma(-be[i], xl, __builtin_fma(-al[i], xp, fr[i]));
        r[i] = t_mul(invc[i], t_nfma(t_nfma(r[i], alc[i], xpc), bec[i], xlc));
    }
    fr[EA - 1] = xl;
    r[EA - 1] = xlc;
    for (int e = 0; e < EA; ++e) { fcr[e] = r[e].re; fci[e] = r[e].im; }
    store_col<EA>(A.yr, task, lane, fr);
    store_col<EA>(A.ycr, task, lane, fcr);
    store_col<EA>(A.yci, task, lane, fci);
}

// ---------------------------------------------------------------------------------
// variant B: 128 lanes per column, neighbour access through LDS
// ---------------------------------------------------------------------------------
template <int E>
__device__ __forceinline__ void loadB(const double* base, size_t col, int L, double (&o)[E]) {
    const double* p = base + col * (size_t)(E * LB) + L;
    for (int e = 0; e < E; ++e) o[e] = p[e * LB];
}
template <int E>
__device__ __forceinline__ void storeB(double* base, size_t col, int L, const double (&v)[E]) {
    double* p = base + col * (size_t)(E * LB) + L;
    for (int e = 0; e < E; ++e) p[e * LB] = v[e];
}

// one exchange through LDS: every lane publishes NV doubles, then reads those of lanes L-s and L+s
// (zeros outside the column).  buf toggles between two areas so that one barrier per exchange is enough.
template <int NV>
__device__ __forceinline__ void exchange(double (*lds)[LB], int& buf, int L, int s, const double (&mine)[NV],
                                         double (&minus)[NV], double (&plus)[NV]) {
    double(*area)[LB] = lds + buf * NV;
    for (int v = 0; v < NV; ++v) area[v][L] = mine[v];
    __syncthreads();
    const bool hm = L >= s, hp = L + s < LB;
    for (int v = 0; v < NV; ++v) {
        minus[v] = hm ? area[v][L - s] : 0.0;
        plus[v] = hp ? area[v][L + s] : 0.0;
    }
    buf ^= 1;
}

struct ArgsB {
    const double *a, *c, *dre;
    double dim;
    double *inv_r, *tab_r, *inv_cr, *inv_ci, *tab_cr, *tab_ci;   // tab: [NCOL][TAB2][LB]
    const double *js, *jn;
    const double *br, *bcr, *bci, *xr, *xcr, *xci;
    double *yr, *ycr, *yci;
};

constexpr int XV = 9;   // doubles per lane in the widest exchange (A, iB, C real + complex)

// factorisation of the real AND the complex system of one column (partition method, PCR via LDS)
__global__ void __launch_bounds__(LB) kB_factor(ArgsB A) {
    __shared__ double lds[2 * XV][LB];
    int buf = 0;
    const int L = threadIdx.x, task = blockIdx.x;
    const int j = task % NY;
    double a[EB], c[EB], d[EB];
    loadB<EB>(A.a, j, L, a);
    loadB<EB>(A.c, j, L, c);
    loadB<EB>(A.dre, j, L, d);
    // ---- real
    double inv[EB], al[EB], be[EB];
    cplx invc[EB], alc[EB], bec[EB], dc[EB];
    for (int e = 0; e < EB; ++e) dc[e] = c_make(d[e], ((L * EB + e) < NZ) ? A.dim : 0.0);
    double dlast = d[0];
    cplx dlastc = dc[0];
    inv[0] = fast_rcp(d[0]);
    invc[0] = t_recip(dc[0]);
    al[0] = a[0];
    alc[0] = c_make(a[0], 0.0);
    for (int i = 1; i < EB; ++i) {
        double m = inv[i - 1] * a[i];
        double dd = __builtin_fma(-m, c[i - 1], d[i]);
        dlast = dd;
        inv[i] = fast_rcp(dd);
        al[i] = -(m * al[i - 1]);
        cplx mc = t_mulr(invc[i - 1], a[i]);
        cplx ddc = t_nfmar(dc[i], mc, c[i - 1]);
        dlastc = ddc;
        invc[i] = t_recip(ddc);
        alc[i] = t_neg(t_mul(mc, alc[i - 1]));
    }
    be[EB - 1] = 0.0;
    be[EB - 2] = c[EB - 2];
    bec[EB - 1] = c_make(0.0, 0.0);
    bec[EB - 2] = c_make(c[EB - 2], 0.0);
    for (int i = EB - 3; i >= 0; --i) {
        double m = inv[i + 1] * c[i];
        al[i] = __builtin_fma(-m, al[i + 1], al[i]);
        be[i] = -(m * be[i + 1]);
        cplx mc = t_mulr(invc[i + 1], c[i]);
        alc[i] = t_nfma(alc[i], mc, alc[i + 1]);
        bec[i] = t_neg(t_mul(mc, bec[i + 1]));
    }
    // values of lane L+1
    double mine[XV] = {al[0], inv[0], be[0], alc[0].re, alc[0].im, invc[0].re, invc[0].im, bec[0].re, bec[0].im};
    double mi[XV], pl[XV];
    exchange<XV>(lds, buf, L, 1, mine, mi, pl);
    const bool last = L == LB - 1;   // no lane L+1: identity continuation
    double G = last ? 0.0 : pl[1] * c[EB - 1];
    double Ar = al[EB - 1], Br = last ? dlast : __builtin_fma(-G, pl[0], dlast), Cr = last ? 0.0 : -(G * pl[2]);
    cplx Gc = last ? c_make(0.0, 0.0) : t_mulr(c_make(pl[5], pl[6]), c[EB - 1]);
    cplx Ac = alc[EB - 1];
    cplx Bc = last ? dlastc : t_nfma(dlastc, Gc, c_make(pl[3], pl[4]));
    cplx Cc = last ? c_make(0.0, 0.0) : t_neg(t_mul(Gc, c_make(pl[7], pl[8])));
    double tab[TAB2];
    cplx tabc[TAB2];
    int lv = 0;
    for (int s = 1; s < LB; s <<= 1, ++lv) {
        const double iB = fast_rcp(Br);
        const cplx iBc = t_recip(Bc);
        double pub[XV] = {Ar, iB, Cr, Ac.re, Ac.im, iBc.re, iBc.im, Cc.re, Cc.im};
        exchange<XV>(lds, buf, L, s, pub, mi, pl);
        // real: partners' (A, iB, C) are mi[0..2] / pl[0..2]; complex: A (3,4), iB (5,6), C (7,8)
        const double k1 = Ar * mi[1], k2 = Cr * pl[1];
        tab[lv] = k1;
        tab[7 + lv] = k2;
        Br = __builtin_fma(-k2, pl[0], __builtin_fma(-k1, mi[2], Br));
        Ar = -(mi[0] * k1);
        Cr = -(pl[2] * k2);
        const cplx k1c = t_mul(Ac, c_make(mi[5], mi[6])), k2c = t_mul(Cc, c_make(pl[5], pl[6]));
        tabc[lv] = k1c;
        tabc[7 + lv] = k2c;
        Bc = t_nfma(t_nfma(Bc, c_make(mi[7], mi[8]), k1c), c_make(pl[3], pl[4]), k2c);
        Ac = t_neg(t_mul(c_make(mi[3], mi[4]), k1c));
        Cc = t_neg(t_mul(c_make(pl[7], pl[8]), k2c));
    }
    tab[14] = fast_rcp(Br);
    tab[15] = G;
    tabc[14] = t_recip(Bc);
    tabc[15] = Gc;
    storeB<EB>(A.inv_r, task, L, inv);
    double re[EB], im[EB];
    for (int e = 0; e < EB; ++e) { re[e] = invc[e].re; im[e] = invc[e].im; }
    storeB<EB>(A.inv_cr, task, L, re);
    storeB<EB>(A.inv_ci, task, L, im);
    for (int i = 0; i < TAB2; ++i) {
        A.tab_r[((size_t)task * TAB2 + i) * LB + L] = tab[i];
        A.tab_cr[((size_t)task * TAB2 + i) * LB + L] = tabc[i].re;
        A.tab_ci[((size_t)task * TAB2 + i) * LB + L] = tabc[i].im;
    }
}

// middle sweep: two columns per workgroup would need two LDS areas; one column (128 threads) per
// workgroup keeps the prototype simple
__global__ void __launch_bounds__(LB) kB_sweep(ArgsB A) {
    __shared__ double lds[2 * 3][LB];
    int buf = 0;
    const int L = threadIdx.x, task = blockIdx.x;
    const int j = task % NY;
    const int cs = (j > 0) ? task - 1 : task, cn = (j < NY - 1) ? task + 1 : task;
    double a[EB], c[EB], js[EB], jn[EB], fr[EB], fcr[EB], fci[EB], xs[EB], xn[EB];
    loadB<EB>(A.a, j, L, a);
    loadB<EB>(A.c, j, L, c);
    loadB<EB>(A.js, j, L, js);
    loadB<EB>(A.jn, j, L, jn);
    loadB<EB>(A.br, task, L, fr);
    loadB<EB>(A.bcr, task, L, fcr);
    loadB<EB>(A.bci, task, L, fci);
    loadB<EB>(A.xr, cs, L, xs); loadB<EB>(A.xr, cn, L, xn);
    for (int e = 0; e < EB; ++e) fr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fr[e]));
    loadB<EB>(A.xcr, cs, L, xs); loadB<EB>(A.xcr, cn, L, xn);
    for (int e = 0; e < EB; ++e) fcr[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fcr[e]));
    loadB<EB>(A.xci, cs, L, xs); loadB<EB>(A.xci, cn, L, xn);
    for (int e = 0; e < EB; ++e) fci[e] = __builtin_fma(jn[e], xn[e], __builtin_fma(js[e], xs[e], fci[e]));
    double inv[EB], t0[EB], t1[EB], tab[TAB2];
    cplx invc[EB], tabc[TAB2], r[EB];
    loadB<EB>(A.inv_r, task, L, inv);
    loadB<EB>(A.inv_cr, task, L, t0);
    loadB<EB>(A.inv_ci, task, L, t1);
    for (int i = 0; i < TAB2; ++i) {
        tab[i] = A.tab_r[((size_t)task * TAB2 + i) * LB + L];
        tabc[i] = c_make(A.tab_cr[((size_t)task * TAB2 + i) * LB + L], A.tab_ci[((size_t)task * TAB2 + i) * LB + L]);
    }
    for (int e = 0; e < EB; ++e) {
        const bool valid = (L * EB + e) < NZ;
        fr[e] = valid ? fr[e] : 0.0;
        invc[e] = c_make(t0[e], t1[e]);
        r[e] = c_make(valid ? fcr[e] : 0.0, valid ? fci[e] : 0.0);
    }
    // local elimination, real and complex side by side
    double al[EB], be[EB];
    cplx alc[EB], bec[EB];
    al[0] = a[0];
    alc[0] = c_make(a[0], 0.0);
    for (int i = 1; i < EB; ++i) {
        double m = inv[i - 1] * a[i];
        fr[i] = __builtin_fma(-m, fr[i - 1], fr[i]);
        al[i] = -(m * al[i - 1]);
        cplx mc = t_mulr(invc[i - 1], a[i]);
        r[i] = t_nfma(r[i], mc, r[i - 1]);
        alc[i] = t_neg(t_mul(mc, alc[i - 1]));
    }
    be[EB - 1] = 0.0;
    be[EB - 2] = c[EB - 2];
    bec[EB - 1] = c_make(0.0, 0.0);
    bec[EB - 2] = c_make(c[EB - 2], 0.0);
    for (int i = EB - 3; i >= 0; --i) {
        double m = inv[i + 1] * c[i];
        fr[i] = __builtin_fma(-m, fr[i + 1], fr[i]);
        al[i] = __builtin_fma(-m, al[i + 1], al[i]);
        be[i] = -(m * be[i + 1]);
        cplx mc = t_mulr(invc[i + 1], c[i]);
        r[i] = t_nfma(r[i], mc, r[i + 1]);
        alc[i] = t_nfma(alc[i], mc, alc[i + 1]);
        bec[i] = t_neg(t_mul(mc, bec[i + 1]));
    }
    double mine[3] = {fr[0], r[0].re, r[0].im}, mi[3], pl[3];
    exchange<3>(lds, buf, L, 1, mine, mi, pl);
    double R = __builtin_fma(-tab[15], pl[0], fr[EB - 1]);
    cplx Rc = t_nfma(r[EB - 1], tabc[15], c_make(pl[1], pl[2]));
    int lv = 0;
    for (int s = 1; s < LB; s <<= 1, ++lv) {
        double pub[3] = {R, Rc.re, Rc.im};
        exchange<3>(lds, buf, L, s, pub, mi, pl);
        R = __builtin_fma(-pl[0], tab[7 + lv], __builtin_fma(-mi[0], tab[lv], R));
        Rc = t_nfma(t_nfma(Rc, c_make(mi[1], mi[2]), tabc[lv]), c_make(pl[1], pl[2]), tabc[7 + lv]);
    }
    const double xl = R * tab[14];
    const cplx xlc = t_mul(Rc, tabc[14]);
    double pub[3] = {xl, xlc.re, xlc.im};
    exchange<3>(lds, buf, L, 1, pub, mi, pl);
    const double xp = mi[0];
    const cplx xpc = c_make(mi[1], mi[2]);
    for (int i = 0; i < EB - 1; ++i) {
        double v = __builtin_fma(-be[i], xl, __builtin_fma(-al[i], xp, fr[i]));
        fr[i] = inv[i] * v;
        cplx vc = t_nfma(t_nfma(r[i], alc[i], xpc), bec[i], xlc);
        r[i] = t_mul(invc[i], vc);
    }
    fr[EB - 1] = xl;
    r[EB - 1] = xlc;
    for (int e = 0; e < EB; ++e) { fcr[e] = r[e].re; fci[e] = r[e].im; }
    storeB<EB>(A.yr, task, L, fr);
    storeB<EB>(A.ycr, task, L, fcr);
    storeB<EB>(A.yci, task, L, fci);
}

// ---------------------------------------------------------------------------------
// host: same numbers in both layouts
// ---------------------------------------------------------------------------------
static size_t idxA(int col, int k) { int lane = k / EA, e = k % EA; return ((size_t)col * EA + e) * 64 + lane; }
static size_t idxB(int col, int k) { int L = k / EB, e = k % EB; return ((size_t)col * EB + e) * LB + L; }

int main() {
    const size_t nA_plane = (size_t)NY * EA * 64, nA = (size_t)NCOL * EA * 64;
    const size_t nB_plane = (size_t)NY * EB * LB, nB = (size_t)NCOL * EB * LB;
    std::vector<double> hA[16], hB[16];
    // planes: 0 a, 1 c, 2 dre, 3 js, 4 jn; vectors: 5 br, 6 bcr, 7 bci, 8 xr, 9 xcr, 10 xci
    for (int q = 0; q < 5; ++q) { hA[q].assign(nA_plane, 0.0); hB[q].assign(nB_plane, 0.0); }
    for (int q = 5; q < 11; ++q) { hA[q].assign(nA, 0.0); hB[q].assign(nB, 0.0); }
    // identity rows in the padding
    for (int j = 0; j < NY; ++j) {
        for (int k = 0; k < EA * 64; ++k) hA[2][idxA(j, k)] = 1.0;
        for (int k = 0; k < EB * LB; ++k) hB[2][idxB(j, k)] = 1.0;
    }
    const double shift = 2.4e-3;
    unsigned long long seed = 12345;
    auto rnd = [&]() { seed = seed * 6364136223846793005ULL + 1442695040888963407ULL; return (double)((seed >> 11) & 0xFFFFFFFFFFFFFULL) / (double)0x10000000000000ULL; };
    for (int j = 0; j < NY; ++j)
        for (int k = 0; k < NZ; ++k) {
            const double kappa_up = (k > 0) ? 0.1 * std::exp(-k / 12.0) + 1e-6 : 0.0;
            const double kappa_dn = (k < NZ - 1) ? 0.1 * std::exp(-(k + 1) / 12.0) + 1e-6 : 0.0;
            const double vals[5] = {-kappa_up, -kappa_dn, shift + kappa_up + kappa_dn + 3e-6, 1e-6 * (1 + rnd()), 1e-6 * (1 + rnd())};
            for (int q = 0; q < 5; ++q) { hA[q][idxA(j, k)] = vals[q]; hB[q][idxB(j, k)] = vals[q]; }
        }
    for (int col = 0; col < NCOL; ++col)
        for (int k = 0; k < NZ; ++k)
            for (int q = 5; q < 11; ++q) { const double v = rnd() - 0.5; hA[q][idxA(col, k)] = v; hB[q][idxB(col, k)] = v; }

    auto up = [&](const std::vector<double>& h, double** d) {
        if (hipMalloc((void**)d, h.size() * 8) != hipSuccess) return 1;
        return hipMemcpy(*d, h.data(), h.size() * 8, hipMemcpyHostToDevice) != hipSuccess ? 1 : 0;
    };
    double *dA[11], *dB[11];
    for (int q = 0; q < 11; ++q) { if (up(hA[q], &dA[q]) || up(hB[q], &dB[q])) { printf("alloc failed\n"); return 1; } }
    ArgsA A = {};
    A.a = dA[0]; A.c = dA[1]; A.dre = dA[2]; A.js = dA[3]; A.jn = dA[4];
    A.br = dA[5]; A.bcr = dA[6]; A.bci = dA[7]; A.xr = dA[8]; A.xcr = dA[9]; A.xci = dA[10];
    A.dim = 0.9 * shift;
    ArgsB B = {};
    B.a = dB[0]; B.c = dB[1]; B.dre = dB[2]; B.js = dB[3]; B.jn = dB[4];
    B.br = dB[5]; B.bcr = dB[6]; B.bci = dB[7]; B.xr = dB[8]; B.xcr = dB[9]; B.xci = dB[10];
    B.dim = A.dim;
    CHECK(hipMalloc((void**)&A.inv_r, nA * 8)); CHECK(hipMalloc((void**)&A.inv_cr, nA * 8)); CHECK(hipMalloc((void**)&A.inv_ci, nA * 8));
    const size_t tA = (size_t)NCOL * NK2D_TAB * 64, tB = (size_t)NCOL * TAB2 * LB;
    CHECK(hipMalloc((void**)&A.tab_r, tA * 8)); CHECK(hipMalloc((void**)&A.tab_cr, tA * 8)); CHECK(hipMalloc((void**)&A.tab_ci, tA * 8));
    CHECK(hipMalloc((void**)&A.yr, nA * 8)); CHECK(hipMalloc((void**)&A.ycr, nA * 8)); CHECK(hipMalloc((void**)&A.yci, nA * 8));
    CHECK(hipMalloc((void**)&B.inv_r, nB * 8)); CHECK(hipMalloc((void**)&B.inv_cr, nB * 8)); CHECK(hipMalloc((void**)&B.inv_ci, nB * 8));
    CHECK(hipMalloc((void**)&B.tab_r, tB * 8)); CHECK(hipMalloc((void**)&B.tab_cr, tB * 8)); CHECK(hipMalloc((void**)&B.tab_ci, tB * 8));
    CHECK(hipMalloc((void**)&B.yr, nB * 8)); CHECK(hipMalloc((void**)&B.ycr, nB * 8)); CHECK(hipMalloc((void**)&B.yci, nB * 8));

    hipLaunchKernelGGL(kA_factor, dim3((NCOL + 3) / 4), dim3(256), 0, 0, A);
    hipLaunchKernelGGL(kB_factor, dim3(NCOL), dim3(LB), 0, 0, B);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = 400;
    float msA = 0, msB = 0, msA2 = 0, msA0 = 0, msA0w = 0;
    for (int pass = 0; pass < 2; ++pass) {   // first pass warms up
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kA_sweep, dim3((NCOL + 3) / 4), dim3(256), 0, 0, A);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&msA, e0, e1));
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kA0_sweep, dim3((NCOL + 3) / 4), dim3(256), 0, 0, A);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&msA0, e0, e1));
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kA0w_sweep, dim3((NCOL + 3) / 4), dim3(256), 0, 0, A);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&msA0w, e0, e1));
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kA2_sweep, dim3((NCOL + 3) / 4), dim3(256), 0, 0, A);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&msA2, e0, e1));
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kB_sweep, dim3(NCOL), dim3(LB), 0, 0, B);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&msB, e0, e1));
    }
    CHECK(hipGetLastError());
    {
        float ms = 0;
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_empty, dim3(208), dim3(256), 0, 0, i);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty kernel, 208 workgroups, back to back: %.2f us per launch\n", 1000.0 * ms / 2000);
    }
    std::vector<double> ya(nA), yb(nB);
    double worst = 0.0, scale = 0.0;
    double* outA[3] = {A.yr, A.ycr, A.yci};
    double* outB[3] = {B.yr, B.ycr, B.yci};
    for (int q = 0; q < 3; ++q) {
        CHECK(hipMemcpy(ya.data(), outA[q], nA * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(yb.data(), outB[q], nB * 8, hipMemcpyDeviceToHost));
        for (int col = 0; col < NCOL; ++col)
            for (int k = 0; k < NZ; ++k) {
                const double va = ya[idxA(col, k)], vb = yb[idxB(col, k)];
                worst = std::fmax(worst, std::fabs(va - vb));
                scale = std::fmax(scale, std::fabs(va));
            }
    }
    printf("columns %d, nz %d: one wave per column (E=7) %.2f us per launch; same with the two solves side by side %.2f us; "
           "two waves per column (E=4, LDS PCR) %.2f us per launch\n",
           NCOL, NZ, 1000.0 * msA / reps, 1000.0 * msA2 / reps, 1000.0 * msB / reps);
    printf("loads and stores of the one-wave sweep without the solves: %.2f us per launch; with 16-byte accesses per lane: %.2f us\n",
           1000.0 * msA0 / reps, 1000.0 * msA0w / reps);
    printf("max |xA - xB| = %.3e (scale %.3e)\n", worst, scale);
    return (worst <= 1e-9 * scale) ? 0 : 2;
}
