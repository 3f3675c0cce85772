// proto_persistent.hip -- stand-alone experiment for the next round (not part of libnk2d.so):
// what would a PERSISTENT line-relaxation kernel gain?  The product runs every sweep of the real +
// complex systems of all columns as one launch (k_newton_fused): factor tables, coefficients and
// right-hand sides are re-read by every launch although only the neighbours' iterates change.
//
//   launches  : nsweeps launches of the product-style sweep (variant A of proto_two_wave.hip)
//   persistent: ONE cooperative launch, one wave per column, everything static held in registers,
//               the iterates exchanged through memory with a grid-wide barrier (device-scope atomic
//               counter + fences) between sweeps; the barrier spin has a bounded count
//
// Both run the same Jacobi sweeps on the same 832 x 2 systems; the program checks that they agree and
// prints the time of each.  Build and run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I include tools/proto_persistent.hip -o /tmp/protop && /tmp/protop
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
constexpr int EA = 7;

struct Args {
    const double *a, *c, *dre;                 // [NY] planes (shared by the tracers)
    double dim;                                // imaginary part of the complex diagonal
    double *inv_r, *tab_r, *inv_cr, *inv_ci, *tab_cr, *tab_ci;
    const double *js, *jn;                     // lateral couplings, [NY] planes
    const double *br, *bcr, *bci;
    double *x[2][3];                           // ping-pong iterates: real, complex re, complex im
    unsigned* counter;
    int* err;
    int nsweeps;
};

__global__ void __launch_bounds__(256) k_factor(Args A) {
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

struct ColStatic {
    double a[EA], c[EA], js[EA], jn[EA], br[EA], bcr[EA], bci[EA], inv_r[EA], tab_r[NK2D_TAB];
    cplx inv_c[EA], tab_c[NK2D_TAB];
};

__device__ __forceinline__ void load_static(const Args& A, int task, int j, int lane, ColStatic& S) {
    load_col<EA>(A.a, j, lane, S.a);
    load_col<EA>(A.c, j, lane, S.c);
    load_col<EA>(A.js, j, lane, S.js);
    load_col<EA>(A.jn, j, lane, S.jn);
    load_col<EA>(A.br, task, lane, S.br);
    load_col<EA>(A.bcr, task, lane, S.bcr);
    load_col<EA>(A.bci, task, lane, S.bci);
    load_col<EA>(A.inv_r, task, lane, S.inv_r);
    for (int i = 0; i < NK2D_TAB; ++i) S.tab_r[i] = A.tab_r[((size_t)task * NK2D_TAB + i) * 64 + lane];
    double t0[EA], t1[EA];
    load_col<EA>(A.inv_cr, task, lane, t0);
    load_col<EA>(A.inv_ci, task, lane, t1);
    for (int e = 0; e < EA; ++e) S.inv_c[e] = c_make(t0[e], t1[e]);
    for (int i = 0; i < NK2D_TAB; ++i)
        S.tab_c[i] = c_make(A.tab_cr[((size_t)task * NK2D_TAB + i) * 64 + lane], A.tab_ci[((size_t)task * NK2D_TAB + i) * 64 + lane]);
}

// one Jacobi sweep of the column: x_new = T^-1 (b + S x_old[j-1] + N x_old[j+1]) for the real and the complex system
__device__ __forceinline__ void sweep_col(const ColStatic& S, double* const (&xo)[3], double* const (&xn_)[3], int task,
                                          int cs, int cn, int lane) {
    double fr[EA], fcr[EA], fci[EA], xs[EA], xn[EA];
    load_col<EA>(xo[0], cs, lane, xs); load_col<EA>(xo[0], cn, lane, xn);
    for (int e = 0; e < EA; ++e) fr[e] = __builtin_fma(S.jn[e], xn[e], __builtin_fma(S.js[e], xs[e], S.br[e]));
    load_col<EA>(xo[1], cs, lane, xs); load_col<EA>(xo[1], cn, lane, xn);
    for (int e = 0; e < EA; ++e) fcr[e] = __builtin_fma(S.jn[e], xn[e], __builtin_fma(S.js[e], xs[e], S.bcr[e]));
    load_col<EA>(xo[2], cs, lane, xs); load_col<EA>(xo[2], cn, lane, xn);
    for (int e = 0; e < EA; ++e) fci[e] = __builtin_fma(S.jn[e], xn[e], __builtin_fma(S.js[e], xs[e], S.bci[e]));
    for (int e = 0; e < EA; ++e) fr[e] = ((lane * EA + e) < NZ) ? fr[e] : 0.0;
    tridiag_apply<EA, double>(S.a, S.c, S.inv_r, S.tab_r, fr, lane);
    cplx r[EA];
    for (int e = 0; e < EA; ++e) {
        const bool valid = (lane * EA + e) < NZ;
        r[e] = c_make(valid ? fcr[e] : 0.0, valid ? fci[e] : 0.0);
    }
    tridiag_apply<EA, cplx>(S.a, S.c, S.inv_c, S.tab_c, r, lane);
    for (int e = 0; e < EA; ++e) { fcr[e] = r[e].re; fci[e] = r[e].im; }
    store_col<EA>(xn_[0], task, lane, fr);
    store_col<EA>(xn_[1], task, lane, fcr);
    store_col<EA>(xn_[2], task, lane, fci);
}

// product style: one launch per sweep, everything re-read
__global__ void __launch_bounds__(256) k_sweep(Args A, int from) {
    const int lane = threadIdx.x & 63, task = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= NCOL) return;
    const int j = task % NY;
    const int cs = (j > 0) ? task - 1 : task, cn = (j < NY - 1) ? task + 1 : task;
    ColStatic S;
    load_static(A, task, j, lane, S);
    sweep_col(S, A.x[from], A.x[1 - from], task, cs, cn, lane);
}

// grid-wide barrier: every workgroup adds one to the counter and waits for `target`; bounded spin.
// MODE 0: every thread fences at device scope before and after; MODE 1: only the arriving thread does (the
// workgroup barrier orders the others' stores before its release and their loads after its acquire);
// MODE 2: as 1 for iterates in uncached memory (no cache maintenance needed, the fences only order)
template <int MODE>
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* err) {
    if (MODE == 0) __threadfence();
    __syncthreads();
    __shared__ int failed;
    if (threadIdx.x == 0) {
        failed = 0;
        __hip_atomic_fetch_add(counter, 1u, (MODE == 2) ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 20)) { failed = 1; *err = 1; break; }
        }
        if (MODE != 2) __atomic_thread_fence(__ATOMIC_ACQUIRE);   // HIP: agent scope
    }
    __syncthreads();
    if (MODE == 0) __threadfence();
    return failed == 0;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_persistent(Args A) {
    const int lane = threadIdx.x & 63, task = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool active = task < NCOL;
    const int tk = active ? task : NCOL - 1;
    const int j = tk % NY;
    const int cs = (j > 0) ? tk - 1 : tk, cn = (j < NY - 1) ? tk + 1 : tk;
    ColStatic S;
    load_static(A, tk, j, lane, S);
    for (int s = 0; s < A.nsweeps; ++s) {
        if (active) sweep_col(S, A.x[s & 1], A.x[1 - (s & 1)], tk, cs, cn, lane);
        if (s + 1 < A.nsweeps && !grid_barrier<MODE>(A.counter, (unsigned)(s + 1) * gridDim.x, A.err)) return;
    }
}

// barrier cost alone
template <int MODE>
__global__ void __launch_bounds__(256) k_barriers(Args A) {
    for (int s = 0; s < A.nsweeps; ++s)
        if (!grid_barrier<MODE>(A.counter, (unsigned)(s + 1) * gridDim.x, A.err)) return;
}

static size_t idxA(int col, int k) { return ((size_t)col * EA + (k % EA)) * 64 + (k / EA); }

int main() {
    const size_t n_plane = (size_t)NY * EA * 64, n_vec = (size_t)NCOL * EA * 64;
    std::vector<double> h[8];
    // planes: 0 a, 1 c, 2 dre, 3 js, 4 jn; vectors: 5 br, 6 bcr, 7 bci
    for (int q = 0; q < 5; ++q) h[q].assign(n_plane, 0.0);
    for (int q = 5; q < 8; ++q) h[q].assign(n_vec, 0.0);
    for (int j = 0; j < NY; ++j)
        for (int k = 0; k < EA * 64; ++k) h[2][idxA(j, k)] = 1.0;
    const double shift = 2.4e-3;
    unsigned long long seed = 12345;
    auto rnd = [&]() { seed = seed * 6364136223846793005ULL + 1442695040888963407ULL; return (double)((seed >> 11) & 0xFFFFFFFFFFFFFULL) / (double)0x10000000000000ULL; };
    for (int j = 0; j < NY; ++j)
        for (int k = 0; k < NZ; ++k) {
            const double kappa_up = (k > 0) ? 0.1 * std::exp(-k / 12.0) + 1e-6 : 0.0;
            const double kappa_dn = (k < NZ - 1) ? 0.1 * std::exp(-(k + 1) / 12.0) + 1e-6 : 0.0;
            const double vals[5] = {-kappa_up, -kappa_dn, shift + kappa_up + kappa_dn + 3e-4, 1e-4 * (1 + rnd()), 1e-4 * (1 + rnd())};
            for (int q = 0; q < 5; ++q) h[q][idxA(j, k)] = vals[q];
        }
    for (int col = 0; col < NCOL; ++col)
        for (int k = 0; k < NZ; ++k)
            for (int q = 5; q < 8; ++q) h[q][idxA(col, k)] = rnd() - 0.5;
    double* d[8];
    for (int q = 0; q < 8; ++q) {
        CHECK(hipMalloc((void**)&d[q], h[q].size() * 8));
        CHECK(hipMemcpy(d[q], h[q].data(), h[q].size() * 8, hipMemcpyHostToDevice));
    }
    Args A = {};
    A.a = d[0]; A.c = d[1]; A.dre = d[2]; A.js = d[3]; A.jn = d[4]; A.br = d[5]; A.bcr = d[6]; A.bci = d[7];
    A.dim = 0.9 * shift;
    CHECK(hipMalloc((void**)&A.inv_r, n_vec * 8)); CHECK(hipMalloc((void**)&A.inv_cr, n_vec * 8)); CHECK(hipMalloc((void**)&A.inv_ci, n_vec * 8));
    const size_t nt = (size_t)NCOL * NK2D_TAB * 64;
    CHECK(hipMalloc((void**)&A.tab_r, nt * 8)); CHECK(hipMalloc((void**)&A.tab_cr, nt * 8)); CHECK(hipMalloc((void**)&A.tab_ci, nt * 8));
    double *xc[2][3], *xu[2][3];
    for (int b = 0; b < 2; ++b)
        for (int q = 0; q < 3; ++q) {
            CHECK(hipMalloc((void**)&xc[b][q], n_vec * 8));
            CHECK(hipExtMallocWithFlags((void**)&xu[b][q], n_vec * 8, hipDeviceMallocUncached));
            A.x[b][q] = xc[b][q];
        }
    CHECK(hipMalloc((void**)&A.counter, 64));
    CHECK(hipMalloc((void**)&A.err, 64));
    CHECK(hipMemset(A.err, 0, 64));
    const int nwg = (NCOL + 3) / 4;
    hipLaunchKernelGGL(k_factor, dim3(nwg), dim3(256), 0, 0, A);
    CHECK(hipDeviceSynchronize());

    int dev = 0, coop = 0, max_blocks = 0, cus = 0;
    CHECK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&max_blocks, k_persistent<0>, 256, 0));
    printf("cooperative launch %d, %d CUs, %d workgroups of k_persistent per CU, grid %d\n", coop, cus, max_blocks, nwg);
    if (!coop || max_blocks * cus < nwg) { printf("grid cannot be co-resident\n"); return 3; }

    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto reset_x = [&]() {
        for (int q = 0; q < 3; ++q) if (hipMemset(A.x[0][q], 0, n_vec * 8) != hipSuccess) return 1;
        return hipMemset(A.counter, 0, 64) != hipSuccess ? 1 : 0;
    };
    std::vector<double> ref(n_vec), got(n_vec);
    for (int mode = 0; mode < 3; ++mode)
    for (int nsweeps : {3, 24}) {
        A.nsweeps = nsweeps;
        for (int b = 0; b < 2; ++b)
            for (int q = 0; q < 3; ++q) A.x[b][q] = (mode == 2) ? xu[b][q] : xc[b][q];
        const void* kp = (mode == 0) ? (const void*)k_persistent<0> : (mode == 1) ? (const void*)k_persistent<1> : (const void*)k_persistent<2>;
        const void* kb = (mode == 0) ? (const void*)k_barriers<0> : (mode == 1) ? (const void*)k_barriers<1> : (const void*)k_barriers<2>;
        const int reps = 200;
        float ms_l = 0, ms_p = 0, ms_b = 0;
        // launches
        for (int pass = 0; pass < 2; ++pass) {
            if (reset_x()) return 1;
            CHECK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r)
                for (int s = 0; s < nsweeps; ++s) hipLaunchKernelGGL(k_sweep, dim3(nwg), dim3(256), 0, 0, A, s & 1);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms_l, e0, e1));
        }
        // result of ONE group of nsweeps from zero
        if (reset_x()) return 1;
        for (int s = 0; s < nsweeps; ++s) hipLaunchKernelGGL(k_sweep, dim3(nwg), dim3(256), 0, 0, A, s & 1);
        CHECK(hipMemcpy(ref.data(), A.x[nsweeps & 1][1], n_vec * 8, hipMemcpyDeviceToHost));
        // persistent: the counter is reset between launches by a memset on the stream
        void* params[1] = {(void*)&A};
        for (int pass = 0; pass < 2; ++pass) {
            if (reset_x()) return 1;
            CHECK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) {
                CHECK(hipMemsetAsync(A.counter, 0, 4, 0));
                CHECK(hipLaunchCooperativeKernel(kp, dim3(nwg), dim3(256), params, 0, 0));
            }
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms_p, e0, e1));
        }
        if (reset_x()) return 1;
        CHECK(hipLaunchCooperativeKernel(kp, dim3(nwg), dim3(256), params, 0, 0));
        CHECK(hipMemcpy(got.data(), A.x[nsweeps & 1][1], n_vec * 8, hipMemcpyDeviceToHost));
        for (int pass = 0; pass < 2; ++pass) {
            CHECK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) {
                CHECK(hipMemsetAsync(A.counter, 0, 4, 0));
                CHECK(hipLaunchCooperativeKernel(kb, dim3(nwg), dim3(256), params, 0, 0));
            }
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms_b, e0, e1));
        }
        int err = 0;
        CHECK(hipMemcpy(&err, A.err, 4, hipMemcpyDeviceToHost));
        double worst = 0.0, scale = 0.0;
        for (int col = 0; col < NCOL; ++col)
            for (int k = 0; k < NZ; ++k) {
                worst = std::fmax(worst, std::fabs(ref[idxA(col, k)] - got[idxA(col, k)]));
                scale = std::fmax(scale, std::fabs(ref[idxA(col, k)]));
            }
        printf("mode %d, %2d sweeps: %d launches %.2f us; one persistent launch %.2f us (memset of the counter included); "
               "%d barriers alone %.2f us; barrier timeout %d; max diff %.3e (scale %.3e)\n",
               mode, nsweeps, nsweeps, 1000.0 * ms_l / reps, 1000.0 * ms_p / reps, nsweeps, 1000.0 * ms_b / reps, err, worst, scale);
        if (err || !(worst <= 1e-12 * scale)) return 2;
    }
    return 0;
}
