// nk2d_krylov.hip -- the Krylov loop itself behind the C ABI:
//
//   nk2d_jvp          ModelStateBase.comp_jacobian_fcn_state_prod, nk_ooc/model_state_base.py:492-527
//   nk2d_gmres_solve  KrylovSolver._solve0 + solve for one tracer module, nk_ooc/krylov_solver.py:85-165,
//                     with _comp_krylov_basis_coeffs (:168-181) as a Givens QR of the Hessenberg
//   nk2d_multi_dot / nk2d_multi_axpy
//                     fused projections of one vector on a whole basis (one launch, one read-back):
//                     the classical Gram-Schmidt building block of the sharded layouts, where every
//                     read-back is an all-reduce
//
// The vector work is the library's own region-weighted algebra (nk2d_api.hip) and forward year
// (nk2d_radau.hip); this file only sequences it, in the reference's operation order, so that the
// Python mirror of KrylovSolver (krylov_solver.py, which keeps the reference's file trail) and this
// all-device loop produce the same Krylov space.
#include "nk2d_common.h"
#include "nk2d_hostmath.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

__device__ __forceinline__ double bcast_r(const double* __restrict__ coef, int m, double fill) {
    return (m > 0) ? coef[m - 1] : fill;
}

// part[(task * n + i) * nreg + r] = sum over the column's cells of region r+1 of wn * (w * v_i)
template <int E>
__global__ void k_multi_dot(int ncol, int ny, int nreg, int n, const double* __restrict__ w,
                            const double* const* __restrict__ vecs, const double* __restrict__ wn,
                            const int32_t* __restrict__ mask, double* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * NK2D_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (task >= ncol) return;
    const int j = task % ny;
    double ww[E], wgt[E], vv[E];
    int mm[E];
    load_col<E>(w, task, lane, ww);
    load_col<E>(wn, j, lane, wgt);
#pragma unroll
    for (int e = 0; e < E; ++e) mm[e] = mask[(size_t)j * (E * 64) + e * 64 + lane];
    for (int i = 0; i < n; ++i) {
        load_col<E>(vecs[i], task, lane, vv);
#pragma unroll
        for (int e = 0; e < E; ++e) vv[e] = wgt[e] * (ww[e] * vv[e]);   // same products as k_dot
        for (int r = 1; r <= nreg; ++r) {
            double acc = 0.0;
            bool any = false;
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (mm[e] == r) { acc += vv[e]; any = true; }
            double tot = 0.0;
            if (__any(any)) tot = wave_sum(acc);
            if (lane == 0) part[((size_t)task * n + i) * nreg + (r - 1)] = tot;
        }
    }
}

// w <- w - bcast(h_0) v_0 - bcast(h_1) v_1 - ...  (subtractions in index order); fill: value of the
// broadcast coefficients where region_mask <= 0 (the reference's broadcast_region_vals fills 1.0,
// tracer_module_state_base.py:502-515)
template <int E>
__global__ void k_multi_axpy(int ncol, int ny, int nreg, int n, double* __restrict__ w,
                             const double* const* __restrict__ vecs, const double* __restrict__ h,
                             const int32_t* __restrict__ mask, double fill) {
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * NK2D_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (task >= ncol) return;
    const int j = task % ny;
    double ww[E], vv[E];
    int mm[E];
    load_col<E>(w, task, lane, ww);
#pragma unroll
    for (int e = 0; e < E; ++e) mm[e] = mask[(size_t)j * (E * 64) + e * 64 + lane];
    for (int i = 0; i < n; ++i) {
        load_col<E>(vecs[i], task, lane, vv);
#pragma unroll
        for (int e = 0; e < E; ++e) ww[e] = ww[e] - bcast_r(h + (size_t)i * nreg, mm[e], fill) * vv[e];
    }
    store_col<E>(w, task, lane, ww);
}

// fixed-order reduction of [ntasks][nout] partials, one thread block per output
__global__ void k_reduce_cols(const double* __restrict__ part, int ntasks, int nout, double* __restrict__ out) {
    __shared__ double sh[NK2D_BLOCK];
    const int r = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < ntasks; i += NK2D_BLOCK) s += part[(size_t)i * nout + r];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = NK2D_BLOCK / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[r] = sh[0];
}

struct Scratch {
    double* part = nullptr;      // [ncol][n][nreg]
    double* red = nullptr;       // [n][nreg]
    double* hred = nullptr;      // pinned
    double* coef = nullptr;      // [n][nreg] + pointers
    double* hcoef = nullptr;     // pinned
    size_t cap = 0;              // n * nreg capacity
    nk2d_ctx* c = nullptr;
    ~Scratch() {
        if (part) (void)hipFree(part);
        if (red) (void)hipFree(red);
        if (coef) (void)hipFree(coef);
        if (hred) (void)hipHostFree(hred);
        if (hcoef) (void)hipHostFree(hcoef);
    }
};

int scratch_alloc(nk2d_ctx* c, Scratch& s, int n) {
    s.c = c;
    s.cap = (size_t)n * c->nreg;
    NK2D_CHECK(c, hipMalloc((void**)&s.part, sizeof(double) * s.cap * c->ncol));
    NK2D_CHECK(c, hipMalloc((void**)&s.red, sizeof(double) * s.cap));
    NK2D_CHECK(c, hipHostMalloc((void**)&s.hred, sizeof(double) * s.cap));
    NK2D_CHECK(c, hipMalloc((void**)&s.coef, sizeof(double) * (s.cap + n)));
    NK2D_CHECK(c, hipHostMalloc((void**)&s.hcoef, sizeof(double) * (s.cap + n)));
    return 0;
}

// pointers (8-byte slots) behind the coefficients, as nk2d_lin_comb stages them
int stage_ptrs(nk2d_ctx* c, Scratch& s, int n, const nk2d_vec* vecs, const double* h) {
    const size_t nco = (size_t)n * c->nreg;
    if (h) std::memcpy(s.hcoef, h, sizeof(double) * nco);
    std::memcpy(s.hcoef + nco, vecs, sizeof(double) * n);
    NK2D_CHECK(c, hipMemcpyAsync(s.coef, s.hcoef, sizeof(double) * (nco + n), hipMemcpyHostToDevice, nk2d_s(c)));
    return 0;
}

// least squares of the Hessenberg: nk2d_hostmath.h (built and run under AddressSanitizer on the CPU, `make asan-host`)
void hessenberg_lstsq(int ncols, const std::vector<double>& H, int ld, double beta, double* coef) {
    nk2d_hm_hessenberg_lstsq(ncols, H, ld, beta, coef);
}

struct VecSet {
    nk2d_ctx* c;
    std::vector<nk2d_vec> v;
    explicit VecSet(nk2d_ctx* ctx) : c(ctx) {}
    ~VecSet() {
        for (nk2d_vec p : v) (void)nk2d_vec_free(c, p);
    }
    int add(nk2d_vec* out) {
        nk2d_vec p = nullptr;
        NK2D_TRY(nk2d_vec_alloc(c, &p));
        v.push_back(p);
        *out = p;
        return 0;
    }
};

}  // namespace

extern "C" int nk2d_multi_dot(nk2d_ctx* c, nk2d_vec w, int32_t n, const nk2d_vec* basis, double* out) {
    NK2D_CHECK(c, hipSetDevice(c->dev));
    if (n < 1 || n > 512) return nk2d_fail(c, "nk2d_multi_dot: n out of range");
    Scratch s;
    NK2D_TRY(scratch_alloc(c, s, n));
    NK2D_TRY(stage_ptrs(c, s, n, basis, nullptr));
    const size_t nco = (size_t)n * c->nreg;
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_multi_dot<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                              c->ncol, c->ny, c->nreg, n, (const double*)w,
                                              (const double* const*)(s.coef + nco), c->WN, c->MASK, s.part));
    hipLaunchKernelGGL(k_reduce_cols, dim3((unsigned)nco), dim3(NK2D_BLOCK), 0, nk2d_s(c), s.part, c->ncol, (int)nco, s.red);
    NK2D_CHECK(c, hipGetLastError());
    NK2D_CHECK(c, hipMemcpyAsync(s.hred, s.red, sizeof(double) * nco, hipMemcpyDeviceToHost, nk2d_s(c)));
    NK2D_CHECK(c, hipStreamSynchronize(nk2d_s(c)));
    std::memcpy(out, s.hred, sizeof(double) * nco);
    return 0;
}

extern "C" int nk2d_multi_axpy(nk2d_ctx* c, nk2d_vec w, int32_t n, const nk2d_vec* basis, const double* h, double fill) {
    NK2D_CHECK(c, hipSetDevice(c->dev));
    if (n < 1 || n > 512) return nk2d_fail(c, "nk2d_multi_axpy: n out of range");
    Scratch s;
    NK2D_TRY(scratch_alloc(c, s, n));
    NK2D_TRY(stage_ptrs(c, s, n, basis, h));
    const size_t nco = (size_t)n * c->nreg;
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_multi_axpy<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                              c->ncol, c->ny, c->nreg, n, (double*)w,
                                              (const double* const*)(s.coef + nco), s.coef, c->MASK, fill));
    NK2D_CHECK(c, hipGetLastError());
    NK2D_CHECK(c, hipStreamSynchronize(nk2d_s(c)));
    return 0;
}

extern "C" int nk2d_jvp(nk2d_ctx* c, nk2d_vec x, nk2d_vec fx, nk2d_vec v, nk2d_vec w, nk2d_vec perturb_fcn,
                        double* sigma_out, nk2d_stats* stats) {
    NK2D_CHECK(c, hipSetDevice(c->dev));
    if (!x || !fx || !v || !w) return nk2d_fail(c, "nk2d_jvp: null vector");
    const int nreg = c->nreg;
    std::vector<double> sigma(nreg), rsig(nreg), one(nreg, 1.0);
    // sigma = 1e-4 * norm(x), 1 where the norm vanishes (model_state_base.py:509-512)
    NK2D_TRY(nk2d_dot(c, x, x, sigma.data()));
    for (int r = 0; r < nreg; ++r) {
        sigma[r] = 1.0e-4 * std::sqrt(sigma[r]);
        if (sigma[r] == 0.0) sigma[r] = 1.0;
        rsig[r] = 1.0 / sigma[r];      // division by an ndarray is multiplication by the reciprocal (:296-301)
    }
    VecSet tmp(c);
    nk2d_vec xp = nullptr, fp = perturb_fcn;
    NK2D_TRY(tmp.add(&xp));
    if (!fp) NK2D_TRY(tmp.add(&fp));
    // perturb_ms = self + sigma * direction (:515)
    NK2D_TRY(nk2d_axpby(c, xp, one.data(), x, sigma.data(), v));
    // the perturbed year: free-running, or -- with a schedule installed (nk2d_set_frozen_schedule: the accepted steps
    // of the year that produced fx) -- on exactly those steps, so that w is the derivative of ONE discrete map
    // instead of the difference of two maps whose adaptive controllers took different decisions
    int rc = -7;
    if (!c->frozen_sched.empty()) {
        rc = nk2d_radau_year(c, xp, fp, stats, c->frozen_sched.data(),
                             (int64_t)(c->frozen_sched.size() / NK2D_SCHED_WIDTH), nullptr, 0, nullptr, true);
        // -7: the recorded iteration counts do not converge for the perturbed state even after the resumes, or its error
        // estimates are out of bounds; -8: the schedule is not this context's (fingerprint) -- a free-running year instead
        // (-7 counted, nk2d_frozen_fallbacks); not with a norm hook, where every shard would have to fall back together
        if (rc != 0 && ((rc != -7 && rc != -8) || c->norm_hook)) return rc;
    }
    if (rc == -7 || rc == -8) NK2D_TRY(nk2d_radau_year(c, xp, fp, stats, nullptr, 0, nullptr, 0, nullptr));
    // (perturb_fcn - fcn) / sigma (:523)
    NK2D_TRY(nk2d_diff_scale(c, w, fp, fx, rsig.data()));
    if (sigma_out) std::memcpy(sigma_out, sigma.data(), sizeof(double) * nreg);
    return 0;
}

extern "C" int nk2d_gmres_solve(nk2d_ctx* c, nk2d_vec x, nk2d_vec fx, double rel_tol, int32_t min_iter,
                                int32_t max_iter, nk2d_vec increment, double* beta_out, double* h_out,
                                double* resid_out, double* coeff_out, int32_t* iters) {
    NK2D_CHECK(c, hipSetDevice(c->dev));
    if (!x || !fx || !increment) return nk2d_fail(c, "nk2d_gmres_solve: null vector");
    if (max_iter < 1 || max_iter > 256) return nk2d_fail(c, "nk2d_gmres_solve: 1 <= max_iter <= 256");
    if (c->kind == 1)
        return nk2d_fail(c, "nk2d_gmres_solve: the phosphorus preconditioner is assembled above the C ABI "
                            "(phosphorus.py); use the KrylovSolver mirror");
    const int nreg = c->nreg, ld = max_iter;
    if (iters) *iters = 0;
    std::vector<double> beta(nreg), rbeta(nreg), one(nreg, 1.0);
    // per region: Hessenberg [max_iter + 1][max_iter] row major
    std::vector<std::vector<double>> H(nreg, std::vector<double>((size_t)(max_iter + 1) * ld, 0.0));
    std::vector<double> hcol((size_t)(max_iter + 1) * nreg), hnorm(nreg), coef((size_t)max_iter * nreg, 0.0),
        rn(nreg);
    VecSet pool(c);
    std::vector<nk2d_vec> V, W;
    nk2d_vec r0 = nullptr, wraw = nullptr, w = nullptr, resid = nullptr;
    NK2D_TRY(pool.add(&r0));
    NK2D_TRY(pool.add(&wraw));
    NK2D_TRY(pool.add(&resid));
    // _solve0 (krylov_solver.py:85-101): r0 = M^-1 fcn, beta = norm(r0), v_0 = -r0 / beta
    NK2D_TRY(nk2d_precond_apply(c, fx, r0));
    NK2D_TRY(nk2d_dot(c, r0, r0, beta.data()));
    for (int r = 0; r < nreg; ++r) {
        beta[r] = std::sqrt(beta[r]);
        rbeta[r] = -(1.0 / beta[r]);       // (-r0) * (1 / beta): the sign commutes with the rounding
    }
    {
        nk2d_vec v0 = nullptr;
        NK2D_TRY(pool.add(&v0));
        NK2D_TRY(nk2d_scale(c, v0, r0, rbeta.data()));
        V.push_back(v0);
    }
    if (beta_out) std::memcpy(beta_out, beta.data(), sizeof(double) * nreg);
    if (h_out) std::memset(h_out, 0, sizeof(double) * (size_t)(max_iter + 1) * max_iter * nreg);
    if (resid_out) std::memset(resid_out, 0, sizeof(double) * (size_t)max_iter * nreg);
    if (coeff_out) std::memset(coeff_out, 0, sizeof(double) * (size_t)max_iter * nreg);
    int j = 0;
    for (;; ++j) {
        // w_raw = J v_j by finite differences, w = M^-1 w_raw (:127-132)
        NK2D_TRY(nk2d_jvp(c, x, fx, V[j], wraw, nullptr, nullptr, nullptr));
        NK2D_TRY(pool.add(&w));
        NK2D_TRY(nk2d_precond_apply(c, wraw, w));
        {
            nk2d_vec keep = nullptr;     // the residual needs w before orthogonalisation (:149-152)
            NK2D_TRY(pool.add(&keep));
            NK2D_TRY(nk2d_vec_copy(c, keep, w));
            W.push_back(keep);
        }
        // modified Gram-Schmidt against v_0..v_j, then the norm (:133-135)
        NK2D_TRY(nk2d_mgs(c, w, j + 1, V.data(), hcol.data()));
        NK2D_TRY(nk2d_dot(c, w, w, hnorm.data()));
        for (int r = 0; r < nreg; ++r) {
            for (int i = 0; i <= j; ++i) H[r][(size_t)i * ld + j] = hcol[(size_t)i * nreg + r];
            hnorm[r] = std::sqrt(hnorm[r]);
            H[r][(size_t)(j + 1) * ld + j] = hnorm[r];
        }
        // coefficients per region (:140, 168-181)
        std::vector<double> cj((size_t)(j + 1) * nreg);
        for (int r = 0; r < nreg; ++r) {
            std::vector<double> cr(j + 1, 0.0);
            hessenberg_lstsq(j + 1, H[r], ld, beta[r], cr.data());
            for (int i = 0; i <= j; ++i) cj[(size_t)i * nreg + r] = cr[i];
        }
        // x_j = sum c_i v_i;  resid = sum c_i w_i + M^-1 fcn (:144-153)
        NK2D_TRY(nk2d_lin_comb(c, increment, j + 1, V.data(), cj.data()));
        NK2D_TRY(nk2d_lin_comb(c, resid, j + 1, W.data(), cj.data()));
        NK2D_TRY(nk2d_axpby(c, resid, one.data(), resid, one.data(), r0));
        NK2D_TRY(nk2d_dot(c, resid, resid, rn.data()));
        bool all_ok = true;
        for (int r = 0; r < nreg; ++r) {
            rn[r] = std::sqrt(rn[r]);
            if (!(j + 1 >= min_iter && rn[r] < rel_tol * beta[r])) all_ok = false;
        }
        if (h_out)
            for (int r = 0; r < nreg; ++r)
                for (int i = 0; i <= j + 1; ++i) h_out[((size_t)i * max_iter + j) * nreg + r] = H[r][(size_t)i * ld + j];
        if (resid_out) std::memcpy(resid_out + (size_t)j * nreg, rn.data(), sizeof(double) * nreg);
        if (coeff_out) {
            std::memset(coeff_out, 0, sizeof(double) * (size_t)max_iter * nreg);
            std::memcpy(coeff_out, cj.data(), sizeof(double) * (size_t)(j + 1) * nreg);
        }
        if (iters) *iters = j + 1;
        if (all_ok || j + 1 >= max_iter) break;
        // v_{j+1} = w / norm(w) (:135, 163)
        std::vector<double> rh(nreg);
        for (int r = 0; r < nreg; ++r) rh[r] = 1.0 / hnorm[r];
        NK2D_TRY(nk2d_scale(c, w, w, rh.data()));
        V.push_back(w);
    }
    NK2D_CHECK(c, hipStreamSynchronize(nk2d_s(c)));
    return 0;
}
