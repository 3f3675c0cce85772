// nk2d_hostmath.h -- the pure host arithmetic of libnk2d.so (no HIP types): what the controller and the Krylov
// entry points compute between kernel launches.  Kept apart so that it also builds with a plain C++ compiler:
// `make -C csrc asan-host` compiles tests/host_asan/test_hostmath.cpp against this header with
// -fsanitize=address,undefined and runs it on the CPU (the GPU code cannot run under a sanitizer on this pool).
#pragma once

#include <cmath>
#include <cstddef>
#include <vector>

// np.interp (numpy/core/src/multiarray/compiled_base.c semantics) at one point
static inline int nk2d_hm_interp(int n, const double* xp, const double* fp, double x, double* out) {
    if (x > xp[n - 1]) { *out = fp[n - 1]; return 0; }
    if (x < xp[0]) { *out = fp[0]; return 0; }
    int j = 0;
    while (j + 1 < n && xp[j + 1] <= x) ++j;  // xp[j] <= x < xp[j+1]
    if (j == n - 1 || xp[j] == x) { *out = fp[j]; return 0; }
    const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    *out = slope * (x - xp[j]) + fp[j];
    return 0;
}

// bracketing interval of x in the increasing knots xs[0..n) as scipy's interp1d picks it (searchsorted,
// clipped to [1, n-1]: the end intervals extrapolate)
static inline void nk2d_hm_bracket(int n, const double* xs, double x, int* lo, double* dx, double* den) {
    int hi = 0;
    while (hi < n && xs[hi] < x) ++hi;   // searchsorted(xs, x), side = "left"
    if (hi < 1) hi = 1;
    if (hi > n - 1) hi = n - 1;
    *lo = hi - 1;
    *dx = x - xs[hi - 1];
    *den = xs[hi] - xs[hi - 1];
}

// sum of n partials in the association of the device reduction k_reduce: `block` strided accumulators, then a
// binary tree (block = 256)
static inline double nk2d_hm_part_sum(const double* part, int n, int block) {
    std::vector<double> sh((size_t)block);
    for (int t = 0; t < block; ++t) {
        double acc = 0.0;
        for (int i = t; i < n; i += block) acc += part[i];
        sh[(size_t)t] = acc;
    }
    for (int o = block / 2; o > 0; o >>= 1)
        for (int t = 0; t < o; ++t) sh[(size_t)t] += sh[(size_t)(t + o)];
    return sh[0];
}

// argmin_c || beta e_1 - H c ||_2 for the (ncols + 1) x ncols upper Hessenberg H stored row major with leading
// dimension ld: Givens rotations to upper triangular form, back substitution.  A zero pivot (exact breakdown)
// leaves that coefficient at zero -- the minimum-norm choice np.linalg.lstsq makes for a rank-deficient column.
static inline void nk2d_hm_hessenberg_lstsq(int ncols, const std::vector<double>& H, int ld, double beta, double* coef) {
    const int nrows = ncols + 1;
    std::vector<double> R(H), g((size_t)nrows, 0.0);
    g[0] = beta;
    for (int k = 0; k < ncols; ++k) {
        const double a = R[(size_t)k * ld + k], b = R[(size_t)(k + 1) * ld + k];
        const double r = std::hypot(a, b);
        if (r == 0.0) continue;
        const double cs = a / r, sn = b / r;
        for (int col = k; col < ncols; ++col) {
            const double u = R[(size_t)k * ld + col], v = R[(size_t)(k + 1) * ld + col];
            R[(size_t)k * ld + col] = cs * u + sn * v;
            R[(size_t)(k + 1) * ld + col] = -sn * u + cs * v;
        }
        const double u = g[(size_t)k], v = g[(size_t)k + 1];
        g[(size_t)k] = cs * u + sn * v;
        g[(size_t)k + 1] = -sn * u + cs * v;
    }
    for (int k = ncols - 1; k >= 0; --k) {
        double acc = g[(size_t)k];
        for (int col = k + 1; col < ncols; ++col) acc -= R[(size_t)k * ld + col] * coef[col];
        const double piv = R[(size_t)k * ld + k];
        coef[k] = (piv != 0.0) ? acc / piv : 0.0;
    }
}
