// Host arithmetic of libnk2d.so (csrc/nk2d_hostmath.h) under AddressSanitizer + UndefinedBehaviorSanitizer on
// the CPU: `make -C newton-krylov_ooc_amd/csrc asan-host` (tests/test_host.py runs it).  Known answers only --
// the numerical parity of the same functions is covered through the library by the GPU tests.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "nk2d_hostmath.h"

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

int main() {
    // np.interp semantics: clamped ends, exact knots, interior slope form
    const double xp[4] = {1.0, 2.0, 4.0, 8.0}, fp[4] = {0.0, 1.0, 1.0, 0.0};
    double v = -1.0;
    nk2d_hm_interp(4, xp, fp, 0.5, &v); CHECK(v == 0.0);
    nk2d_hm_interp(4, xp, fp, 9.0, &v); CHECK(v == 0.0);
    nk2d_hm_interp(4, xp, fp, 2.0, &v); CHECK(v == 1.0);
    nk2d_hm_interp(4, xp, fp, 1.5, &v); CHECK(v == 0.5);
    nk2d_hm_interp(4, xp, fp, 6.0, &v); CHECK(v == 0.5);
    nk2d_hm_interp(4, xp, fp, 8.0, &v); CHECK(v == 0.0);
    // interp1d bracket: extrapolating end intervals
    int lo = -1; double dx = 0, den = 0;
    nk2d_hm_bracket(4, xp, 0.0, &lo, &dx, &den); CHECK(lo == 0 && dx == -1.0 && den == 1.0);
    nk2d_hm_bracket(4, xp, 3.0, &lo, &dx, &den); CHECK(lo == 1 && dx == 1.0 && den == 2.0);
    nk2d_hm_bracket(4, xp, 100.0, &lo, &dx, &den); CHECK(lo == 2 && dx == 96.0 && den == 4.0);
    nk2d_hm_bracket(4, xp, 2.0, &lo, &dx, &den); CHECK(lo == 0 && dx == 1.0);
    // fixed-order sum: every length from empty to beyond two strides, integers sum exactly
    for (int n = 0; n < 700; n += 7) {
        std::vector<double> part((size_t)n);
        double want = 0.0;
        for (int i = 0; i < n; ++i) { part[(size_t)i] = (double)(i % 17) - 3.0; want += part[(size_t)i]; }
        CHECK(nk2d_hm_part_sum(part.data(), n, 256) == want);
    }
    // Hessenberg least squares: a consistent system is solved exactly, an inconsistent one satisfies the
    // normal equations, a zero column is left at zero
    {
        const int n = 3, ld = 5;
        std::vector<double> H((size_t)(n + 1) * ld, 0.0);
        const double h[4][3] = {{2.0, 1.0, -1.0}, {0.5, 3.0, 0.25}, {0.0, 0.75, 1.5}, {0.0, 0.0, 0.125}};
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) H[(size_t)i * ld + j] = h[i][j];
        double c[3] = {0, 0, 0};
        nk2d_hm_hessenberg_lstsq(n, H, ld, 4.0, c);
        for (int j = 0; j < n; ++j) {   // H^T (beta e1 - H c) = 0
            double g = 0.0;
            for (int i = 0; i < n + 1; ++i) {
                double r = (i == 0) ? 4.0 : 0.0;
                for (int k = 0; k < n; ++k) r -= h[i][k] * c[k];
                g += h[i][j] * r;
            }
            CHECK(std::fabs(g) < 1e-12);
        }
        std::vector<double> Z((size_t)2 * ld, 0.0);
        double z[1] = {7.0};
        nk2d_hm_hessenberg_lstsq(1, Z, ld, 1.0, z);
        CHECK(z[0] == 0.0);
        // the largest size nk2d_gmres_solve accepts
        const int big = 256;
        std::vector<double> B((size_t)(big + 1) * big, 0.0), cb((size_t)big, 0.0);
        for (int j = 0; j < big; ++j) { B[(size_t)j * big + j] = 1.0 + j; B[(size_t)(j + 1) * big + j] = 0.5; }
        nk2d_hm_hessenberg_lstsq(big, B, big, 1.0, cb.data());
        CHECK(std::isfinite(cb[0]) && std::isfinite(cb[(size_t)big - 1]));
    }
    if (failures == 0) std::printf("hostmath ok\n");
    return failures == 0 ? 0 : 1;
}
