// nk2d_precond.hip -- preconditioner of the iage-type (state independent, linear) modules.
//
// Reference (nk_ooc/py_driver_2d/iage.py:66-93):
//     M^-1 v = (I - A_0 A_1 A_2)^-1 v - v,   A_k = I - dt J(t_k),  dt = T/3, t_k = (k + 1/2) dt.
// In exact arithmetic this is  M^-1 v = -(I - G)^-1 v  with G = A_2^-1 A_1^-1 A_0^-1, i.e.
// the inverse of "three backward-Euler steps around the year minus identity".  The reference
// forms the triple product explicitly; its entries reach (dt kappa/dz^2)^3 ~ 1e21, the
// identity is lost in rounding and the computed result is roundoff dominated beyond toy grids
// (measured in tests/test_oracle_precond.py: a 1e-16 relative perturbation of the product's
// entries moves the reference result by 4e-3 at 26x26 and by 0.7 at 52x52).  This file
// therefore solves the SAME operator in a backward-stable form, the time-periodic system
//
//     [ A_0   0  -I ] [u_1]   [v]
//     [ -I  A_1   0 ] [u_2] = [0] ,      M^-1 v = -(u_3 + v),
//     [  0  -I  A_2 ] [u_3]   [0]
//
// an irreducibly diagonally dominant M-matrix, by block-tridiagonal elimination over the
// ypos columns.  A block is one column at the three time levels (m = 3 nz unknowns); the
// couplings between neighbouring columns are diagonal matrices, so the Schur complements
//     S_j = D_j - diag(l_j) S_{j-1}^-1 diag(u_{j-1})
// need no matrix product, only one dense inversion per column (Gauss-Jordan, no pivoting
// needed for M-matrices).  The explicit inverses S_j^-1 are kept in HBM ([tc][ny][m][m]
// doubles: 10.4 GB for iage at 416x416 -- sized for the 288 GB of an MI355X) and an apply is
// 2 ny dense matrix-vector products streamed from HBM at full chip width.
#include "nk2d_common.h"

#include <algorithm>
#include <vector>

namespace {

struct Precond {
    int m, nb, nz, nt, tc;
    int mode;      // 0: time-periodic system of the linear modules, 1: shifted systems (nk2d_shift_factor)
    int nsys;      // independent systems held in SINV: tracers (mode 0) or shifts (mode 1)
    int cap_sys;   // systems the buffers were allocated for
    double scale, sigma[NK2D_MAX_SHIFTS];
    double* PJ;    // Jacobian planes, natural layout [nt][6][nz][ny]  (L, S, C, N, U, d uptake / d po4)
    double* SINV;  // [tc][nb][m][m]
    // option "pc_fp32": the explicit inverses are KEPT in single precision (half the HBM: 832 x 832 fits an MI355X); the
    // elimination itself stays in double precision -- the inverse of the column before is held once more in PREV -- and an
    // apply is refined once against the exact block tridiagonal operator (k_pc_residual), which brings it back to the
    // accuracy of the double precision inverses
    float* SINV32; // [tc][nb][m][m]
    double* PREV;  // [tc][m][m] Schur inverse of the column before, double precision
    double* RV;    // right-hand sides kept for the residual [tc][nb][m]
    double* X0;    // first solution [tc][nb][m]
    int fp32;
    double* BUF;   // Gauss-Jordan ping-pong [2][tc][m][m]
    double* PINV;  // inverses of this and the next panel step's pivot block [2][tc][32][32] (k_pc_gj_step)
    double* ROWS;  // scaled pivot rows    [tc][NB][m]
    double* YV;    // forward-sweep vectors [tc][nb][m]
    double* XV;    // solution vectors      [tc][nb][m]
    double dt;
};

// planes index
enum { PL_L = 0, PL_S = 1, PL_C = 2, PL_N = 3, PL_U = 4, PL_UPR = 5, PL_COUNT = 6 };

struct PcDev {
    int m, nb, nz, ny, nt, tc, mode, kind, E;
    const double* PJ;
    const double* DZR;   // packed depth.delta_r
    double dt;
    double scale, sigma[NK2D_MAX_SHIFTS];
    double surf[NK2D_MAX_TRACERS], decay[NK2D_MAX_TRACERS];
    double ph_sig, ph_rd, ph_rp, ph_vs;
};

__device__ __forceinline__ double pj(const PcDev& P, int tau, int pl, int k, int j) {
    return P.PJ[(((size_t)tau * PL_COUNT + pl) * P.nz + k) * P.ny + j];
}

__device__ __forceinline__ double dzr_at(const PcDev& P, int k) { return P.DZR[(size_t)(k % P.E) * 64 + k / P.E]; }

// coupling of unknown (slot, k) of block j to the same unknown of block j-1 (lat_l) and of
// unknown (slot, k) of block j to block j+1 (lat_u): mode 0 blocks hold I - dt J at time level
// `slot`, mode 1 blocks hold scale J (one set of planes) for tracer `slot`
__device__ __forceinline__ double lat_l(const PcDev& P, int slot, int k, int j) {
    return (P.mode == 0) ? -(P.dt * pj(P, slot, PL_S, k, j)) : P.scale * pj(P, 0, PL_S, k, j);
}
__device__ __forceinline__ double lat_u(const PcDev& P, int slot, int k, int j) {
    return (P.mode == 0) ? -(P.dt * pj(P, slot, PL_N, k, j)) : P.scale * pj(P, 0, PL_N, k, j);
}

// diagonal block of the shifted system scale J - sigma I at column j: rows / columns are
// (tracer, level); tracers couple through the phosphorus terms only (phosphorus.py:119-170)
__device__ __forceinline__ double shifted_entry(const PcDev& P, int sys, int j, int trr, int k, int trc, int kc) {
    if (trr == trc) {
        if (kc == k) {
            double d = pj(P, 0, PL_C, k, j) - P.decay[trr];
            if (k == 0) d = d - P.surf[trr];
            if (P.kind == 1) {
                if (trr == 0) d = d - pj(P, 0, PL_UPR, k, j);
                else if (trr == 1) d = d - P.ph_rd;
                else d = d - (P.ph_rp + ((k < P.nz - 1) ? P.ph_vs * dzr_at(P, k) : 0.0));
            }
            return P.scale * d - P.sigma[sys];
        }
        if (kc == k - 1) {
            double lo = pj(P, 0, PL_L, k, j);
            if (P.kind == 1 && trr == 2) lo = lo + P.ph_vs * dzr_at(P, k);
            return P.scale * lo;
        }
        if (kc == k + 1) return P.scale * pj(P, 0, PL_U, k, j);
        return 0.0;
    }
    if (P.kind == 1 && kc == k) {
        if (trr == 0) return P.scale * ((trc == 1) ? P.ph_rd : P.ph_rp);
        if (trc == 0) return P.scale * (((trr == 1) ? P.ph_sig : 1.0 - P.ph_sig) * pj(P, 0, PL_UPR, k, j));
    }
    return 0.0;
}

// S_j = D_j - diag(l_j) Sinv_{j-1} diag(u_{j-1});  one thread per entry (r, c)
__global__ void k_pc_schur(PcDev P, int j, const double* __restrict__ sinv_prev, size_t prev_sys_stride,
                           double* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    const int tr = blockIdx.z;
    if (c >= P.m) return;
    const int tau = r / P.nz, k = r - tau * P.nz;
    const int tc_ = c / P.nz, kc = c - tc_ * P.nz;
    double val = 0.0;
    if (P.mode == 1) {
        val = shifted_entry(P, tr, j, tau, k, tc_, kc);
    } else if (tc_ == tau) {
        if (kc == k) {
            double jc = pj(P, tau, PL_C, k, j) - P.decay[tr];
            if (k == 0) jc = jc - P.surf[tr];
            if (P.kind == 2) jc = jc - pj(P, tau, PL_UPR, k, j);   // sink threshold of the forced module
            val = 1.0 - P.dt * jc;
        } else if (kc == k - 1) {
            val = -(P.dt * pj(P, tau, PL_L, k, j));
        } else if (kc == k + 1) {
            val = -(P.dt * pj(P, tau, PL_U, k, j));
        }
    } else if (kc == k && tc_ == (tau + P.nt - 1) % P.nt) {
        val = -1.0;
    }
    if (sinv_prev) {
        // l_j[r] = -dt JS[tau][k][j] (coupling to column j-1), u_{j-1}[c] = -dt JN[tc_][kc][j-1]
        const double l = lat_l(P, tau, k, j);
        const double u = lat_u(P, tc_, kc, j - 1);
        val = val - (l * sinv_prev[(size_t)tr * prev_sys_stride + (size_t)r * P.m + c]) * u;
    }
    out[((size_t)tr * P.m + r) * P.m + c] = val;
}

// ---------------------------------------------------------------------------------
// Blocked Gauss-Jordan inversion without pivoting (the Schur complements are M-matrices),
// NB pivots per step, two launches per step, src -> dst ping-pong:
//   k_pc_gj_rows  : R = Pinv * src[pb, :]   with the pivot columns of R replaced by Pinv
//   k_pc_gj_update: dst[i, :] = (src[i, :] with pivot columns zeroed) - src[i, pb] * R   (i not in pb)
//                   dst[pb, :] = R
// ---------------------------------------------------------------------------------
#define PC_NB 32

// R[p][c] for p < nb, c < m.  Every workgroup first inverts the nb x nb pivot block itself (in LDS, the
// Gauss-Jordan without pivoting): redundant arithmetic on 1024 numbers instead of a launch of a
// single workgroup that everything else would wait for.
__global__ void __launch_bounds__(256) k_pc_gj_rows(int m, int p0, int nb, const double* __restrict__ src,
                                                    double* __restrict__ rows) {
    __shared__ double ab[2][PC_NB][PC_NB + 1];   // ping-pong: one barrier per pivot
    const int tr = blockIdx.z;
    const size_t base = (size_t)tr * m * m;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int p = blockIdx.y;
    // this thread's column of the pivot rows, requested first: it arrives while the block is inverted
    double col[PC_NB];
#pragma unroll
    for (int q = 0; q < PC_NB; ++q) col[q] = (c < m && q < nb) ? src[base + (size_t)(p0 + q) * m + c] : 0.0;
    // the thread's four entries of the pivot block: column cc, rows r0 + 8 q
    const int r0 = threadIdx.x / PC_NB, cc = threadIdx.x - r0 * PC_NB;
    double mine[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r0 + 8 * q;
        mine[q] = (r < nb && cc < nb) ? src[base + (size_t)(p0 + r) * m + (p0 + cc)] : ((r == cc) ? 1.0 : 0.0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) ab[0][r0 + 8 * q][cc] = mine[q];
    __syncthreads();
    int cur = 0;
    for (int pp = 0; pp < nb; ++pp) {
        double (*a)[PC_NB + 1] = ab[cur];
        const double piv = 1.0 / a[pp][pp];
        const double prow = ((cc == pp) ? 1.0 : a[pp][cc]) * piv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + 8 * q;
            const double f = a[r][pp];
            const double old = (cc == pp) ? 0.0 : mine[q];
            mine[q] = (r == pp) ? prow : __builtin_fma(-f, prow, old);
            ab[1 - cur][r][cc] = mine[q];
        }
        cur = 1 - cur;
        __syncthreads();
    }
    double (*a)[PC_NB + 1] = ab[cur];
    if (c >= m) return;
    const double* pi = a[p];
    double val;
    if (c >= p0 && c < p0 + nb) {
        val = pi[c - p0];
    } else {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < PC_NB; ++q) acc = __builtin_fma(pi[q], col[q], acc);   // col[q] = 0 beyond nb
        val = acc;
    }
    rows[((size_t)tr * PC_NB + p) * m + c] = val;
}

// 64 x 64 output tile per workgroup, 4 x 4 outputs per thread, operands staged through LDS
__global__ void __launch_bounds__(256) k_pc_gj_update(int m, int p0, int nb, const double* __restrict__ src,
                                                      const double* __restrict__ rows, double* __restrict__ dst) {
    __shared__ double sf[64][PC_NB + 1];   // src[i, pb]
    __shared__ double sr[PC_NB][64 + 1];   // R[:, c]
    const int tr = blockIdx.z;
    const size_t base = (size_t)tr * m * m;
    const int i0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    const int ty = tid / 16, tx = tid - ty * 16;   // thread owns rows ty*4.., columns tx + 16*k
    // every request of the workgroup goes out before the first use: operands, then the tile itself
    double lf[8], lr[8], tile[4][4];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int idx = tid + 256 * k;
        const int ii = idx / PC_NB, q = idx - ii * PC_NB;
        const int i = i0 + ii;
        lf[k] = (i < m && q < nb) ? src[base + (size_t)i * m + p0 + q] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int idx = tid + 256 * k;
        const int q = idx / 64, cc = idx - q * 64;
        const int c = c0 + cc;
        lr[k] = (c < m && q < nb) ? rows[((size_t)tr * PC_NB + q) * m + c] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = i0 + ty * 4 + a, c = c0 + tx + 16 * b;
            tile[a][b] = (i < m && c < m) ? src[base + (size_t)i * m + c] : 0.0;
        }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int idx = tid + 256 * k;
        sf[idx / PC_NB][idx % PC_NB] = lf[k];
        sr[idx / 64][idx % 64] = lr[k];
    }
    __syncthreads();
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int q = 0; q < nb; ++q) {
        double f[4], r[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) f[a] = sf[ty * 4 + a][q];
#pragma unroll
        for (int b = 0; b < 4; ++b) r[b] = sr[q][tx + 16 * b];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_fma(f[a], r[b], acc[a][b]);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = i0 + ty * 4 + a;
        if (i >= m) continue;
        const bool pivot_row = i >= p0 && i < p0 + nb;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int c = c0 + tx + 16 * b;
            if (c >= m) continue;
            double val;
            if (pivot_row) {
                val = sr[i - p0][tx + 16 * b];
            } else {
                const bool pivot_col = c >= p0 && c < p0 + nb;
                val = (pivot_col ? 0.0 : tile[a][b]) - acc[a][b];
            }
            dst[base + (size_t)i * m + c] = val;
        }
    }
}

// The same rank-nb update on the matrix cores: v_mfma_f64_16x16x4_f64.  A wave owns a 32 x 32 quadrant of the
// 64 x 64 tile = 2 x 2 MFMA tiles, 8 k-steps of 4 pivots each: 32 MFMAs per wave instead of 512 VALU FMAs and a
// quarter of the LDS reads.  Operand lane maps (cdna_hip_programming.md section 3, f64): A[i = lane & 15][k = lane >> 4],
// B[k = lane >> 4][j = lane & 15], one double each; C/D row = (lane >> 4) + 4 reg, col = lane & 15.  On gfx950 the
// fp64 matrix rate equals the fp64 vector rate, so this buys instruction and LDS slots, not flops -- the kernel
// streams the m x m matrix through HBM once per nb pivots either way (section 4 of DESIGN.md).
typedef double pc_v4d __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_pc_gj_update_mfma(int m, int p0, int nb, const double* __restrict__ src,
                                                           const double* __restrict__ rows, double* __restrict__ dst) {
    __shared__ double sf[64][PC_NB + 1];   // src[i, pb]
    __shared__ double sr[PC_NB][64 + 1];   // R[:, c]
    const int tr = blockIdx.z;
    const size_t base = (size_t)tr * m * m;
    const int i0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    const int wv = tid >> 6, l = tid & 63;
    const int ib = (wv >> 1) * 32, jb = (wv & 1) * 32;
    const int lr16 = l >> 4, lc16 = l & 15;
    double lf[8], lrr[8], tile[2][2][4];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int idx = tid + 256 * k;
        const int ii = idx / PC_NB, q = idx - ii * PC_NB;
        const int i = i0 + ii;
        lf[k] = (i < m && q < nb) ? src[base + (size_t)i * m + p0 + q] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int idx = tid + 256 * k;
        const int q = idx / 64, cc = idx - q * 64;
        const int c = c0 + cc;
        lrr[k] = (c < m && q < nb) ? rows[((size_t)tr * PC_NB + q) * m + c] : 0.0;
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = i0 + ib + 16 * ti + lr16 + 4 * reg, c = c0 + jb + 16 * tj + lc16;
                tile[ti][tj][reg] = (i < m && c < m) ? src[base + (size_t)i * m + c] : 0.0;
            }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int idx = tid + 256 * k;
        sf[idx / PC_NB][idx % PC_NB] = lf[k];
        sr[idx / 64][idx % 64] = lrr[k];
    }
    __syncthreads();
    pc_v4d acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = (pc_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < PC_NB / 4; ++ks) {
        const int k = 4 * ks + lr16;      // pivots beyond nb were staged as zeros
        const double a0 = sf[ib + lc16][k], a1 = sf[ib + 16 + lc16][k];
        const double b0 = sr[k][jb + lc16], b1 = sr[k][jb + 16 + lc16];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = i0 + ib + 16 * ti + lr16 + 4 * reg, c = c0 + jb + 16 * tj + lc16;
                if (i >= m || c >= m) continue;
                double val;
                if (i >= p0 && i < p0 + nb) {
                    val = sr[i - p0][c - c0];
                } else {
                    const bool pivot_col = c >= p0 && c < p0 + nb;
                    val = (pivot_col ? 0.0 : tile[ti][tj][reg]) - acc[ti][tj][reg];
                }
                dst[base + (size_t)i * m + c] = val;
            }
}

// One panel step of the Gauss-Jordan inversion in ONE launch, the pivot block's inverse computed one step AHEAD (option
// "pc_fused").  The two-launch step is bound by what is sequential in it: every workgroup of k_pc_gj_rows inverts the
// 32 x 32 pivot block by itself -- 32 dependent eliminations with a barrier each, 13 of its 16.7 us at m = 1248 -- before
// k_pc_gj_update streams the matrix once more (15 us).  (Fusing the two as they are, every workgroup of the update
// inverting the block while its tile is on its way, was measured first: 800 workgroups at two to a compute unit are two
// rounds of 13 us, 0.53 -> 0.85 s for the set-up at 416 x 416.)  Here a workgroup owns a 64 x 64 tile of the result as in
// the update; it reads the pivot block's inverse from `pinv_in`, scales its 32 x 64 piece of the pivot rows itself (32 fused
// multiply-adds per entry, the order of k_pc_gj_rows) and applies the rank-32 update on the matrix cores.  The ONE
// workgroup whose tile holds the NEXT pivot block -- a 32 x 32 quadrant of it, one wave's -- then inverts that block and
// leaves the inverse in `pinv_out` for the next launch: the sequential part runs once per step, beside the streaming part
// instead of ahead of it (that workgroup is the launch's first).  Same operations in the same order as the two launches:
// the same bits (tests/test_gpu_krylov.py).
__device__ __forceinline__ int pc_invert_block(double (*ab)[PC_NB][PC_NB + 1], double (&mine)[4], int r0, int cc, int nb) {
    int cur = 0;
    for (int pp = 0; pp < nb; ++pp) {
        double (*a)[PC_NB + 1] = ab[cur];
        const double piv = 1.0 / a[pp][pp];
        const double prow = ((cc == pp) ? 1.0 : a[pp][cc]) * piv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + 8 * q;
            const double f = a[r][pp];
            const double old = (cc == pp) ? 0.0 : mine[q];
            mine[q] = (r == pp) ? prow : __builtin_fma(-f, prow, old);
            ab[1 - cur][r][cc] = mine[q];
        }
        cur = 1 - cur;
        __syncthreads();
    }
    return cur;
}

// inverse of the FIRST pivot block of every system (the later ones come out of k_pc_gj_step): pinv [nsys][32][32], identity
// beyond nb
__global__ void __launch_bounds__(256) k_pc_pivot_invert(int m, int p0, int nb, const double* __restrict__ src, double* __restrict__ pinv) {
    __shared__ double ab[2][PC_NB][PC_NB + 1];
    const int tr = blockIdx.z;
    const size_t base = (size_t)tr * m * m;
    const int r0 = threadIdx.x / PC_NB, cc = threadIdx.x - r0 * PC_NB;
    double mine[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r0 + 8 * q;
        mine[q] = (r < nb && cc < nb) ? src[base + (size_t)(p0 + r) * m + (p0 + cc)] : ((r == cc) ? 1.0 : 0.0);
        ab[0][r][cc] = mine[q];
    }
    __syncthreads();
    pc_invert_block(ab, mine, r0, cc, nb);
#pragma unroll
    for (int q = 0; q < 4; ++q) pinv[((size_t)tr * PC_NB + r0 + 8 * q) * PC_NB + cc] = mine[q];
}

__global__ void __launch_bounds__(256) k_pc_gj_step(int m, int p0, int nb, const double* __restrict__ src, double* __restrict__ dst,
                                                    const double* __restrict__ pinv_in, double* __restrict__ pinv_out) {
    __shared__ double sr[PC_NB][64 + 1];         // R[:, c]
    // src[i, pb] and src[pb, c]; once the update has read them, the next pivot block's ping-pong (its workgroup only)
    __shared__ double lds_ops[64 * (PC_NB + 1) + PC_NB * (64 + 1)];
    static_assert(2 * PC_NB * (PC_NB + 1) <= 64 * (PC_NB + 1) + PC_NB * (64 + 1), "the ping-pong must fit the operand staging");
    double (*sf)[PC_NB + 1] = reinterpret_cast<double (*)[PC_NB + 1]>(lds_ops);
    double (*sraw)[64 + 1] = reinterpret_cast<double (*)[64 + 1]>(lds_ops + 64 * (PC_NB + 1));
    double (*ab)[PC_NB][PC_NB + 1] = reinterpret_cast<double (*)[PC_NB][PC_NB + 1]>(lds_ops);
    const int tr = blockIdx.z;
    const size_t base = (size_t)tr * m * m;
    // the next pivot block: rows / columns p1 .. p1 + nb1 of the result, inside tile (tn, tn); its workgroup runs first
    const int p1 = p0 + PC_NB;
    const bool has_next = p1 < m;
    const int nb1 = has_next ? min(PC_NB, m - p1) : 0;
    const int tn = p1 / 64;
    int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    if (has_next) {
        if (bx == 0 && by == 0) { bx = tn; by = tn; }
        else if (bx == tn && by == tn) { bx = 0; by = 0; }
    }
    const bool owns_next = has_next && bx == tn && by == tn;
    const int i0 = by * 64, c0 = bx * 64;
    const int tid = threadIdx.x;
    const int wv = tid >> 6, l = tid & 63;
    const int ib = (wv >> 1) * 32, jb = (wv & 1) * 32;
    const int lr16 = l >> 4, lc16 = l & 15;
    const int r0 = tid / PC_NB, cc = tid - r0 * PC_NB;
    // ---- every request first.  (Holding the other workgroups' requests back behind those of the one with the inversion ahead
    // of it was tried -- s_sleep in units of 0.4 us: every unit is added to the launch, 416 x 416 0.491 -> 0.494 / 0.503 / 0.523 s
    // for 2 / 4 / 6 units: the launch is as long as its streaming part, the inversion is hidden.)
    double lf[8], lrw[8], tile[2][2][4];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int idx = tid + 256 * k;
        const int ii = idx / PC_NB, q = idx - ii * PC_NB;
        const int i = i0 + ii;
        lf[k] = (i < m && q < nb) ? src[base + (size_t)i * m + p0 + q] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int idx = tid + 256 * k;
        const int q = idx / 64, c = c0 + (idx - q * 64);
        lrw[k] = (c < m && q < nb) ? src[base + (size_t)(p0 + q) * m + c] : 0.0;
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = i0 + ib + 16 * ti + lr16 + 4 * reg, c = c0 + jb + 16 * tj + lc16;
                tile[ti][tj][reg] = (i < m && c < m) ? src[base + (size_t)i * m + c] : 0.0;
            }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int idx = tid + 256 * k;
        sf[idx / PC_NB][idx % PC_NB] = lf[k];
        sraw[idx / 64][idx % 64] = lrw[k];
    }
    __syncthreads();
    // ---- this tile's piece of the scaled pivot rows: R[p][c] = sum_q Pinv[p][q] src[p0 + q][c], the pivot columns replaced by
    // Pinv.  A thread owns column c0 + l of rows wave + 4 k: the rows of Pinv it needs are the same for the whole wave -- scalar
    // operands, fetched through the scalar cache -- and an entry of the column is read from LDS once for its eight rows
    {
        const int wvu = __builtin_amdgcn_readfirstlane(wv);
        const double* __restrict__ pin = pinv_in + (size_t)tr * PC_NB * PC_NB;
        double accr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) accr[k] = 0.0;
#pragma unroll
        for (int q = 0; q < PC_NB; ++q) {
            const double sv = sraw[q][l];                                                   // (zeros beyond nb)
#pragma unroll
            for (int k = 0; k < 8; ++k) accr[k] = __builtin_fma(pin[(wvu + 4 * k) * PC_NB + q], sv, accr[k]);
        }
        const int c = c0 + l;
        const bool pivot_col = c >= p0 && c < p0 + nb;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p = wvu + 4 * k;
            const double val = pivot_col ? pin[p * PC_NB + (c - p0)] : accr[k];
            sr[p][l] = (p < nb) ? val : 0.0;
        }
    }
    __syncthreads();
    // ---- the rank-nb update (k_pc_gj_update_mfma)
    pc_v4d acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = (pc_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < PC_NB / 4; ++ks) {
        const int k = 4 * ks + lr16;
        const double a0 = sf[ib + lc16][k], a1 = sf[ib + 16 + lc16][k];
        const double b0 = sr[k][jb + lc16], b1 = sr[k][jb + 16 + lc16];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    // (the wave whose quadrant is the next pivot block keeps what it stores -- where the operands were staged, once every
    // wave of the workgroup has read them)
    const bool my_quadrant = owns_next && ib == p1 - i0 && jb == p1 - c0;
    if (owns_next) __syncthreads();
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int il = 16 * ti + lr16 + 4 * reg, cl = 16 * tj + lc16;
                const int i = i0 + ib + il, c = c0 + jb + cl;
                double val = (il == cl) ? 1.0 : 0.0;        // (beyond the matrix: the identity the pivot block is padded with)
                if (i < m && c < m) {
                    if (i >= p0 && i < p0 + nb) {
                        val = sr[i - p0][c - c0];
                    } else {
                        const bool pivot_col = c >= p0 && c < p0 + nb;
                        val = (pivot_col ? 0.0 : tile[ti][tj][reg]) - acc[ti][tj][reg];
                    }
                    dst[base + (size_t)i * m + c] = val;
                }
                if (my_quadrant) ab[0][il][cl] = val;
            }
    if (!owns_next) return;
    // ---- the next step's pivot block, inverted here (k_pc_gj_rows' elimination, once instead of by every workgroup)
    __syncthreads();
    double mine[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) mine[q] = ab[0][r0 + 8 * q][cc];
    __syncthreads();
    pc_invert_block(ab, mine, r0, cc, nb1);
#pragma unroll
    for (int q = 0; q < 4; ++q) pinv_out[((size_t)tr * PC_NB + r0 + 8 * q) * PC_NB + cc] = mine[q];
}

// Dense mat-vec of the block substitution for even m: 16-byte loads, the whole row of a wave requested at once
// (up to 10 x 1 KB per wave in flight), the vector staged once per workgroup in LDS instead of being re-read
// through L1 by every wave.  Same epilogues as k_pc_gemv below.
#define PC_GEMV_CHUNK 10
#define PC_GEMV_XMAX 4096
__global__ void __launch_bounds__(256) k_pc_gemv2(PcDev P, int mode, int j, const double* __restrict__ M,
                                                  size_t m_tr_stride, const double* __restrict__ a,
                                                  const double* __restrict__ rhs, size_t v_tr_stride,
                                                  double* __restrict__ out, double* __restrict__ prev) {
    __shared__ double2 xs[PC_GEMV_XMAX / 2];
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int tr = blockIdx.y;
    const int m2 = P.m >> 1;
    const double2* av2 = reinterpret_cast<const double2*>(a + (size_t)tr * v_tr_stride);
    const bool live = r < P.m;
    const double2* row2 = reinterpret_cast<const double2*>(M + (size_t)tr * m_tr_stride + (size_t)(live ? r : 0) * P.m);
    // the matrix row first (it comes from HBM), then the vector (L2)
    double2 mv[PC_GEMV_CHUNK];
#pragma unroll
    for (int q = 0; q < PC_GEMV_CHUNK; ++q) {
        const int c = lane + 64 * q;
        mv[q] = (live && c < m2) ? row2[c] : make_double2(0.0, 0.0);
    }
    for (int i = threadIdx.x; i < m2; i += blockDim.x) xs[i] = av2[i];
    __syncthreads();
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < PC_GEMV_CHUNK; ++q) {
        const int c = lane + 64 * q;
        const double2 xv = (c < m2) ? xs[c] : make_double2(0.0, 0.0);
        acc[(2 * q) & 3] = __builtin_fma(mv[q].x, xv.x, acc[(2 * q) & 3]);
        acc[(2 * q + 1) & 3] = __builtin_fma(mv[q].y, xv.y, acc[(2 * q + 1) & 3]);
    }
    for (int c0 = 64 * PC_GEMV_CHUNK; c0 < m2; c0 += 64 * PC_GEMV_CHUNK) {   // rows longer than one chunk
        double2 mw[PC_GEMV_CHUNK];
#pragma unroll
        for (int q = 0; q < PC_GEMV_CHUNK; ++q) {
            const int c = c0 + lane + 64 * q;
            mw[q] = (live && c < m2) ? row2[c] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int q = 0; q < PC_GEMV_CHUNK; ++q) {
            const int c = c0 + lane + 64 * q;
            const double2 xv = (c < m2) ? xs[c] : make_double2(0.0, 0.0);
            acc[(2 * q) & 3] = __builtin_fma(mw[q].x, xv.x, acc[(2 * q) & 3]);
            acc[(2 * q + 1) & 3] = __builtin_fma(mw[q].y, xv.y, acc[(2 * q + 1) & 3]);
        }
    }
    if (!live) return;
    const double sum = wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
    if (lane == 0) {
        const int slot = r / P.nz, k = r - slot * P.nz;
        const size_t at = (size_t)tr * v_tr_stride + r;
        if (mode == 0) {
            out[at] = rhs[at] - lat_l(P, slot, k, j) * sum;
        } else {
            out[at] = sum;
            if (prev) prev[at] = prev[at] - lat_u(P, slot, k, j - 1) * sum;
        }
    }
}

// dense mat-vec with the block-Thomas epilogues; one wave per row, eight 512-byte requests per wave in flight
//   mode 0 (forward):  out[r] = rhs[r] - l[r] * sum_c M[r][c] a[c]
//   mode 1 (backward): out[r] = x_j[r] = sum_c M[r][c] a[c], where a = y_j - U_j x_{j+1} was left behind by the
//                      launch of block j+1: this one turns y_{j-1} into y_{j-1} - U_{j-1} x_j in place (prev; the
//                      couplings are diagonal, every row touches its own entry only)
__global__ void __launch_bounds__(256) k_pc_gemv(PcDev P, int mode, int j, const double* __restrict__ M,
                                                 size_t m_tr_stride, const double* __restrict__ a,
                                                 const double* __restrict__ rhs, size_t v_tr_stride,
                                                 double* __restrict__ out, double* __restrict__ prev) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int tr = blockIdx.y;
    if (r >= P.m) return;
    const double* row = M + (size_t)tr * m_tr_stride + (size_t)r * P.m;
    const double* av = a + (size_t)tr * v_tr_stride;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int c0 = lane; c0 < P.m; c0 += 512) {
        double mv[8], xv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = c0 + 64 * q;
            const bool in = c < P.m;
            mv[q] = in ? row[c] : 0.0;
            xv[q] = in ? av[c] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q & 3] = __builtin_fma(mv[q], xv[q], acc[q & 3]);
    }
    const double sum = wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
    if (lane == 0) {
        const int slot = r / P.nz, k = r - slot * P.nz;
        const size_t at = (size_t)tr * v_tr_stride + r;
        if (mode == 0) {
            out[at] = rhs[at] - lat_l(P, slot, k, j) * sum;
        } else {
            out[at] = sum;
            if (prev) prev[at] = prev[at] - lat_u(P, slot, k, j - 1) * sum;
        }
    }
}

// the same mat-vec on single precision matrices (option "pc_fp32"): 4-byte loads widened in registers, double precision
// accumulation -- half the bytes of the stream that bounds an apply
__global__ void __launch_bounds__(256) k_pc_gemv32(PcDev P, int mode, int j, const float* __restrict__ M,
                                                   size_t m_tr_stride, const double* __restrict__ a,
                                                   const double* __restrict__ rhs, size_t v_tr_stride,
                                                   double* __restrict__ out, double* __restrict__ prev) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int tr = blockIdx.y;
    if (r >= P.m) return;
    const float* row = M + (size_t)tr * m_tr_stride + (size_t)r * P.m;
    const double* av = a + (size_t)tr * v_tr_stride;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int c0 = lane; c0 < P.m; c0 += 1024) {
        float mv[16];
        double xv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = c0 + 64 * q;
            const bool in = c < P.m;
            mv[q] = in ? row[c] : 0.0f;
            xv[q] = in ? av[c] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q & 3] = __builtin_fma((double)mv[q], xv[q], acc[q & 3]);
    }
    const double sum = wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
    if (lane == 0) {
        const int slot = r / P.nz, k = r - slot * P.nz;
        const size_t at = (size_t)tr * v_tr_stride + r;
        if (mode == 0) {
            out[at] = rhs[at] - lat_l(P, slot, k, j) * sum;
        } else {
            out[at] = sum;
            if (prev) prev[at] = prev[at] - lat_u(P, slot, k, j - 1) * sum;
        }
    }
}

__global__ void k_pc_to_f32(const double* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

// res = rhs - A x for the time-periodic block system of mode 0 (the rows of k_pc_schur without the Schur term, and the
// diagonal couplings to the neighbouring columns); one thread per unknown (tracer, column, (time level, depth level))
__global__ void k_pc_residual(PcDev P, const double* __restrict__ rhs, const double* __restrict__ x, double* __restrict__ res) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, tr = blockIdx.z;
    if (r >= P.m) return;
    const int tau = r / P.nz, k = r - tau * P.nz;
    const size_t base = ((size_t)tr * P.nb + j) * P.m;
    const double* xj = x + base;
    double jc = pj(P, tau, PL_C, k, j) - P.decay[tr];
    if (k == 0) jc = jc - P.surf[tr];
    if (P.kind == 2) jc = jc - pj(P, tau, PL_UPR, k, j);
    double ax = (1.0 - P.dt * jc) * xj[r];
    if (k > 0) ax -= (P.dt * pj(P, tau, PL_L, k, j)) * xj[r - 1];
    if (k < P.nz - 1) ax -= (P.dt * pj(P, tau, PL_U, k, j)) * xj[r + 1];
    ax -= xj[((tau + P.nt - 1) % P.nt) * P.nz + k];
    if (j > 0) ax += lat_l(P, tau, k, j) * x[base - P.m + r];
    if (j < P.nb - 1) ax += lat_u(P, tau, k, j) * x[base + P.m + r];
    res[base + r] = rhs[base + r] - ax;
}

// res = rhs - A x for ONE shifted system of mode 1, A = scale J - sigma[sys] I with the tracers of a column in one block (the
// rows of k_pc_schur's shifted_entry and the lateral couplings); rhs, x, res: that system's [nb][m]; one thread per unknown
__global__ void k_pc_residual_shift(PcDev P, int sys, const double* __restrict__ rhs, const double* __restrict__ x,
                                    double* __restrict__ res) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (r >= P.m) return;
    const int trr = r / P.nz, k = r - trr * P.nz;
    const size_t base = (size_t)j * P.m;
    const double* xj = x + base;
    double ax = shifted_entry(P, sys, j, trr, k, trr, k) * xj[r];
    if (k > 0) ax += shifted_entry(P, sys, j, trr, k, trr, k - 1) * xj[r - 1];
    if (k < P.nz - 1) ax += shifted_entry(P, sys, j, trr, k, trr, k + 1) * xj[r + 1];
    for (int trc = 0; trc < P.tc; ++trc)
        if (trc != trr) ax += shifted_entry(P, sys, j, trr, k, trc, k) * xj[trc * P.nz + k];
    if (j > 0) ax += lat_l(P, trr, k, j) * x[base - P.m + r];
    if (j < P.nb - 1) ax += lat_u(P, trr, k, j) * x[base + P.m + r];
    res[base + r] = rhs[base + r] - ax;
}

__global__ void k_pc_add(double* __restrict__ x, const double* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x[i] + y[i];
}

// packed state v -> right-hand sides [tc][nb][m] (time level 0 rows = v, others 0)
template <int E>
__global__ void k_pc_rhs(int ncol, int ny, int nz, int m, const double* __restrict__ v, double* __restrict__ rhs) {
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * NK2D_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (task >= ncol) return;
    double vv[E];
    load_col<E>(v, task, lane, vv);
    double* dst = rhs + (size_t)task * m;  // task = tr*ny + j = tr*nb + j
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        if (k < nz) {
            dst[k] = vv[e];
            dst[nz + k] = 0.0;
            dst[2 * nz + k] = 0.0;
        }
    }
}

// out = -(u_3 + v), packed
template <int E>
__global__ void k_pc_result(int ncol, int ny, int nz, int m, const double* __restrict__ v, const double* __restrict__ x,
                            double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * NK2D_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (task >= ncol) return;
    double vv[E];
    load_col<E>(v, task, lane, vv);
    const double* src = x + (size_t)task * m + 2 * nz;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        vv[e] = (k < nz) ? -(src[k] + vv[e]) : 0.0;
    }
    store_col<E>(out, task, lane, vv);
}

// shifted systems: block j of the right-hand side holds every tracer of ypos column j
template <int E>
__global__ void k_pc_rhs_all(int ncol, int ny, int nz, int m, const double* __restrict__ v, double* __restrict__ rhs) {
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * NK2D_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (task >= ncol) return;
    const int tr = task / ny, j = task - tr * ny;
    double vv[E];
    load_col<E>(v, task, lane, vv);
    double* dst = rhs + (size_t)j * m + (size_t)tr * nz;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        if (k < nz) dst[k] = vv[e];
    }
}

template <int E>
__global__ void k_pc_result_all(int ncol, int ny, int nz, int m, const double* __restrict__ x, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * NK2D_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (task >= ncol) return;
    const int tr = task / ny, j = task - tr * ny;
    const double* src = x + (size_t)j * m + (size_t)tr * nz;
    double vv[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        vv[e] = (k < nz) ? src[k] : 0.0;
    }
    store_col<E>(out, task, lane, vv);
}

PcDev make_pcdev(const nk2d_ctx* c, const Precond* pc) {
    PcDev P;
    P.m = pc->m; P.nb = pc->nb; P.nz = c->nz; P.ny = c->ny; P.nt = pc->nt; P.tc = c->tc;
    P.mode = pc->mode; P.kind = c->kind; P.E = c->E; P.DZR = c->DZR;
    P.scale = pc->scale;
    for (int i = 0; i < NK2D_MAX_SHIFTS; ++i) P.sigma[i] = pc->sigma[i];
    P.ph_sig = c->d.phos_params[2]; P.ph_rd = c->d.phos_params[3]; P.ph_rp = c->d.phos_params[4];
    P.ph_vs = c->d.phos_params[5];
    P.PJ = pc->PJ; P.dt = pc->dt;
    for (int i = 0; i < NK2D_MAX_TRACERS; ++i) { P.surf[i] = c->d.surf_rate[i]; P.decay[i] = c->d.decay_rate[i]; }
    return P;
}

}  // namespace

void nk2d_precond_free(nk2d_ctx* c) {
    Precond* pc = (Precond*)c->precond;
    if (!pc) return;
    double* bufs[] = {pc->PJ, pc->SINV, pc->BUF, pc->YV, pc->XV, pc->ROWS, pc->PREV, pc->RV, pc->X0, pc->PINV};
    for (double* b : bufs)
        if (b) (void)hipFree(b);
    if (pc->SINV32) (void)hipFree(pc->SINV32);
    delete pc;
    c->precond = nullptr;
}

namespace {

// allocate for `nsys` systems of block size m = nslot * nz, load the Jacobian planes at the
// `nt` times and run the block elimination (Schur complements + their explicit inverses)
// ylin: linearisation state of every time level (state dependent modules), or null
int precond_build(nk2d_ctx* c, int mode, int nt, int nslot, int nsys, const double* times, const double* const* ylin) {
    Precond* pc = (Precond*)c->precond;
    const int m = nslot * c->nz;
    // gigabytes of Schur inverses: keep the allocation when the next factorisation fits in it (the
    // phosphorus preconditioner factorises three shifted systems per Newton iteration)
    // (option "pc_fp32": single precision storage of the explicit inverses, for the shifted systems of mode 1 too)
    const int want32 = c->pc_fp32 ? 1 : 0;
    const bool reuse = pc && pc->mode == mode && pc->nt == nt && pc->m == m && pc->nb == c->ny && pc->cap_sys >= nsys &&
                       pc->fp32 == want32;
    if (!reuse) {
        nk2d_precond_free(c);
        pc = new Precond();
        c->precond = pc;
        pc->PJ = pc->SINV = pc->BUF = pc->YV = pc->XV = pc->ROWS = pc->PREV = pc->RV = pc->X0 = pc->PINV = nullptr;
        pc->SINV32 = nullptr;
        pc->fp32 = want32;
        pc->cap_sys = std::max(nsys, mode == 1 ? 2 : nsys);
    }
    pc->mode = mode;
    pc->nsys = nsys;
    pc->nt = nt;
    pc->nz = c->nz;
    pc->tc = c->tc;
    pc->m = m;
    pc->nb = c->ny;
    pc->dt = (c->d.t1 - c->d.t0) / 3;
    pc->scale = 0.0;
    for (int i = 0; i < NK2D_MAX_SHIFTS; ++i) pc->sigma[i] = 0.0;
    const size_t P = (size_t)c->nz * c->ny, mm = (size_t)pc->m * pc->m;
    if (!reuse) {
        const size_t cap = (size_t)pc->cap_sys;
        NK2D_CHECK(c, hipMalloc((void**)&pc->PJ, sizeof(double) * pc->nt * PL_COUNT * P));
        if (pc->fp32) {
            NK2D_CHECK(c, hipMalloc((void**)&pc->SINV32, sizeof(float) * cap * pc->nb * mm));
            NK2D_CHECK(c, hipMalloc((void**)&pc->PREV, sizeof(double) * cap * mm));
            NK2D_CHECK(c, hipMalloc((void**)&pc->RV, sizeof(double) * cap * pc->nb * pc->m));
            NK2D_CHECK(c, hipMalloc((void**)&pc->X0, sizeof(double) * cap * pc->nb * pc->m));
        } else {
            NK2D_CHECK(c, hipMalloc((void**)&pc->SINV, sizeof(double) * cap * pc->nb * mm));
        }
        NK2D_CHECK(c, hipMalloc((void**)&pc->BUF, sizeof(double) * 2 * cap * mm));
        NK2D_CHECK(c, hipMalloc((void**)&pc->PINV, sizeof(double) * 2 * cap * PC_NB * PC_NB));
        NK2D_CHECK(c, hipMalloc((void**)&pc->ROWS, sizeof(double) * cap * PC_NB * pc->m));
        NK2D_CHECK(c, hipMalloc((void**)&pc->YV, sizeof(double) * cap * pc->nb * pc->m));
        NK2D_CHECK(c, hipMalloc((void**)&pc->XV, sizeof(double) * cap * pc->nb * pc->m));
    }
    for (int tau = 0; tau < pc->nt; ++tau) {
        double t = times[tau];
        double* out[1] = {c->KV[4]};
        NK2D_TRY(nk2d_k_vmix(c, 1, &t, out));
        NK2D_TRY(nk2d_k_jac(c, c->KV[4], ylin ? ylin[tau] : nullptr));
        const double* planes[PL_COUNT] = {c->JL, c->JS, c->JC, c->JN, c->JU, c->UPR};
        for (int pl = 0; pl < PL_COUNT; ++pl)
            NK2D_TRY(nk2d_k_unpack_plane(c, planes[pl], c->nz, c->ny, pc->PJ + ((size_t)tau * PL_COUNT + pl) * P));
    }
    return 0;
}

int precond_eliminate(nk2d_ctx* c) {
    Precond* pc = (Precond*)c->precond;
    const int nsys = pc->nsys;
    const size_t mm = (size_t)pc->m * pc->m;
    PcDev D = make_pcdev(c, pc);
    const int m = pc->m;
    const dim3 blk(256), grd((m + 255) / 256, m, nsys);
    for (int j = 0; j < pc->nb; ++j) {
        // the Schur inverse of the column before: inside SINV ([nsys][nb][m][m]), or -- single precision storage -- the
        // double precision copy kept of that one column ([nsys][m][m])
        const double* prev = (j > 0) ? (pc->fp32 ? pc->PREV : pc->SINV + (size_t)(j - 1) * mm) : nullptr;
        hipLaunchKernelGGL(k_pc_schur, grd, blk, 0, nk2d_s(c), D, j, prev, pc->fp32 ? mm : (size_t)pc->nb * mm, pc->BUF);
        int src = 0;
        for (int p0 = 0; p0 < m; p0 += PC_NB) {
            const int nbk = std::min(PC_NB, m - p0);
            const double* from = pc->BUF + (size_t)src * nsys * mm;
            double* to = pc->BUF + (size_t)(1 - src) * nsys * mm;
            // (measured, set-up of iage: 416 x 416 0.529 -> 0.496 s, 208 x 208 0.088 -> 0.096 s, 104 x 104 0.022 -> 0.024 s --
            // the launch is as long as its one workgroup that also inverts the next pivot block, 13 of ~ 22 us at every size;
            // value 1 therefore takes it from m = 1024, 2 everywhere: profiles/r04_pc_fused_panel_step.log)
            if (!c->pc_valu && (c->pc_fused >= 2 || (c->pc_fused == 1 && m >= 1024))) {
                // (the first pivot block's inverse by a launch of its own, the others by the step before)
                double* pin = pc->PINV + (size_t)((p0 / PC_NB) & 1) * nsys * PC_NB * PC_NB;
                double* pout = pc->PINV + (size_t)(1 - ((p0 / PC_NB) & 1)) * nsys * PC_NB * PC_NB;
                if (p0 == 0)
                    hipLaunchKernelGGL(k_pc_pivot_invert, dim3(1, 1, nsys), dim3(256), 0, nk2d_s(c), m, 0, nbk, from, pin);
                hipLaunchKernelGGL(k_pc_gj_step, dim3((m + 63) / 64, (m + 63) / 64, nsys), dim3(256), 0, nk2d_s(c), m, p0, nbk, from, to,
                                   (const double*)pin, pout);
                src = 1 - src;
                continue;
            }
            hipLaunchKernelGGL(k_pc_gj_rows, dim3((m + 255) / 256, nbk, nsys), dim3(256), 0, nk2d_s(c), m, p0, nbk, from,
                               pc->ROWS);
            if (c->pc_valu)
                hipLaunchKernelGGL(k_pc_gj_update, dim3((m + 63) / 64, (m + 63) / 64, nsys), dim3(256), 0, nk2d_s(c), m, p0,
                                   nbk, from, pc->ROWS, to);
            else
                hipLaunchKernelGGL(k_pc_gj_update_mfma, dim3((m + 63) / 64, (m + 63) / 64, nsys), dim3(256), 0, nk2d_s(c),
                                   m, p0, nbk, from, pc->ROWS, to);
            src = 1 - src;
        }
        for (int sys = 0; sys < nsys; ++sys) {
            const double* inv = pc->BUF + ((size_t)src * nsys + sys) * mm;
            if (pc->fp32) {
                hipLaunchKernelGGL(k_pc_to_f32, dim3((unsigned)((mm + 255) / 256)), dim3(256), 0, nk2d_s(c), inv,
                                   pc->SINV32 + ((size_t)sys * pc->nb + j) * mm, mm);
                NK2D_CHECK(c, hipMemcpyAsync(pc->PREV + (size_t)sys * mm, inv, sizeof(double) * mm, hipMemcpyDeviceToDevice, nk2d_s(c)));
            } else {
                NK2D_CHECK(c, hipMemcpyAsync(pc->SINV + ((size_t)sys * pc->nb + j) * mm, inv, sizeof(double) * mm,
                                             hipMemcpyDeviceToDevice, nk2d_s(c)));
            }
        }
        NK2D_CHECK(c, hipGetLastError());
        // bounded queue depth: a column is ~80 launches, and the whole elimination used to be queued (33 000
        // launches at 416 x 416) before the first synchronisation.  Under `rocprofv3 --pmc` that crashed the
        // profiler's dispatch interceptor (SIGSEGV in librocprofiler-sdk.so reached from this loop's
        // hipLaunchKernel, with or without torch in the process: gpurun_out/r02_pc_fetch.log resolved against
        // the probe's /proc/self/maps).  The GPU is never idle for it: a column is 1.3 ms of work.
        if ((j & 3) == 3) NK2D_CHECK(c, hipStreamSynchronize(nk2d_s(c)));
    }
    NK2D_CHECK(c, hipStreamSynchronize(nk2d_s(c)));
    return 0;
}

// block forward / backward substitution of systems [sys0, sys0 + nsys) with the right-hand
// sides already in XV; the solutions end up in XV
int precond_substitute(nk2d_ctx* c, int sys0, int nsys) {
    Precond* pc = (Precond*)c->precond;
    PcDev D = make_pcdev(c, pc);
    const int m = pc->m, nb = pc->nb;
    const size_t mm = (size_t)m * m;
    const size_t mstride = (size_t)nb * mm;      // system stride inside SINV
    const size_t vstride = (size_t)nb * m;       // system stride inside YV / XV
    const double* sinv = pc->fp32 ? nullptr : pc->SINV + (size_t)sys0 * mstride;
    const float* sinv32 = pc->fp32 ? pc->SINV32 + (size_t)sys0 * mstride : nullptr;
    double* yv = pc->YV + (size_t)sys0 * vstride;
    double* xv = pc->XV + (size_t)sys0 * vstride;
    const dim3 blk(256), grd((m + 3) / 4, nsys);
    // even m (16-byte aligned rows) and a vector that fits the LDS staging: the wide kernel
    const bool wide = !c->pc_valu && (m % 2 == 0) && m <= PC_GEMV_XMAX;
    // y_0 = r_0
    for (int sys = 0; sys < nsys; ++sys)
        NK2D_CHECK(c, hipMemcpyAsync(yv + (size_t)sys * vstride, xv + (size_t)sys * vstride, sizeof(double) * m,
                                     hipMemcpyDeviceToDevice, nk2d_s(c)));
    for (int j = 1; j < nb; ++j) {
        if (pc->fp32)
            hipLaunchKernelGGL(k_pc_gemv32, grd, blk, 0, nk2d_s(c), D, 0, j, sinv32 + (size_t)(j - 1) * mm, mstride,
                               yv + (size_t)(j - 1) * m, xv + (size_t)j * m, vstride, yv + (size_t)j * m,
                               (double*)nullptr);
        else if (wide)
            hipLaunchKernelGGL(k_pc_gemv2, grd, blk, 0, nk2d_s(c), D, 0, j, sinv + (size_t)(j - 1) * mm, mstride,
                               yv + (size_t)(j - 1) * m, xv + (size_t)j * m, vstride, yv + (size_t)j * m,
                               (double*)nullptr);
        else
            hipLaunchKernelGGL(k_pc_gemv, grd, blk, 0, nk2d_s(c), D, 0, j, sinv + (size_t)(j - 1) * mm, mstride,
                               yv + (size_t)(j - 1) * m, xv + (size_t)j * m, vstride, yv + (size_t)j * m,
                               (double*)nullptr);
    }
    // backward: x_j = Sinv_j (y_j - U_j x_{j+1}); each launch leaves y_{j-1} - U_{j-1} x_j behind for the next
    for (int j = nb - 1; j >= 0; --j) {
        if (pc->fp32)
            hipLaunchKernelGGL(k_pc_gemv32, grd, blk, 0, nk2d_s(c), D, 1, j, sinv32 + (size_t)j * mm, mstride,
                               yv + (size_t)j * m, (const double*)nullptr, vstride, xv + (size_t)j * m,
                               (j > 0) ? yv + (size_t)(j - 1) * m : (double*)nullptr);
        else if (wide)
            hipLaunchKernelGGL(k_pc_gemv2, grd, blk, 0, nk2d_s(c), D, 1, j, sinv + (size_t)j * mm, mstride,
                               yv + (size_t)j * m, (const double*)nullptr, vstride, xv + (size_t)j * m,
                               (j > 0) ? yv + (size_t)(j - 1) * m : (double*)nullptr);
        else
            hipLaunchKernelGGL(k_pc_gemv, grd, blk, 0, nk2d_s(c), D, 1, j, sinv + (size_t)j * mm, mstride,
                               yv + (size_t)j * m, (const double*)nullptr, vstride, xv + (size_t)j * m,
                               (j > 0) ? yv + (size_t)(j - 1) * m : (double*)nullptr);
    }
    NK2D_CHECK(c, hipGetLastError());
    return 0;
}

}  // namespace

static int precond_setup_levels(nk2d_ctx* c, const double* const* states);

extern "C" int nk2d_precond_setup(nk2d_ctx* c) {
    NK2D_CHECK(c, hipSetDevice(c->dev));
    if (c->kind == 1) return nk2d_fail(c, "nk2d_precond_setup: the phosphorus module uses nk2d_shift_factor / nk2d_shift_solve");
    if (c->kind == 2 && c->d.sms_nrec > 0 && c->d.sink_thres > 0.0)
        return nk2d_fail(c, "nk2d_precond_setup: a forced module with a sink threshold needs nk2d_precond_setup_states");
    return precond_setup_levels(c, nullptr);
}

// states[i]: the tracer at the end of the i-th third of the year (forced.apply_precond_jacobian,
// forced.py:222-236); they enter through the sink threshold only
extern "C" int nk2d_precond_setup_states(nk2d_ctx* c, const nk2d_vec* states) {
    NK2D_CHECK(c, hipSetDevice(c->dev));
    if (c->kind != 2) return nk2d_fail(c, "nk2d_precond_setup_states: forced modules with forcing files only");
    if (!states || !states[0] || !states[1] || !states[2]) return nk2d_fail(c, "nk2d_precond_setup_states: three states needed");
    const double* lv[3] = {(const double*)states[0], (const double*)states[1], (const double*)states[2]};
    return precond_setup_levels(c, lv);
}

static int precond_setup_levels(nk2d_ctx* c, const double* const* states) {
    // Jacobian planes at the three mid-interval times (iage.py:85-89)
    const double dt = (c->d.t1 - c->d.t0) / 3;
    const double times[3] = {c->d.t0 + 0.5 * dt, c->d.t0 + 1.5 * dt, c->d.t0 + 2.5 * dt};
    NK2D_TRY(precond_build(c, 0, 3, 3, c->tc, times, states));
    return precond_eliminate(c);
}

// A_i = scale * J(t, lin_state) - shifts[i] * I, all tracers of the module in one system
extern "C" int nk2d_shift_factor(nk2d_ctx* c, double t, double scale, int32_t nshift, const double* shifts) {
    NK2D_CHECK(c, hipSetDevice(c->dev));
    if (nshift < 1 || nshift > NK2D_MAX_SHIFTS) return nk2d_fail(c, "nk2d_shift_factor: 1 <= nshift <= NK2D_MAX_SHIFTS");
    const double* ylin = nullptr;
    if (c->kind == 1) {
        if (!c->ylin_set) return nk2d_fail(c, "nk2d_shift_factor: call nk2d_set_lin_state first");
        ylin = c->YLIN;
    }
    NK2D_TRY(precond_build(c, 1, 1, c->tc, nshift, &t, ylin ? &ylin : nullptr));
    Precond* pc = (Precond*)c->precond;
    pc->scale = scale;
    for (int i = 0; i < nshift; ++i) pc->sigma[i] = shifts[i];
    return precond_eliminate(c);
}

extern "C" int nk2d_shift_solve(nk2d_ctx* c, int32_t i, nk2d_vec v, nk2d_vec out) {
    NK2D_CHECK(c, hipSetDevice(c->dev));
    Precond* pc = (Precond*)c->precond;
    if (!pc || pc->mode != 1) return nk2d_fail(c, "nk2d_shift_solve: call nk2d_shift_factor first");
    if (i < 0 || i >= pc->nsys) return nk2d_fail(c, "nk2d_shift_solve: no such system");
    const size_t vstride = (size_t)pc->nb * pc->m;
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_pc_rhs_all<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                              c->ncol, c->ny, c->nz, pc->m, (const double*)v, pc->XV + (size_t)i * vstride));
    if (pc->fp32)      // (the right-hand sides are kept for the residual)
        NK2D_CHECK(c, hipMemcpyAsync(pc->RV + (size_t)i * vstride, pc->XV + (size_t)i * vstride, sizeof(double) * vstride,
                                     hipMemcpyDeviceToDevice, nk2d_s(c)));
    NK2D_TRY(precond_substitute(c, i, 1));
    if (pc->fp32) {
        // single precision inverses: "pc_refine" corrections x += S32^-1 (rhs - A x) against the exact shifted operator, as in
        // nk2d_precond_apply
        const size_t nvec = (size_t)pc->nb * pc->m;
        PcDev D = make_pcdev(c, pc);
        double* xv = pc->XV + (size_t)i * vstride;
        double* rv = pc->RV + (size_t)i * vstride;
        double* x0 = pc->X0 + (size_t)i * vstride;
        for (int it = 0; it < c->pc_refine; ++it) {
            NK2D_CHECK(c, hipMemcpyAsync(x0, xv, sizeof(double) * nvec, hipMemcpyDeviceToDevice, nk2d_s(c)));
            hipLaunchKernelGGL(k_pc_residual_shift, dim3((pc->m + 255) / 256, pc->nb, 1), dim3(256), 0, nk2d_s(c), D, (int)i, rv, x0, xv);
            NK2D_TRY(precond_substitute(c, i, 1));
            hipLaunchKernelGGL(k_pc_add, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, nk2d_s(c), xv, x0, nvec);
        }
    }
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_pc_result_all<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                              c->ncol, c->ny, c->nz, pc->m, pc->XV + (size_t)i * vstride, (double*)out));
    NK2D_CHECK(c, hipGetLastError());
    return 0;
}

extern "C" int nk2d_precond_apply(nk2d_ctx* c, nk2d_vec v, nk2d_vec out) {
    NK2D_CHECK(c, hipSetDevice(c->dev));
    Precond* pc = (Precond*)c->precond;
    if (!pc || pc->mode != 0) return nk2d_fail(c, "nk2d_precond_apply: call nk2d_precond_setup first");
    const int m = pc->m;
    // right-hand sides into XV (used as r_j), forward sweep writes YV
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_pc_rhs<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c), c->ncol,
                                              c->ny, c->nz, m, (const double*)v, pc->XV));
    if (pc->fp32) {
        // single precision inverses: x0 = S32^-1 rhs, then "pc_refine" corrections x += S32^-1 (rhs - A x) against the exact
        // operator -- each multiplies the error by that of the single precision inverses
        const size_t nvec = (size_t)pc->nsys * pc->nb * m;
        PcDev D = make_pcdev(c, pc);
        NK2D_CHECK(c, hipMemcpyAsync(pc->RV, pc->XV, sizeof(double) * nvec, hipMemcpyDeviceToDevice, nk2d_s(c)));
        NK2D_TRY(precond_substitute(c, 0, pc->nsys));
        for (int it = 0; it < c->pc_refine; ++it) {
            NK2D_CHECK(c, hipMemcpyAsync(pc->X0, pc->XV, sizeof(double) * nvec, hipMemcpyDeviceToDevice, nk2d_s(c)));
            hipLaunchKernelGGL(k_pc_residual, dim3((m + 255) / 256, pc->nb, pc->nsys), dim3(256), 0, nk2d_s(c), D, pc->RV,
                               pc->X0, pc->XV);
            NK2D_TRY(precond_substitute(c, 0, pc->nsys));
            hipLaunchKernelGGL(k_pc_add, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, nk2d_s(c), pc->XV, pc->X0, nvec);
        }
    } else {
        NK2D_TRY(precond_substitute(c, 0, pc->nsys));
    }
    NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_pc_result<EE>, dim3(nk2d_grid(c->ncol)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                              c->ncol, c->ny, c->nz, m, (const double*)v, pc->XV, (double*)out));
    NK2D_CHECK(c, hipGetLastError());
    return 0;
}
