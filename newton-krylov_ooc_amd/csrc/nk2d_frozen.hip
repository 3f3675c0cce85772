// nk2d_frozen.hip -- the frozen year in ONE launch on a schedule cache (k_frozen_persistent).
#include "nk2d_bodies.h"

#include <algorithm>
#include <vector>

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

// TEAM = 1: a workgroup is ONE column, its four waves the team of newton_team_body (a stage tendency each on three of them,
// the complex system on the fourth, exchanges through LDS): the phase of a small grid is the dependent arithmetic of one
// column's Newton iteration, and the team cuts that chain (three tendencies one after the other, then the real and the
// complex solve one after the other -> one tendency, then both solves side by side).  Same arithmetic, same bits.
template <int E, int KIND, int TEAM = 0>
__global__ void __launch_bounds__(NK2D_BLOCK) k_frozen_persistent(DevP P, FrozenArgs A) {
    constexpr int XCD = 0, NB = 1;      // (rounds 2 - 3 also had all workgroups on one XCD and a grid barrier between the phases)
    __shared__ int lds_ok;
    __shared__ double team_lds[TEAM ? sizeof(TeamLds<E, 3>) / sizeof(double) : 1];
    constexpr int MPX = XCD ? 2 : 1;
    const int lane = threadIdx.x & 63;
    int wg = (int)blockIdx.x;
    const int tw = uni_i((int)(threadIdx.x >> 6));                      // TEAM: the wave's place in its team
    // the column of this wave.  A team: the workgroup's.  A wave per column: adjacent columns of one tracer to a workgroup, or
    // -- `by_column` -- the workgroup is ONE ypos column and its waves that column's tracers (what is the same for every tracer
    // of a ypos column is then shared through LDS)
    const bool by_col = !TEAM && !XCD && NB != 0 && A.by_column != 0;
    const int wave = TEAM ? uni_i(wg) : (by_col ? uni_i(tw * P.ny + wg) : uni_i(wg * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6)));
    const bool col_wave = wave < P.ncol && (!by_col || (tw < P.tc && wg < P.ny));
    // NB: neighbour-to-neighbour hand-over instead of the grid barrier.  The unit is the workgroup: one column (teams), or
    // the columns of its waves -- then the workgroup to the left matters if its first column has a left neighbour, the one to
    // the right if its last column has a right neighbour (a tracer boundary inside the workgroup needs nothing)
    int nb_left, nb_right;
    if constexpr (TEAM) {
        const int nb_j = wave % P.ny;
        nb_left = (nb_j > 0) ? wg - 1 : -1;
        nb_right = (nb_j < P.ny - 1) ? wg + 1 : -1;
    } else if (by_col) {
        nb_left = (wg > 0) ? wg - 1 : -1;
        nb_right = (wg < P.ny - 1) ? wg + 1 : -1;
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
    if (!nbs.sync()) { status = 1; goto finish; }
    // a wave per column, three and more levels per lane (option "frozen_coef_lds"): the static coefficients of the wave's column
    // in LDS for the whole year (dynamic shared memory of the launch: NK2D_COEF_LDS_DOUBLES(E) doubles per wave)
    constexpr bool COEF_LDS = NB != 0 && !TEAM && !XCD && E >= 3;
    // Layout of the dynamic shared memory.  Adjacent columns: per wave [coefficients][W].  By column: [coefficients of the ypos
    // column][step block: 3 mixing columns, JL, JU][per wave: W][per wave: pivots of the real system] (each part present where
    // its bit of A.coef_lds is set; frozen_lds_doubles() on the host computes the same)
    extern __shared__ double dyn_lds[];
    const bool w_in_lds = COEF_LDS && (A.coef_lds & 2) != 0;
    const bool step_in_lds = COEF_LDS && by_col && (A.coef_lds & 4) != 0;
    const bool piv_in_lds = COEF_LDS && by_col && (A.coef_lds & 8) != 0;
    const int nwv = (int)(blockDim.x >> 6);
    double* my_coef;
    double* my_w;
    double* step_lds = nullptr;
    double* my_piv = nullptr;
    if (by_col) {
        double* p = dyn_lds;
        my_coef = p; p += NK2D_COEF_LDS_DOUBLES(E);
        step_lds = p; p += step_in_lds ? 5 * E * 64 : 0;
        my_w = p + (size_t)tw * (3 * E * 64); p += w_in_lds ? (size_t)nwv * 3 * E * 64 : 0;
        my_piv = p + (size_t)tw * (E * 64);
    } else {
        my_coef = dyn_lds + (size_t)(threadIdx.x >> 6) * (NK2D_COEF_LDS_DOUBLES(E) + (w_in_lds ? 3 * E * 64 : 0));
        my_w = my_coef + NK2D_COEF_LDS_DOUBLES(E);
    }
    const LdsSrc L = {my_coef, my_w, step_lds, my_piv};
    (void)L;
    if constexpr (COEF_LDS) {
        if (A.coef_lds && col_wave && (!by_col || tw == 0)) {
            ColCoef<E> cf;
            load_coef<E>(P, wave % P.ny, lane, cf);
            store_coef_lds<E>(my_coef, lane, cf);
        }
        if (by_col && A.coef_lds) __syncthreads();      // (the other tracers' waves read what wave 0 stored)
    }
    int lds_step = -1;      // the step whose constants the step block / the pivots hold
    (void)lds_step;
    // first attempt of the year: Z0 = 0, W0 = 0 (radau.py:445-446)
    if (col_wave && (!TEAM || tw == 0)) {
        double zero[E];
#pragma unroll
        for (int e = 0; e < E; ++e) zero[e] = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            store_col<E, MPX>(FZ_Z + i * nv, wave, lane, zero);
            store_col<E, (TEAM ? MPX : 0)>(A.W + i * nv, wave, lane, zero);
            if constexpr (COEF_LDS) {
                if (w_in_lds) w_lds_put<E>(my_w, i, lane, zero);
            }
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
                if constexpr (COEF_LDS) {
                    if (step_in_lds && lds_step != i) {
                        // first phase of a step: what is constant over the step's iterations goes to LDS once -- the three mixing
                        // columns and JL, JU of the ypos column shared out over the workgroup's waves, each wave's own pivots
                        for (int r = tw; r < 5; r += nwv) {
                            double v[E];
                            // (selects, not an index: a struct indexed at run time would live in scratch memory)
                            const double* src = (r == 0) ? FA.st.kv[0] : ((r == 1) ? FA.st.kv[1] : ((r == 2) ? FA.st.kv[2]
                                                : ((r == 3) ? FA.sw.JL : FA.sw.JU)));
                            load_col<E, MPX>(src, wave % P.ny, lane, v);
                            w_lds_put<E>(step_lds, r, lane, v);
                        }
                        if (piv_in_lds && col_wave) {
                            double v[E];
                            load_col<E>(FA.sw.fr_inv, wave, lane, v);
                            w_lds_put<E>(my_piv, 0, lane, v);
                        }
                        lds_step = i;
                        __syncthreads();
                    }
                }
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
                        if constexpr (COEF_LDS) {
                            if (step_in_lds && piv_in_lds) { newton_fused_body<E, KIND, 0, 1, MPX, 1, 15>(P, FA, wave, lane, &Fin, &L); taken = true; }
                            else if (step_in_lds) { newton_fused_body<E, KIND, 0, 1, MPX, 1, 7>(P, FA, wave, lane, &Fin, &L); taken = true; }
                            else if (w_in_lds) { newton_fused_body<E, KIND, 0, 1, MPX, 1, 3>(P, FA, wave, lane, &Fin, &L); taken = true; }
                            else if (A.coef_lds) { newton_fused_body<E, KIND, 0, 1, MPX, 1, 1>(P, FA, wave, lane, &Fin, &L); taken = true; }
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
                        if constexpr (COEF_LDS) {
                            if (step_in_lds && piv_in_lds) { newton_fused_body<E, KIND, 0, 1, MPX, 0, 15>(P, FA, wave, lane, nullptr, &L); taken = true; }
                            else if (step_in_lds) { newton_fused_body<E, KIND, 0, 1, MPX, 0, 7>(P, FA, wave, lane, nullptr, &L); taken = true; }
                            else if (w_in_lds) { newton_fused_body<E, KIND, 0, 1, MPX, 0, 3>(P, FA, wave, lane, nullptr, &L); taken = true; }
                            else if (A.coef_lds) { newton_fused_body<E, KIND, 0, 1, MPX, 0, 1>(P, FA, wave, lane, nullptr, &L); taken = true; }
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
                if constexpr (COEF_LDS) {
                    if (w_in_lds) {     // (this rare phase writes W to memory: into the column's LDS copy from there)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            double wv[E];
                            load_col<E, (TEAM ? MPX : 0)>(A.W + r * nv, wave, lane, wv);
                            w_lds_put<E>(my_w, r, lane, wv);
                        }
                    }
                }
            }
            swapY ^= 1;
            swapZ ^= 1;
            FZ_SYNC()
        }
        done = i + 1;
    }
finish:
    if constexpr (COEF_LDS) {
        if (w_in_lds && col_wave) {      // W of the last phase back where the launch-per-phase path keeps it
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                double wv[E];
                w_lds_get<E>(my_w, r, lane, wv);
                store_col<E>(A.W + r * nv, wave, lane, wv);
            }
        }
    }
    // no barrier behind the last phase: every workgroup reports a failure of its own (the host cleared `out`), the first the
    // rest -- a workgroup that gave up raised the abort flag, its neighbours give up on it in turn
    if (status != 0 && threadIdx.x == 0) A.out[0] = (double)status;
    if (wg == 0 && threadIdx.x == 0) {
        A.out[1] = (double)done; A.out[2] = (double)swapY; A.out[3] = (double)swapZ; A.out[4] = (double)nbs.phase;
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

// bytes of HBM the schedule cache of this context holds right now
int64_t nk2d_frozen_cache_bytes(const nk2d_ctx* c) {
    const nk2d_frozen_cache* fc = (const nk2d_frozen_cache*)c->frozen_cache;
    if (!fc || !fc->slab) return 0;
    const size_t ntab = (size_t)c->ncol * NK2D_TAB * 64;
    return (int64_t)(8 * fc->cap_rows * (3 * c->kv_len + 5 * c->np + 3 * c->nv + 3 * ntab));
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
// `coop`: every workgroup of the grid must be resident at once (the grid-barrier and hand-over flavours: a workgroup waits
// for others).  Launched PLAINLY all the same: the launch is refused here (hipErrorCooperativeLaunchTooLarge) when the chip
// cannot hold the grid -- occupancy of the kernel times the compute units --, the process runs one resident kernel at a
// time (nk2d_turn_take), and every wait inside the kernel is bounded by time, so a grid that did not become fully
// resident after all (a co-tenant on the chip) hands the year back instead of hanging.  hipLaunchCooperativeKernel did the
// same check and nothing else for this kernel -- through a queue of its own that the HIP runtime crashed on when the process
// ended under rocprofv3 (rounds 2 - 3: AqlQueue::~AqlQueue under hsa_shut_down).
// doubles of dynamic shared memory of a wave-per-column workgroup of `nwaves` waves (the kernel's layout; bits: LdsSrc)
static size_t frozen_lds_doubles(int E, int bits, bool by_column, int nwaves) {
    if (!bits) return 0;
    if (!by_column) return (size_t)nwaves * (NK2D_COEF_LDS_DOUBLES(E) + ((bits & 2) ? 3 * E * 64 : 0));
    return NK2D_COEF_LDS_DOUBLES(E) + ((bits & 4) ? 5 * E * 64 : 0) + (size_t)nwaves * (((bits & 2) ? 3 * E * 64 : 0) + ((bits & 8) ? E * 64 : 0));
}

template <class K>
static hipError_t launch_resident(nk2d_ctx* c, K kernel, dim3 grid, dim3 block, DevP& P, FrozenArgs& A, size_t lds_bytes = 0) {
    int per_cu = 0;
    hipError_t rc = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)block.x, lds_bytes);
    if (rc != hipSuccess) return rc;
    hipDeviceProp_t prop;
    rc = hipGetDeviceProperties(&prop, c->dev);
    if (rc != hipSuccess) return rc;
    if ((long long)per_cu * prop.multiProcessorCount < (long long)grid.x) return hipErrorCooperativeLaunchTooLarge;
    if (std::getenv("NK2D_DIAG_COOPERATIVE")) {
        // diagnostic only (profiles/r04_slowdown_ab.log): the launch as rounds 2 - 3 made it, through the runtime's cooperative queue
        void* args[2] = {&P, &A};
        return hipLaunchCooperativeKernel((const void*)kernel, grid, block, args, (unsigned)lds_bytes, nk2d_s(c));
    }
    hipLaunchKernelGGL(kernel, grid, block, lds_bytes, nk2d_s(c), P, A);
    return hipGetLastError();
}

template <int E, int KIND, int TEAM>
static hipError_t launch_frozen_one(nk2d_ctx* c, dim3 grid, DevP& P, FrozenArgs& A) {
    // a team per column: the workgroup IS the column.  A wave per column: option "frozen_wpb" adjacent columns of one tracer to
    // a workgroup (its waves move in lock step, its neighbours are the workgroups to the left and right), or -- by_column --
    // one ypos column with all its tracers.  Three and more levels per lane: a wave's own data of the year in LDS
    const int wpb = TEAM ? NK2D_WAVES_PER_BLOCK : std::max(1, std::min(NK2D_WAVES_PER_BLOCK, c->frozen_wpb));
    const dim3 g = TEAM ? grid : dim3((unsigned)((c->ncol + wpb - 1) / wpb));
    if (!TEAM && E >= 3 && A.by_column)
        return launch_resident(c, k_frozen_persistent<E, KIND, TEAM>, dim3((unsigned)c->ny), dim3(64 * c->tc), P, A,
                               sizeof(double) * frozen_lds_doubles(E, A.coef_lds, true, c->tc));
    const size_t lds = (!TEAM && E >= 3 && A.coef_lds) ? sizeof(double) * frozen_lds_doubles(E, A.coef_lds, false, wpb) : 0;
    return launch_resident(c, k_frozen_persistent<E, KIND, TEAM>, g, dim3(64 * wpb), P, A, lds);
}
template <int KIND, int TEAM>
static hipError_t launch_frozen_e(nk2d_ctx* c, dim3 grid, DevP& P, FrozenArgs& A) {
    switch (c->E) {
        case 1: return launch_frozen_one<1, KIND, TEAM>(c, grid, P, A);
        case 2: return launch_frozen_one<2, KIND, TEAM>(c, grid, P, A);
        // three and four levels per lane: a wave per column (all module kinds)
        case 3: if constexpr (!TEAM) return launch_frozen_one<3, KIND, 0>(c, grid, P, A); else break;
        case 4: if constexpr (!TEAM) return launch_frozen_one<4, KIND, 0>(c, grid, P, A); else break;
        // five to eight levels per lane (up to 512 levels): a wave per column, linear sources
        case 5: if constexpr (!TEAM && KIND == 0) return launch_frozen_one<5, 0, 0>(c, grid, P, A); else break;
        case 6: if constexpr (!TEAM && KIND == 0) return launch_frozen_one<6, 0, 0>(c, grid, P, A); else break;
        case 7: if constexpr (!TEAM && KIND == 0) return launch_frozen_one<7, 0, 0>(c, grid, P, A); else break;
        case 8: if constexpr (!TEAM && KIND == 0) return launch_frozen_one<8, 0, 0>(c, grid, P, A); else break;
        default: break;
    }
    return hipErrorInvalidValue;
}
static hipError_t launch_frozen(nk2d_ctx* c, bool team, dim3 grid, DevP& P, FrozenArgs& A) {
    const bool forced = c->kind == 2;
    if (team) return forced ? launch_frozen_e<2, 1>(c, grid, P, A) : launch_frozen_e<0, 1>(c, grid, P, A);
    return forced ? launch_frozen_e<2, 0>(c, grid, P, A) : launch_frozen_e<0, 0>(c, grid, P, A);
}

// the slab of a schedule cache with room for `cap` rows from a thread of the library's own (hipMalloc of 120 GB: 0.03 - 3 s);
// what the cache held before is given back by that thread first.  alloc_state 1 while it runs, 2 / 3 when done / refused
static void frozen_alloc_in_thread(nk2d_frozen_cache* fc, int dev, size_t slab_bytes, size_t cap, double* old_slab, CacheRow* old_rows,
                                   FrozenRow* old_frows) {
    if (fc->alloc_thread.joinable()) fc->alloc_thread.join();
    fc->alloc_state.store(1);
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
}

// (Asking for that slab EARLIER -- at the end of the free-running year whose steps the products will repeat, so that it is
// there when the first product comes -- was built and measured in round 4 and taken out again: in a fresh process the
// hipMalloc of 120 GB holds the driver for ~ 4 s, and the preconditioner's set-up, which lies between F(x) and the first
// product and allocates its 10 GB of inverses, waited behind it: set-up 0.50 -> 4.18 s, spin-up of iage 416 x 416 3.98 / 5.14 ->
// 9.34 s (profiles/r04_prealloc_experiment.log).  Beside the first products, which allocate nothing, the same request costs
// them 0.1 - 1.2 s in all.)
int nk2d_frozen_persistent(nk2d_ctx* c, const double* sched, int64_t n, std::vector<char>* err_rows) {
    const bool needs_state = c->kind == 1 || (c->kind == 2 && c->d.sms_nrec > 0 && c->d.sink_thres > 0.0);
    if (!c->frozen_persistent || needs_state || c->hist_n != 0 || c->norm_hook || n < 1) return 1;
    // the cache holds the factorisation in double precision only: with option "factor_fp32" the recorded year read the single
    // precision copies, and the one-launch year would not be the same discrete map (round-3 ADVICE) -- launch by launch then
    if (c->factor_fp32) return 1;
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
        // option "frozen_cache_after": that many years of a schedule run launch by launch before its cache is built.  Default
        // 0; -1: 0 for caches below 8 GB, 3 above.  Building a 100 GB cache takes 26 ms where a one-launch year saves 40
        // (tools/probe_cache_build.py) -- but its FIRST allocation has been seen to take 0.8 s inside a Newton run, and a Newton
        // iteration with two or three Krylov iterations has nothing to pay that back with; a long Krylov solve has.
        const int after = c->frozen_cache_after >= 0 ? c->frozen_cache_after : (bytes > 8.0e9 ? 3 : 0);
        if (c->frozen_seen_key != key) { c->frozen_seen_key = key; c->frozen_seen_years = 0; }
        if (c->frozen_seen_years++ < after) return 1;
    }
    DevP P = make_devp(c);
    if (fc->key != key || fc->n != n) {
        // ---- (re)build the cache for this schedule
        if (fc->cap_rows < (size_t)n) {
            // ONE allocation for the whole cache, with room for the longer schedules of later Newton iterates: giving 100 GB
            // back and asking for them again costs seconds (measured inside a Newton run: 4.4 s), the first request 0.03 - 0.8 s
            NK2D_CHECK(c, hipStreamSynchronize(nk2d_s(c)));
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
                frozen_alloc_in_thread(fc, c->dev, slab_bytes, cap, old_slab, old_rows, old_frows);
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
        NK2D_CHECK(c, hipMemcpyAsync(fc->rows_dev, rows.data(), sizeof(CacheRow) * (size_t)n, hipMemcpyHostToDevice, nk2d_s(c)));
        NK2D_CHECK(c, hipMemcpyAsync(fc->frows_dev, fc->frows.data(), sizeof(FrozenRow) * (size_t)n, hipMemcpyHostToDevice, nk2d_s(c)));
        NK2D_CHECK(c, hipStreamSynchronize(nk2d_s(c)));    // `rows` leaves scope
        {
            const long long tasks = 4LL * c->ny * n;
            NK2D_DISPATCH_E(c->E, hipLaunchKernelGGL(k_cache_planes<EE>, dim3((unsigned)((tasks + 3) / 4)), dim3(NK2D_BLOCK), 0, nk2d_s(c),
                                                      P, fc->rows_dev, fc->C, (int)n));
            NK2D_CHECK(c, hipGetLastError());
            const long long ftasks = 2LL * c->ncol * n;
            NK2D_DISPATCH_EK(c->E, c->kind, hipLaunchKernelGGL((k_cache_factor<EE, KK>), dim3((unsigned)((ftasks + 3) / 4)), dim3(NK2D_BLOCK), 0,
                                                               nk2d_s(c), P, fc->rows_dev, fc->C, (int)n));
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
    NK2D_CHECK(c, hipMemsetAsync(c->YR_SYNC, 0, yr_sync_bytes(c), nk2d_s(c)));
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
    // what lives in LDS (option "frozen_coef_lds", bits of LdsSrc) and which columns share a workgroup (option "frozen_by_column"):
    // as much as lets a compute unit hold what the grid needs of it -- ceil(waves / compute units) waves: pivots, then the step
    // block, then W are given up until it fits the 160 KB (E = 8 keeps everything but the pivots)
    A.coef_lds = (c->E >= 3) ? c->frozen_coef_lds : 0;
    // (by column -- and with it the step block and the pivots -- from five levels per lane: 416^2 125.2 -> 120.3 ms; at four the shared
    // block does not pay for the wider hand-over, 208^2 65.2 -> 68.1 ms: profiles/r04_frozen_lds_by_column.log; option value 2 forces it)
    A.by_column = (c->frozen_by_column && c->tc <= NK2D_WAVES_PER_BLOCK && A.coef_lds &&
                   (c->E >= 5 || (c->frozen_by_column >= 2 && c->E >= 3))) ? 1 : 0;
    if (!A.by_column) A.coef_lds &= 3;
    {
        hipDeviceProp_t prop;
        NK2D_CHECK(c, hipGetDeviceProperties(&prop, c->dev));
        const int nw = A.by_column ? c->tc : std::max(1, std::min(NK2D_WAVES_PER_BLOCK, c->frozen_wpb));
        const int wgs = A.by_column ? c->ny : (c->ncol + nw - 1) / nw;
        const int per_cu = (wgs + prop.multiProcessorCount - 1) / prop.multiProcessorCount;
        for (const int drop : {8, 4, 2}) {
            if (8 * frozen_lds_doubles(c->E, A.coef_lds, A.by_column != 0, nw) * (size_t)per_cu <= 160u * 1024u) break;
            A.coef_lds &= ~drop;
        }
    }
    // option "frozen_team": a workgroup per column (four waves: newton_team_body) instead of a wave per column, up to two levels per
    // lane.  Measured in round 3 (profiles/r03_frozen_team.log, r03_frozen_nbsync.log): teams want a CU each and beat the
    // wave-per-column year at every such size -- 26^2 9.4 ms, 52^2 14.9, 104^2 27.0 with the neighbour hand-over.
    const bool team = c->frozen_team && c->E <= 2;
    const int nblk = team ? c->ncol : nk2d_grid(c->ncol);
    const double* o = c->hYR_OUT;
    {
        A.spin_ticks = (long long)(c->barrier_timeout_ms * 1.0e5);
        hipError_t rc = hipErrorInvalidValue;
        NK2D_CHECK(c, hipMemsetAsync(c->YR_OUT, 0, sizeof(double) * 32, nk2d_s(c)));
        NK2D_CHECK(c, hipEventRecord(c->yr_ev[0], nk2d_s(c)));
        // one resident kernel at a time in this process (the turn is held until this one has ended)
        struct Turn {
            int waves;
            explicit Turn(int w) : waves(w) { nk2d_turn_take(waves); }
            ~Turn() { nk2d_turn_give(waves); }
        } turn(team ? 4 * c->ncol : c->ncol);
        rc = launch_frozen(c, team, dim3(nblk), P, A);
        if (rc == hipErrorCooperativeLaunchTooLarge) { (void)hipGetLastError(); return 1; }
        NK2D_CHECK(c, rc);
        NK2D_CHECK(c, hipEventRecord(c->yr_ev[1], nk2d_s(c)));
        NK2D_CHECK(c, hipMemcpyAsync(c->hYR_OUT, c->YR_OUT, sizeof(double) * 32, hipMemcpyDeviceToHost, nk2d_s(c)));
        NK2D_CHECK(c, hipStreamSynchronize(nk2d_s(c)));
        if ((int)o[0] != 0 || (int64_t)o[1] != n) return 2;
    }
    if (team) c->frozen_team_years++;
    {   // the launch itself, between two events on the context's stream (bench.py's roofline of the one-launch year)
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

