// nk2d_radau.hip -- host controller of the device-resident Radau IIA(5) year.
//
// Replaces the solve_ivp("Radau") call of the reference's comp_fcn
// (nk_ooc/py_driver_2d/model_state.py:95-121).  The controller restates SciPy's
// published algorithm decision for decision (scipy/integrate/_ivp/radau.py:48-176,
// 295-572, common.py:68-134, base.py:181-208, ivp.py:707-723): initial-step
// heuristic, simplified Newton on the transformed collocation system, error
// estimate + filter, predictive step controller, Jacobian / factorisation reuse,
// cubic dense output, final value from the interpolant.  All vectors stay in HBM;
// the host only sees the scalar norms the decisions need.
//
// The sparse LU factorisations of SciPy are replaced by line relaxation: the
// vertical (stiff) direction of  (mu/h) I - J  is solved exactly per column by a
// wave-level tridiagonal solve, the weak horizontal coupling is swept to the
// a-priori contraction bound (nk2d_sweeps_for).  "LU" below therefore only records
// (h_lu, t_jac) -- there is nothing to factor.
//
// Step-replay mode: given a recorded accepted-step schedule the controller skips
// every decision (and the error estimate) and reproduces the smooth map
// schedule -> y(T); see tests/test_gpu_comp_fcn.py.
#include "nk2d_common.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <limits>
#include <string>
#include <vector>

namespace {

const double RC[3] = {0.15505102572168222, 0.6449489742783178, 1.0};
const double MU_REAL = 3.637834252744496;
const double MU_CR = 2.6810828736277523, MU_CI = -3.050430199247411;
const int NEWTON_MAXITER = 6;
const double MIN_FACTOR = 0.2, MAX_FACTOR = 10.0;

struct Ctl {
    nk2d_ctx* c;
    double t, t1;
    double h_abs, h_abs_old, err_old;
    bool has_old_h, has_old_err;
    bool current_jac, have_lu, have_dense;
    double h_lu, t_jac;
    double dense_t_old, dense_h;
    double newton_tol, max_step;
    int m_real, m_cplx;  // sweeps per solve for the current h_lu
    double n_total;      // number of unknowns (tc*nz*ny)
    // the set-up of the next step's first attempt (stage planes, predicted stage values) and, where due, the Jacobian
    // at its start were already queued with the commit of the step before (nk2d_r_step_boundary): for this (t, h)
    bool pre_setup;
    bool pre_jac;        // ... and so was the Jacobian of that attempt (option "jac_stage")
    double pre_t, pre_h;
    double fingerprint;  // nk2d_fingerprint of the context at the start of the year (recorded with every step)
    bool no_persistent = false;   // the one-launch frozen year was tried and given up for this year
    int n_iter_prev = 0;          // Newton iterations of the last solve that converged (0: none yet)
};

double rms_from_sum(double s, double count) { return std::sqrt(s) / std::sqrt(count); }

// a sum of squares over this context's tracers -> over the whole tracer module (nk2d_set_norm_hook:
// the caller's all-reduce when the module's tracers are sharded over contexts; identity otherwise)
inline void couple(const nk2d_ctx* c, double* sum) {
    if (c->norm_hook) *sum = c->norm_hook(c->norm_hook_user, *sum);
}

int eval_kv(nk2d_ctx* c, double t, int slot) {
    double* out[1] = {c->KV[slot]};
    return nk2d_k_vmix(c, 1, &t, out);
}

// J <- jac(t) : vertical mixing plane at t, then the five Jacobian planes
int refresh_jac(Ctl& s, double t, bool kv_in_slot3) {
    nk2d_ctx* c = s.c;
    if (!kv_in_slot3) NK2D_TRY(eval_kv(c, t, 4));
    NK2D_TRY(nk2d_k_jac(c, kv_in_slot3 ? c->KV[3] : c->KV[4], c->Y));  // c->Y is the state at t at every call site
    s.t_jac = t;
    return 0;
}

// SciPy's two factorisations of (mu/h) I - J: here the per-column pivot reciprocals and
// PCR tables of the line solves (k_factor) and the sweep counts from the contraction bound
int set_lu(Ctl& s, double h) {
    s.h_lu = h;
    s.have_lu = true;
    s.m_real = nk2d_sweeps_for(s.c, MU_REAL / h);
    s.m_cplx = nk2d_sweeps_for(s.c, MU_CR / h);
    if ((!s.c->single_swap || s.c->min_sweeps > 1) && nk2d_has_lateral(s.c)) {
        // device-side decisions: a launch may return at entry, the host cannot swap stage buffers behind it, and an
        // in-place single launch would let the update of one column race with the stage reads of its neighbours
        s.m_real = std::max(s.m_real, 2);
        s.m_cplx = std::max(s.m_cplx, 2);
    }
    s.c->st.nlu += 2;
    // the factorisation itself happens inside the first fused Newton launch that needs it
    s.c->lu_cre = MU_REAL / h; s.c->lu_ccr = MU_CR / h; s.c->lu_cci = MU_CI / h;
    s.c->factor_pending = 1;
    return 0;
}

// x = ((mu/h_lu) I - J)^-1 b for the real and/or the complex system; result in
// ping-pong buffer *buf
int solve_systems(Ctl& s, bool do_real, bool do_cplx, int* buf) {
    nk2d_ctx* c = s.c;
    const double cre = MU_REAL / s.h_lu, ccr = MU_CR / s.h_lu, cci = MU_CI / s.h_lu;
    const int m = std::max(do_real ? s.m_real : 0, do_cplx ? s.m_cplx : 0);
    int src = 0;
    for (int it = 0; it < m; ++it) {
        const bool r = do_real && it < s.m_real, q = do_cplx && it < s.m_cplx;
        // a system that has finished keeps its result in whichever buffer it last
        // wrote; keep both systems sweeping to the common count so that the result
        // buffer index is shared (extra sweeps only tighten the solution)
        (void)r; (void)q;
        NK2D_TRY(nk2d_k_sweep(c, do_real, do_cplx, it == 0, cre, ccr, cci, c->BR, c->BCR, c->BCI, src));
        src = 1 - src;
    }
    *buf = src;
    if (do_real) c->st.nsolve++;
    if (do_cplx) c->st.nsolve++;
    return 0;
}

int predict(Ctl& s, double t, double h) {
    nk2d_ctx* c = s.c;
    if (!s.have_dense) {
        NK2D_CHECK(c, hipMemsetAsync(c->Z, 0, sizeof(double) * 3 * c->nv, nk2d_s(c)));
        NK2D_CHECK(c, hipMemsetAsync(c->W, 0, sizeof(double) * 3 * c->nv, nk2d_s(c)));
        return 0;
    }
    double x[3];
    for (int i = 0; i < 3; ++i) x[i] = ((t + h * RC[i]) - s.dense_t_old) / s.dense_h;
    return nk2d_r_predict(c, x[0], x[1], x[2]);
}

// stage planes + predicted stage values of a step attempt in one launch (needs the dense output
// of a previous step; the very first step falls back to the two separate calls)
int setup_attempt(Ctl& s, double t, double h, int jac_stage = -1) {
    nk2d_ctx* c = s.c;
    double times[3], x[3];
    for (int i = 0; i < 3; ++i) {
        times[i] = t + (h * RC[i]);
        x[i] = ((t + h * RC[i]) - s.dense_t_old) / s.dense_h;
    }
    double* out[3] = {c->KV[0], c->KV[1], c->KV[2]};
    return nk2d_r_attempt_setup(c, times, out, x[0], x[1], x[2], jac_stage);
}

int stage_planes(Ctl& s, double t, double h) {
    double times[3];
    for (int i = 0; i < 3; ++i) times[i] = t + (h * RC[i]);
    double* out[3] = {s.c->KV[0], s.c->KV[1], s.c->KV[2]};
    return nk2d_k_vmix(s.c, 3, times, out);
}

// one simplified-Newton iteration: stage residuals, m line sweeps of both systems, update
// (fused: m launches)
int newton_iteration(Ctl& s, double mreal, double mcr, double mci) {
    nk2d_ctx* c = s.c;
    const int m = std::max(s.m_real, s.m_cplx);
    int src = 0;
    for (int it = 0; it < m; ++it) {
        NK2D_TRY(nk2d_r_newton_fused(c, it == 0, it == 0, it == m - 1, mreal, mcr, mci, src, m == 2));
        src = 1 - src;
    }
    c->st.nsolve += 2;
    return 0;
}

// Replay: exactly n_iters simplified-Newton iterations, no tests, nothing read back.
// Iterations k0 .. k1-1 of the n_total a step has.  row_last / row_prev (a frozen year): where the norm partials of the
// step's last and last-but-one iteration go, for the check after the year (run_replay); the others go to PART.
int newton_fixed(Ctl& s, double h, int k0, int k1, int n_total, double* row_last = nullptr, double* row_prev = nullptr) {
    nk2d_ctx* c = s.c;
    const double mreal = MU_REAL / h, mcr = MU_CR / h, mci = MU_CI / h;
    for (int k = k0; k < k1; ++k) {
        c->part_cur = (row_last && k == n_total - 1) ? row_last : ((row_prev && k == n_total - 2) ? row_prev : nullptr);
        NK2D_TRY(newton_iteration(s, mreal, mcr, mci));
        c->st.nfev += 3;
        c->st.nnewton++;
    }
    c->part_cur = nullptr;
    return 0;
}

// the launches of one simplified-Newton iteration that change nothing the integrator keeps
// (right-hand sides and ping-pong iterates only): all but the last, which carries the update
int newton_front(Ctl& s, double mreal, double mcr, double mci) {
    const int m = std::max(s.m_real, s.m_cplx);
    int src = 0;
    for (int it = 0; it + 1 < m; ++it) {
        NK2D_TRY(nk2d_r_newton_fused(s.c, it == 0, it == 0, false, mreal, mcr, mci, src, m == 2));
        src = 1 - src;
    }
    return 0;
}
int newton_back(Ctl& s, double mreal, double mcr, double mci) {
    const int m = std::max(s.m_real, s.m_cplx);
    NK2D_TRY(nk2d_r_newton_fused(s.c, m == 1, m == 1, true, mreal, mcr, mci, (m - 1) & 1, m == 2));
    s.c->st.nsolve += 2;
    return 0;
}

// "the launch that carries these partials is queued" / "wait for it".  By launches: an event on the context's stream.
// As a command stream (nk2d_stream.h): nothing to record -- the partials were marked before their command went out and say
// themselves when they have arrived.
inline int mark_queued(nk2d_ctx* c, hipEvent_t ev) {
    if (c->stream_on) return 0;
    NK2D_CHECK(c, hipEventRecord(ev, nk2d_s(c)));
    return 0;
}
inline int await_partials(nk2d_ctx* c, hipEvent_t ev, const double* part) {
    if (c->stream_on) return nk2d_stream_wait_part(c, part, c->ncol);
    NK2D_CHECK(c, hipEventSynchronize(ev));
    return 0;
}

// simplified Newton iterations on the collocation system (radau.py:48-136), decisions on
// the host: one read-back of the per-column norm partials per iteration.  While the host waits
// for them the device already runs the front launches of the NEXT iteration; if the iteration
// stops here they were wasted work on buffers nobody reads.  force_iters is unused here (replay
// has newton_fixed) but kept for symmetry.
// err_buf != nullptr: when the known contraction rate predicts that the iteration in flight is the
// last one, the error estimate is queued right behind it (partials into hPART2); *err_buf >= 0 on
// return then names the ping-pong buffer holding it, and the caller only has to wait for snap_ev[1].
//
// A vector norm hook (nk2d_set_norm_hook_vec: a tracer module sharded over contexts, every norm the controller reads an
// all-reduce): the sum of the iteration being judged travels in ONE call with the sums of up to two iterations queued whole
// behind it, and with the error estimate queued behind the last of them when that one is expected to end the step (from
// the contraction rate once it is known, from the iteration count of the step before until then).  The later values
// are kept for the turns of the loop -- or the caller -- that would have asked for them.  Same sums, same tests, same order:
// an iteration or an estimate that turns out unwanted is dropped (its stage values sit in spare buffers).
int newton(Ctl& s, double h, int force_iters, bool* converged, int* n_iter, double* rate_out, bool* have_rate,
           int* err_buf = nullptr, double* err_sum_coupled = nullptr, bool* err_sum_is_coupled = nullptr) {
    nk2d_ctx* c = s.c;
    const double mreal = MU_REAL / h, mcr = MU_CR / h, mci = MU_CI / h;
    const bool single = std::max(s.m_real, s.m_cplx) == 1;   // the whole iteration is ONE launch
    double dW_norm_old = 0.0, rate = 0.0;
    bool has_old = false, has_rate = false;
    *converged = false;
    const int kmax = force_iters >= 0 ? force_iters : NEWTON_MAXITER;
    const bool speculate = c->part_on_host && force_iters < 0 && c->speculate;
    const bool pairing = c->norm_hook_vec != nullptr && speculate;
    const int max_depth = (pairing && c->ZS) ? std::max(1, std::min(2, c->hook_spec_depth)) : 1;
    const bool whole = single || (pairing && c->swap_updates);   // an iteration can be queued whole, and dropped
    // work queued ahead of the verdict on the iteration before: the launches before the last one (several sweeps),
    // or whole iterations (single launch: the update goes to a spare stage buffer and to a partial buffer of its own,
    // so an unwanted one is dropped by swapping the stage buffers back)
    bool front_queued = false;
    int launched = -1;        // the last iteration queued whole
    double held[2] = {0.0, 0.0};   // module-wide sums of iterations held_from, held_from + 1 (they came with an earlier one)
    int held_from = 0, held_n = 0;
    double held_err = 0.0;
    bool have_held_err = false;
    int err_behind = -1;      // the iteration the queued error estimate follows
    if (err_buf) *err_buf = -1;
    if (err_sum_is_coupled) *err_sum_is_coupled = false;
    int k = -1;
    auto part_of = [&](int it) { return !whole ? c->hPART : (it % 3 == 0 ? c->hPART : (it % 3 == 1 ? c->hPARTB : c->hPARTC)); };
    auto event_of = [&](int it) { return whole ? c->snap_ev[2 + it % 3] : c->snap_ev[0]; };
    for (k = 0; k < kmax; ++k) {
        if (err_buf && err_behind != k) {   // an estimate queued behind a non-final iteration is void
            if (*err_buf >= 0) c->cnt_err_void++;
            *err_buf = -1;
        }
        if (k > launched) {
            const bool timed = !front_queued;   // front and back launches are queued back to back
            if (timed) {
                NK2D_TRY(nk2d_prof_window_begin(c));
                NK2D_TRY(newton_front(s, mreal, mcr, mci));
            }
            front_queued = false;
            c->part_cur = c->part_on_host ? part_of(k) : nullptr;
            NK2D_TRY(newton_back(s, mreal, mcr, mci));
            if (timed) NK2D_TRY(nk2d_prof_window_end(c));
            if (speculate) NK2D_TRY(mark_queued(c, event_of(k)));
            launched = k;
        }
        c->st.nfev += 3;
        c->st.nnewton++;
        if (force_iters >= 0) continue;
        double sum = 0.0;
        bool coupled = false;
        if (held_n > 0 && k >= held_from && k < held_from + held_n) {
            // came with the iteration that queued this one: nothing is queued ahead of the verdict on this one
            sum = held[k - held_from];
            coupled = true;
        } else if (speculate) {
            held_n = 0;
            have_held_err = false;
            // which iteration is expected to pass SciPy's convergence test?  (far: none in sight)
            const int far = k + 8;
            int last_pred = far;
            if (has_rate && rate < 1.0) {
                double next_norm = rate * dW_norm_old;
                for (int j = 0; j <= 2; ++j, next_norm *= rate)
                    if (rate / (1.0 - rate) * next_norm < c->spec_bias * s.newton_tol) { last_pred = k + j; break; }
            } else if (pairing && !has_rate) {
                last_pred = s.n_iter_prev > 0 ? std::max(k, s.n_iter_prev - 1) : k + 1;
            }
            int depth = 0;
            bool want_err = false;
            if (pairing) {
                if (whole) depth = std::max(0, std::min(std::min(max_depth, last_pred - k), kmax - 1 - k));
                want_err = err_buf && err_sum_coupled && s.m_real <= 2 && k + depth == last_pred;
            } else {
                if (single && last_pred > k && k + 1 < kmax) depth = 1;
                want_err = err_buf && s.m_real <= 2 && last_pred == k;
            }
            for (int j = k + 1; j <= k + depth; ++j) {
                if (j - k == 2) std::swap(c->ZN, c->ZS);     // keep the stage values of iteration k
                if (!single) NK2D_TRY(newton_front(s, mreal, mcr, mci));
                c->part_cur = part_of(j);
                NK2D_TRY(newton_back(s, mreal, mcr, mci));
                NK2D_TRY(mark_queued(c, event_of(j)));
                launched = j;
            }
            if (!whole && last_pred > k && k + 1 < kmax) {
                NK2D_TRY(newton_front(s, mreal, mcr, mci));
                front_queued = true;
            }
            if (want_err) {
                c->cnt_err_queued++;
                NK2D_TRY(nk2d_r_err_fused(c, h, s.m_real, err_buf, c->hPART2));
                NK2D_TRY(mark_queued(c, c->snap_ev[1]));
                err_behind = k + depth;
            }
            if (pairing && (depth > 0 || want_err)) {
                double v[4];
                int n = 0;
                NK2D_CHECK(c, hipEventSynchronize(want_err ? c->snap_ev[1] : event_of(k + depth)));
                for (int j = 0; j <= depth; ++j) NK2D_TRY(nk2d_part_sum(c, c->ncol, &v[n++], part_of(k + j)));
                if (want_err) NK2D_TRY(nk2d_part_sum(c, c->ncol, &v[n++], c->hPART2));
                c->norm_hook_vec(c->norm_hook_vec_user, v, n);
                sum = v[0];
                held_from = k + 1;
                held_n = depth;
                for (int j = 0; j < depth; ++j) held[j] = v[1 + j];
                if (want_err) { held_err = v[n - 1]; have_held_err = true; }
                coupled = true;
            } else {
                NK2D_TRY(await_partials(c, event_of(k), part_of(k)));
                NK2D_TRY(nk2d_part_sum(c, c->ncol, &sum, part_of(k)));
            }
        } else {
            NK2D_TRY(nk2d_k_reduce(c, c->ncol, 1, &sum));
        }
        if (!coupled) couple(c, &sum);
        const double dW_norm = rms_from_sum(sum, 3.0 * s.n_total);
        bool stop = false;
        if (!(dW_norm == dW_norm)) stop = true;  // NaN: treat as divergence
        if (!stop) {
            if (has_old) { rate = dW_norm / dW_norm_old; has_rate = true; }
            if (has_rate && (rate >= 1.0 || std::pow(rate, NEWTON_MAXITER - k) / (1.0 - rate) * dW_norm > s.newton_tol)) stop = true;
        }
        if (!stop && (dW_norm == 0.0 || (has_rate && rate / (1.0 - rate) * dW_norm < s.newton_tol))) {
            *converged = true;
            stop = true;
        }
        if (stop) {
            // iterations queued ahead are not wanted: the stage values of this one sit in a spare buffer, swap back
            const int unwind = launched - k;
            if (unwind == 1) std::swap(c->Z, c->ZN);
            else if (unwind == 2) std::swap(c->Z, c->ZS);
            c->st.nsolve -= 2 * unwind;
            c->cnt_spec_dropped += (int64_t)unwind * std::max(s.m_real, s.m_cplx);
            if (front_queued) c->cnt_front_dropped += std::max(s.m_real, s.m_cplx) - 1;
            if (err_buf && err_behind != k) {
                if (*err_buf >= 0) c->cnt_err_void++;
                *err_buf = -1;    // ... nor is an estimate behind them
            } else if (*converged && err_buf && err_behind == k && have_held_err) {
                // the estimate queued behind this iteration came with the norms: the caller has nothing to wait for
                *err_sum_coupled = held_err;
                *err_sum_is_coupled = true;
            }
            break;
        }
        dW_norm_old = dW_norm;
        has_old = true;
    }
    c->part_cur = nullptr;
    if (force_iters >= 0) { *converged = true; *n_iter = force_iters; }
    else *n_iter = k + 1;
    if (*converged && force_iters < 0) s.n_iter_prev = k + 1;
    *rate_out = rate;
    *have_rate = has_rate;
    return 0;
}

double predict_factor(double h_abs, bool has_h_old, double h_abs_old, double err, bool has_err_old, double err_old) {
    double mult = 1.0;
    if (has_err_old && has_h_old && err != 0.0) mult = h_abs / h_abs_old * std::pow(err_old / err, 0.25);
    return std::min(1.0, mult) * std::pow(err, -0.25);
}

int initial_step(Ctl& s, double* h_out) {
    nk2d_ctx* c = s.c;
    const double t0 = s.t, interval = std::fabs(s.t1 - t0);
    if (interval == 0.0) { *h_out = 0.0; return 0; }
    double s0 = 0, s1 = 0, s2 = 0;
    NK2D_TRY(nk2d_r_wnorm(c, c->Y, nullptr, 1.0, 0.0, c->Y));
    NK2D_TRY(nk2d_k_reduce(c, c->ncol, 1, &s0));
    couple(c, &s0);
    NK2D_TRY(nk2d_r_wnorm(c, c->F, nullptr, 1.0, 0.0, c->Y));
    NK2D_TRY(nk2d_k_reduce(c, c->ncol, 1, &s1));
    couple(c, &s1);
    const double d0 = rms_from_sum(s0, s.n_total), d1 = rms_from_sum(s1, s.n_total);
    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    h0 = std::min(h0, interval);
    NK2D_TRY(nk2d_r_axpy(c, c->Y, h0 * 1.0, c->F, c->TMP));   // y1 = y0 + h0*direction*f0
    NK2D_TRY(eval_kv(c, t0 + h0 * 1.0, 4));
    NK2D_TRY(nk2d_k_tend(c, c->TMP, c->KV[4], c->TMP2));        // f1
    c->st.nfev++;
    NK2D_TRY(nk2d_r_wnorm(c, c->TMP2, c->F, 1.0, -1.0, c->Y));
    NK2D_TRY(nk2d_k_reduce(c, c->ncol, 1, &s2));
    couple(c, &s2);
    const double d2 = rms_from_sum(s2, s.n_total) / h0;
    double h1;
    if (d1 <= 1e-15 && d2 <= 1e-15) h1 = std::max(1e-6, h0 * 1e-3);
    else h1 = std::pow(0.01 / std::max(d1, d2), 1.0 / (3 + 1));
    *h_out = std::min(std::min(100 * h0, h1), std::min(interval, s.max_step));
    return 0;
}

// commit an accepted step: y_old <- y, y <- y + Z2, Z_prev <- Z.  with_tend: also F <- fun(t_new,
// y_new) in the same launch; KV[3] must then hold the vertical mixing plane at t_new.
int commit_step(Ctl& s, double t, double t_new, bool with_tend = false) {
    nk2d_ctx* c = s.c;
    if (with_tend) NK2D_TRY(nk2d_r_commit_tend(c, c->KV[3]));
    else NK2D_TRY(nk2d_r_axpy(c, c->Y, 1.0, c->Z + 2 * c->nv, c->YOLD));  // y_new into the spare buffer
    std::swap(c->Y, c->YOLD);                                          // Y = y_new, YOLD = y
    std::swap(c->Z, c->ZP);                                            // ZP = Z of this step
    s.have_dense = true;
    s.dense_t_old = t;
    s.dense_h = t_new - t;
    s.t = t_new;
    c->st.nsteps++;
    return 0;
}

int run_free(Ctl& s, double* record, int64_t record_cap, int64_t* record_n) {
    nk2d_ctx* c = s.c;
    int64_t nrec = 0;
    while (s.t < s.t1) {
        const double t = s.t;
        const double min_step = 10.0 * std::fabs(std::nextafter(t, std::numeric_limits<double>::infinity()) - t);
        double h_abs, h_abs_old = 0, err_old = 0;
        bool has_h_old, has_err_old;
        if (s.h_abs > s.max_step) { h_abs = s.max_step; has_h_old = has_err_old = false; }
        else if (s.h_abs < min_step) { h_abs = min_step; has_h_old = has_err_old = false; }
        else { h_abs = s.h_abs; h_abs_old = s.h_abs_old; err_old = s.err_old; has_h_old = s.has_old_h; has_err_old = s.has_old_err; }
        const bool jac_needs_state0 = c->kind == 1 || (c->kind == 2 && c->d.sms_nrec > 0 && c->d.sink_thres > 0.0);
        // (modules whose Jacobian reads the state: only with option "jac_stage_state" -- the mixing plane of the stage time,
        // the state of the step start, in a launch of its own)
        const bool jac_at_stage = c->jac_stage >= 0 && c->jac_fresh && (!jac_needs_state0 || c->jac_stage_state);
        const int jst_inlaunch = (jac_at_stage && !jac_needs_state0) ? c->jac_stage : -1;
        if (c->jac_fresh && !s.current_jac && !jac_at_stage) {
            // evaluating J costs two small launches here (SciPy pays a Python double loop and two
            // SuperLU factorisations, hence its reuse heuristics): never start a step on a stale J
            // (normally done by the step boundary launch of the step before)
            NK2D_TRY(refresh_jac(s, t, true));
            c->st.njev++;
            s.current_jac = true;
            s.have_lu = false;
        }
        bool rejected = false, accepted = false, newton_failed = false;
        double h = 0, t_new = 0, err = 0, safety = 0, rate = 0;
        bool have_rate = false;
        int n_iter = 0;
        while (!accepted) {
            // a NaN / Inf in the state or the tendency makes every comparison below false: SciPy's loop
            // (radau.py:445-476) would halve NaN forever; fail instead
            if (!std::isfinite(h_abs)) return nk2d_fail(c, "Radau: step size is not finite (non-finite state or tendency)", -3);
            if (h_abs < min_step) return nk2d_fail(c, "Radau: required step size is less than spacing between numbers", -3);
            h = h_abs;
            t_new = t + h;
            if (t_new - s.t1 > 0) t_new = s.t1;
            h = t_new - t;
            h_abs = std::fabs(h);
            // host control: planes and prediction of the first try in one launch
            bool predicted = false, jac_done = false;
            if (s.pre_setup && s.pre_t == t && s.pre_h == h) {
                predicted = true;            // queued with the commit of the step before
                jac_done = s.pre_jac;
            } else if (s.have_dense) {
                NK2D_TRY(setup_attempt(s, t, h, jst_inlaunch));
                predicted = true;
                jac_done = jst_inlaunch >= 0;
            } else {
                NK2D_TRY(stage_planes(s, t, h));
            }
            s.pre_setup = false;
            if (jac_at_stage) {
                // option "jac_stage": the Jacobian of this attempt from the vertical mixing plane of one of ITS stage
                // times instead of the step start (normally derived by the launch that computed the plane)
                if (!jac_done) NK2D_TRY(nk2d_k_jac(c, c->KV[c->jac_stage], jac_needs_state0 ? c->Y : nullptr));
                s.t_jac = t + (h * RC[c->jac_stage]);
                c->st.njev++;
                s.current_jac = true;
                s.have_lu = false;
            }
            bool converged = false;
            double err_sum = 0.0;
            int buf = 0;
            int queued_err = -1;   // >= 0: the error estimate is already queued, in XR[queued_err]
            double err_coupled = 0.0;      // ... and (vector norm hook) its module-wide sum came with the last Newton norm
            bool err_is_coupled = false;
            while (!converged) {
                if (!s.have_lu) NK2D_TRY(set_lu(s, h));
                if (!predicted) NK2D_TRY(predict(s, t, h));
                predicted = false;
                NK2D_TRY(newton(s, h, -1, &converged, &n_iter, &rate, &have_rate, &queued_err, &err_coupled, &err_is_coupled));
                if (!converged) {
                    if (s.current_jac) break;
                    NK2D_TRY(refresh_jac(s, t, true));  // KV[3] holds the plane at the current t
                    c->st.njev++;
                    s.current_jac = true;
                    s.have_lu = false;
                }
            }
            if (!converged) {
                h_abs *= 0.5;
                s.have_lu = false;
                newton_failed = true;
                continue;
            }
            // error estimate (radau.py:477-487)
            double sum = err_sum;
            bool sum_coupled = false;
            if (queued_err >= 0) {
                buf = queued_err;
                if (err_is_coupled) {
                    sum = err_coupled;
                    sum_coupled = true;
                } else {
                    NK2D_TRY(await_partials(c, c->snap_ev[1], c->hPART2));
                    NK2D_TRY(nk2d_part_sum(c, c->ncol, &sum, c->hPART2));
                }
            } else {
                if (s.m_real <= 2) {
                    NK2D_TRY(nk2d_r_err_fused(c, h, s.m_real, &buf, nullptr));
                } else {
                    NK2D_TRY(nk2d_r_err_rhs(c, h));
                    NK2D_TRY(solve_systems(s, true, false, &buf));
                    NK2D_TRY(nk2d_r_err_norm(c, c->XR[buf]));
                }
                NK2D_TRY(nk2d_k_reduce(c, c->ncol, 1, &sum));
            }
            if (!sum_coupled) couple(c, &sum);
            err = rms_from_sum(sum, s.n_total);
            safety = 0.9 * (2 * NEWTON_MAXITER + 1) / (2 * NEWTON_MAXITER + n_iter);
            if (rejected && err > 1) {
                NK2D_TRY(nk2d_r_copy(c, c->TMP, c->XR[buf]));
                NK2D_TRY(nk2d_r_err_rhs2(c, c->TMP, h));
                c->st.nfev++;
                NK2D_TRY(solve_systems(s, true, false, &buf));
                NK2D_TRY(nk2d_r_err_norm(c, c->XR[buf]));
                NK2D_TRY(nk2d_k_reduce(c, c->ncol, 1, &sum));
                couple(c, &sum);
                err = rms_from_sum(sum, s.n_total);
            }
            if (err > 1) {
                const double factor = predict_factor(h_abs, has_h_old, h_abs_old, err, has_err_old, err_old);
                h_abs *= std::max(MIN_FACTOR, safety * factor);
                s.have_lu = false;
                rejected = true;
                c->st.nrejected++;
            } else {
                accepted = true;
            }
        }
        const bool recompute_jac = n_iter > 2 && have_rate && rate > 1e-3;
        double factor = predict_factor(h_abs, has_h_old, h_abs_old, err, has_err_old, err_old);
        factor = std::min(MAX_FACTOR, safety * factor);
        // option "growth_cap" (RADAU5's rule with 1.0: after a step whose simplified Newton iteration failed and
        // had to be repeated with half the step size, the next step is not allowed to grow -- Hairer & Wanner's
        // radau5.f, `IF (REJECT) HNEW = MIN(HNEW, H)`; SciPy's Radau has no such memory and tries up to 10 h again)
        if (newton_failed && c->growth_cap > 0.0) factor = std::min(factor, c->growth_cap);
        const double h_lu_used = s.h_lu;
        if (!recompute_jac && factor < 1.2) factor = 1;
        else s.have_lu = false;
        if (record && nrec < record_cap) {
            double* r = record + nrec * NK2D_SCHED_WIDTH;
            r[0] = t; r[1] = t_new; r[2] = h; r[3] = (double)n_iter; r[4] = s.t_jac; r[5] = h_lu_used;
            r[6] = err; r[7] = s.fingerprint;
        }
        ++nrec;
        // y_new, f_new = fun(t_new, y_new)
        const double h_abs_next = h_abs * factor;
        // One launch for the whole boundary where nothing in between needs the host: the commit, the Jacobian at
        // t_new when one is due (SciPy's recompute_jac, or the engines' Jacobian at every step start) and the set-up
        // of the next step's first attempt, whose step size is known now.  Steps with a history sample and the
        // device-side controllers keep the separate launches.
        const bool jac_due = recompute_jac || c->jac_fresh;
        const bool jac_needs_state = c->kind == 1 || (c->kind == 2 && c->d.sms_nrec > 0 && c->d.sink_thres > 0.0);
        // (a year with history samples: only the steps that hold a sample time keep the separate launches)
        const bool sample_due = c->hist_n > 0 && c->hist_next < c->hist_n && c->hist_t[c->hist_next] <= t_new;
        bool fused = !sample_due && t + h == t_new && t_new < s.t1;
        double h2 = 0.0;
        if (fused) {
            // the next step's first attempt, as the top of this loop will compute it
            const double min_step2 = 10.0 * std::fabs(std::nextafter(t_new, std::numeric_limits<double>::infinity()) - t_new);
            double h_abs2 = h_abs_next;
            if (h_abs2 > s.max_step) h_abs2 = s.max_step;
            else if (h_abs2 < min_step2) h_abs2 = min_step2;
            double t_new2 = t_new + h_abs2;
            if (t_new2 - s.t1 > 0) t_new2 = s.t1;
            h2 = t_new2 - t_new;
            fused = std::isfinite(h2) && h2 > 0.0;
        }
        if (fused) {
            double times[3], x[3];
            for (int i = 0; i < 3; ++i) {
                times[i] = t_new + (h2 * RC[i]);
                x[i] = ((t_new + h2 * RC[i]) - t) / (t_new - t);
            }
            // planes: the third stage plane of this step IS the plane at t_new; the new stage planes go to the two
            // stage buffers nobody needs any more and to the buffer of the plane at the old t
            double* out[3] = {c->KV[0], c->KV[1], c->KV[3]};
            NK2D_TRY(nk2d_r_step_boundary(c, c->KV[2], jac_due && !jac_needs_state && !jac_at_stage, times, out, x[0], x[1], x[2],
                                          jst_inlaunch));
            std::swap(c->KV[3], c->KV[2]);
            std::swap(c->Y, c->YOLD);
            std::swap(c->Z, c->ZP);
            s.have_dense = true;
            s.dense_t_old = t;
            s.dense_h = t_new - t;
            s.t = t_new;
            c->st.nsteps++;
            c->st.nfev++;
            s.pre_setup = true; s.pre_t = t_new; s.pre_h = h2; s.pre_jac = jst_inlaunch >= 0;
            if (jac_at_stage) {
                s.current_jac = false;   // evaluated inside the next attempt
            } else if (jac_due) {
                // a Jacobian that reads the state (phosphorus, forced with a sink threshold) needs y_new complete:
                // its own launch, after the boundary
                if (jac_needs_state) NK2D_TRY(refresh_jac(s, t_new, true));
                s.t_jac = t_new;
                c->st.njev++;
                s.current_jac = true;
                if (!recompute_jac) s.have_lu = false;   // the Jacobian at every step start drops the factorisation
            } else {
                s.current_jac = false;
            }
        } else {
            if (t + h == t_new) std::swap(c->KV[3], c->KV[2]);  // stage-3 plane is the plane at t_new
            else NK2D_TRY(eval_kv(c, t_new, 3));
            NK2D_TRY(commit_step(s, t, t_new, true));
            c->st.nfev++;
            if (c->hist_n > 0) NK2D_TRY(nk2d_hist_sample(c, t, t_new, nrec == 1));
            if (recompute_jac) {
                NK2D_TRY(refresh_jac(s, t_new, true));
                c->st.njev++;
                s.current_jac = true;
            } else {
                s.current_jac = false;
            }
        }
        s.h_abs_old = s.h_abs; s.has_old_h = true;
        s.err_old = err; s.has_old_err = true;
        s.h_abs = h_abs_next;
    }
    if (record_n) *record_n = nrec;
    // (the context's own record buffer is a convenience: a year longer than it simply leaves no schedule behind)
    if (record && nrec > record_cap && record != c->own_rec.data())
        return nk2d_fail(c, "nk2d_comp_fcn: schedule record buffer too small", -4);
    return 0;
}

// Step replay: the accepted steps of a recorded year (SciPy's, for the 1e-10 parity checks; this library's own, for
// the finite-difference products with a frozen controller) with no decisions and nothing read back -- the whole year
// is queued without a host round trip.  With host-side launches (device_ctl 0) the year runs on the launches of the
// free-running loop: step boundary (commit + planes + predicted stage values + Jacobian of the next row) and fused
// Newton iterations; a row whose Jacobian time is a stage time of its own attempt takes the Jacobian from that
// stage's plane, as the free run with option "jac_stage" does.
//
// A frozen year (check = true: a schedule this library recorded, replayed for another state) additionally
//   * keeps the norm partials of the last two Newton iterations of every step (rows of STEP_PART) for the check after
//     the year, and -- every err_every-th step -- evaluates SciPy's error estimate (a third row);
//   * keeps the state every NK2D_CKPT_EVERY steps (Y, YOLD, ZP), so that a year whose recorded Newton iteration count
//     turns out not to be enough at some step can be resumed from the checkpoint before it (run_frozen).
struct ReplayLocal {
    double h_lu_cur = 0.0;
    bool have = false;
    int pre_jstage = -1;     // stage whose Jacobian the boundary launch of the row before derived for this row
    bool kv3_at_t = true;    // KV[3] holds the mixing plane at the current t (year start; after a boundary launch)
    bool f_at_t = true;      // F holds the tendency at the current (t, y) (year start; after a boundary launch with it)
    std::vector<char>* err_done = nullptr;   // per row: its error estimate was evaluated in this pass
};

// steps whose error estimate a frozen year evaluates
inline bool err_checked(const nk2d_ctx* c, bool check, int64_t i) {
    return check && c->frozen_err_check > 0 && (i % c->frozen_err_check) == 0;
}

int replay_rows(Ctl& s, const double* sched, int64_t n, int64_t i0, bool check, ReplayLocal& L) {
    nk2d_ctx* c = s.c;
    const bool needs_state = c->kind == 1 || (c->kind == 2 && c->d.sms_nrec > 0 && c->d.sink_thres > 0.0);
    const bool fast = c->hist_n == 0;
    // stage of the attempt (t, h) whose time is t_jac, or -1
    auto stage_of = [&](double t, double h, double t_jac) {
        if ((needs_state && !c->jac_stage_state) || !fast) return -1;
        for (int k = 0; k < 3; ++k)
            if (t_jac == t + (h * RC[k])) return k;
        return -1;
    };
    for (int64_t i = i0; i < n; ++i) {
        const double* r = sched + i * NK2D_SCHED_WIDTH;
        const double t = r[0], t_new = r[1], h = r[2], t_jac = r[4], h_lu = r[5];
        const int n_iter = (int)r[3];
        if (t != s.t) return nk2d_fail(c, "nk2d_comp_fcn: replay schedule does not start where the state is", -5);
        if (check && fast && (i % NK2D_CKPT_EVERY) == 0) {
            // checkpoint: the row before ended in a boundary launch of its own (below), so Y, YOLD and ZP are what a
            // restart needs (state, and the collocation polynomial of the last step for the predicted stage values)
            const size_t slot = (size_t)(i / NK2D_CKPT_EVERY);
            while (c->ckpt.size() <= slot) {
                double* buf = nullptr;
                NK2D_CHECK(c, hipMalloc((void**)&buf, sizeof(double) * 5 * c->nv));
                c->ckpt.push_back(buf);
            }
            double* buf = c->ckpt[slot];
            NK2D_TRY(nk2d_r_copy(c, buf, c->Y));
            NK2D_TRY(nk2d_r_copy(c, buf + c->nv, c->YOLD));
            for (int k = 0; k < 3; ++k) NK2D_TRY(nk2d_r_copy(c, buf + (2 + k) * c->nv, c->ZP + k * c->nv));
        }
        // a Jacobian that reads the state can only be refreshed where the state is: at a step start; the others are
        // functions of time alone (option "jac_stage": the recorded year took it at a stage time of the attempt)
        if (t_jac != s.t_jac && t_jac != t && needs_state && stage_of(t, h, t_jac) < 0)
            return nk2d_fail(c, "nk2d_comp_fcn: replay schedule refreshes the Jacobian off a step start", -5);
        const int jstage = (t_jac != s.t_jac) ? stage_of(t, h, t_jac) : -1;
        bool predicted = false, jac_done = false;
        const int jstage_inlaunch = needs_state ? -1 : jstage;    // a Jacobian that reads the state: a launch of its own
        if (s.pre_setup && s.pre_t == t && s.pre_h == h) {
            predicted = true;
            jac_done = jstage_inlaunch >= 0 && L.pre_jstage == jstage_inlaunch;
        } else if (fast && s.have_dense) {
            NK2D_TRY(setup_attempt(s, t, h, jstage_inlaunch));
            predicted = true;
            jac_done = jstage_inlaunch >= 0;
        } else {
            NK2D_TRY(stage_planes(s, t, h));
        }
        s.pre_setup = false;
        L.pre_jstage = -1;
        if (t_jac != s.t_jac) {
            if (jstage >= 0) {
                if (!jac_done) NK2D_TRY(nk2d_k_jac(c, c->KV[jstage], needs_state ? c->Y : nullptr));
                s.t_jac = t_jac;
            } else {
                NK2D_TRY(refresh_jac(s, t_jac, t_jac == t && L.kv3_at_t));   // the plane at t is at hand after a boundary launch
            }
            c->st.njev++;
            L.have = false;
        }
        if (!L.have || h_lu != L.h_lu_cur) { NK2D_TRY(set_lu(s, h_lu)); L.h_lu_cur = h_lu; L.have = true; }
        if (!predicted) NK2D_TRY(predict(s, t, h));
        double* row_last = check ? c->STEP_PART + (size_t)(3 * i) * c->ncol : nullptr;
        double* row_prev = check ? c->STEP_PART + (size_t)(3 * i + 1) * c->ncol : nullptr;
        double* row_err = check ? c->STEP_PART + (size_t)(3 * i + 2) * c->ncol : nullptr;
        // the error estimate of this step (frozen year, every err_every-th step): needs the tendency at the step start
        // and the stage values in memory, i.e. neither this step nor the one before may end in a final launch
        const bool want_err = err_checked(c, check, i) && fast && L.f_at_t && s.m_real <= 2 && n_iter >= 1;
        const bool next_err = err_checked(c, check, i + 1) && fast;
        // With a next row that starts where this one ends, its planes, predicted stage values and (where its Jacobian
        // time is the step start or one of its stage times) its Jacobian are computed before this step is left:
        const double* r2 = (i + 1 < n) ? r + NK2D_SCHED_WIDTH : nullptr;
        const bool chained = fast && r2 && t + h == t_new && r2[0] == t_new && r2[2] > 0.0 && std::isfinite(r2[2]);
        double h2 = 0.0, times[3] = {0, 0, 0}, x[3] = {0, 0, 0};
        int jstage2 = -1;
        bool jac_at_tnew = false;
        if (chained) {
            h2 = r2[2];
            const double t_jac2 = r2[4];
            const bool jac_new = t_jac2 != s.t_jac;
            jstage2 = (jac_new && !needs_state) ? stage_of(t_new, h2, t_jac2) : -1;
            jac_at_tnew = jac_new && jstage2 < 0 && t_jac2 == t_new && !needs_state;
            for (int k = 0; k < 3; ++k) {
                times[k] = t_new + (h2 * RC[k]);
                x[k] = ((t_new + h2 * RC[k]) - t) / (t_new - t);
            }
        }
        // ... in the launch that ends the last Newton iteration (nk2d_r_newton_final) where the Jacobian does not read
        // the state and no Jacobian at t_new is due; otherwise in a step boundary launch of its own (also before a
        // checkpoint and around a step whose error estimate is evaluated)
        const bool ckpt_next = check && ((i + 1) % NK2D_CKPT_EVERY) == 0;
        const bool final_fused = chained && !needs_state && !jac_at_tnew && n_iter >= 1 && c->final_fuse && !ckpt_next &&
                                 !want_err && !next_err;
        if (final_fused) {
            NK2D_TRY(newton_fixed(s, h, 0, n_iter - 1, n_iter, row_last, row_prev));
            const double mreal = MU_REAL / h, mcr = MU_CR / h, mci = MU_CI / h;
            const int m = std::max(s.m_real, s.m_cplx);
            int src = 0;
            c->part_cur = nullptr;
            for (int it = 0; it + 1 < m; ++it) {
                NK2D_TRY(nk2d_r_newton_fused(c, it == 0, it == 0, false, mreal, mcr, mci, src, m == 2));
                src = 1 - src;
            }
            c->part_cur = row_last;
            NK2D_TRY(nk2d_r_newton_final(c, m == 1, m == 1, mreal, mcr, mci, (m - 1) & 1, m == 2, times, x[0], x[1], x[2], jstage2));
            c->part_cur = nullptr;
            c->st.nsolve += 2;
            c->st.nfev += 3;
            c->st.nnewton++;
        } else {
            NK2D_TRY(newton_fixed(s, h, 0, n_iter, n_iter, row_last, row_prev));
        }
        if (want_err) {
            int buf = 0;
            NK2D_TRY(nk2d_r_err_fused(c, h, s.m_real, &buf, row_err));
            c->st.nerr_checked++;
            if (L.err_done) (*L.err_done)[(size_t)i] = 1;
        }
        if (chained) {
            if (!final_fused) {
                double* out[3] = {c->KV[0], c->KV[1], c->KV[3]};
                NK2D_TRY(nk2d_r_step_boundary(c, c->KV[2], jac_at_tnew, times, out, x[0], x[1], x[2], jstage2, next_err));
                if (next_err) c->st.nfev++;
                std::swap(c->KV[3], c->KV[2]);
                std::swap(c->Y, c->YOLD);
                std::swap(c->Z, c->ZP);
            }
            L.kv3_at_t = !final_fused;
            L.f_at_t = !final_fused && next_err;
            s.have_dense = true;
            s.dense_t_old = t;
            s.dense_h = t_new - t;
            s.t = t_new;
            c->st.nsteps++;
            s.pre_setup = true; s.pre_t = t_new; s.pre_h = h2;
            L.pre_jstage = jstage2;
            if (jac_at_tnew) {
                s.t_jac = t_new;
                c->st.njev++;
                L.have = false;
            }
        } else {
            NK2D_TRY(commit_step(s, t, t_new));
            L.kv3_at_t = false;
            L.f_at_t = false;
        }
        // nothing is read back during a replay: bound the depth of the launch queue (a year is 10^4 launches; the
        // counter-collecting profiler of this ROCm falls over behind a few thousand unsynchronised dispatches, as it
        // did behind the preconditioner's elimination) -- the host is far ahead at that point
        // and a drain every 64 steps costs forty hand-overs a year
        // (a command stream bounds its own depth: the host never runs more than half a ring ahead of the slowest workgroup)
        if ((i & 63) == 63 && !c->stream_on) NK2D_CHECK(c, hipStreamSynchronize(nk2d_s(c)));
    }
    return 0;
}

int run_replay(Ctl& s, const double* sched, int64_t n, bool check) {
    nk2d_ctx* c = s.c;
    if (!check) {
        ReplayLocal L;
        NK2D_TRY(replay_rows(s, sched, n, 0, false, L));
        if (c->stream_on) {
            const int erc = nk2d_stream_end(c);
            if (erc == NK2D_RC_STREAM_LOST) {
                c->stream_on = 0;
                if (++c->stream_lost >= 2) c->stream_years = 0;
                c->st.nbarrier_timeouts++;
                return 3;
            }
            if (erc != 0) return erc;
        }
        return 0;
    }
    // ---- a frozen year ------------------------------------------------------------------------------------------------
    if (n > NK2D_OWN_REC_CAP) return nk2d_fail(c, "nk2d_comp_fcn_frozen: schedule too long", -4);
    // the schedule must be this context's own, recorded under the options it has now: anything else is not the discrete
    // map the caller is differentiating (a side file from another run, an option changed in between)
    for (int64_t i = 0; i < n; ++i)
        if (sched[i * NK2D_SCHED_WIDTH + 7] != s.fingerprint)
            return nk2d_fail(c, "nk2d_comp_fcn_frozen: the schedule was recorded under other options, another grid or another "
                                "build of the library (fingerprint of step " + std::to_string(i) + ")", -8);
    if (c->step_part_rows < (size_t)(3 * n)) {
        // norm partials of the last two Newton iterations and of the error estimate of every step, one row of ncol each
        NK2D_CHECK(c, hipStreamSynchronize(nk2d_s(c)));
        if (c->STEP_PART) NK2D_CHECK(c, hipFree(c->STEP_PART));
        c->STEP_PART = nullptr;
        c->step_part_rows = 0;
        NK2D_CHECK(c, hipMalloc((void**)&c->STEP_PART, sizeof(double) * (size_t)(3 * n) * c->ncol));
        c->step_part_rows = (size_t)(3 * n);
    }
    std::vector<double> mine;            // the schedule with the iteration counts a resume has raised
    const double* cur = sched;
    std::vector<double> sums((size_t)3 * n);
    const double slack = 30.0;
    // a sharded module (norm hook) checks its own tracers against their own count: nothing is exchanged, and if
    // every shard passes so does the module
    const bool hooked = c->norm_hook != nullptr;
    const double n_unknowns = hooked ? (double)c->tc * c->nz * c->ny : s.n_total;
    // SciPy's convergence test on the last recorded iteration of the rows from `from` on: the first row that fails, or -1
    auto first_unconverged = [&](const double* rows, int64_t from) -> int64_t {
        for (int64_t i = from; i < n; ++i) {
            const int n_it = (int)rows[i * NK2D_SCHED_WIDTH + 3];
            if (n_it < 1) continue;
            const double s_last = sums[3 * i];
            const double s_prev = (n_it >= 2) ? sums[3 * i + 1] : -1.0;
            const double n_last = rms_from_sum(s_last, 3.0 * n_unknowns);
            bool ok = n_last == n_last;
            if (ok && n_last != 0.0) {
                if (s_prev >= 0.0) {
                    const double n_prev = rms_from_sum(s_prev, 3.0 * n_unknowns);
                    const double rate = (n_prev > 0.0) ? n_last / n_prev : 0.0;
                    ok = rate < 1.0 && rate / (1.0 - rate) * n_last < slack * s.newton_tol;
                } else {
                    ok = n_last < slack * s.newton_tol;   // one iteration: the recorded year's first correction vanished
                }
            }
            if (!ok) return i;
        }
        return -1;
    };
    auto fetch_sums = [&](int64_t from) -> int {
        NK2D_TRY(nk2d_r_rows_sum(c, c->STEP_PART + (size_t)(3 * from) * c->ncol, 3 * (n - from), c->STEP_NORM));
        NK2D_CHECK(c, hipMemcpyAsync(sums.data() + 3 * from, c->STEP_NORM, sizeof(double) * 3 * (n - from),
                                     hipMemcpyDeviceToHost, nk2d_s(c)));
        NK2D_CHECK(c, hipStreamSynchronize(nk2d_s(c)));
        return 0;
    };
    // ---- small grids: the whole year in one launch on the schedule cache (k_frozen_persistent).  Its steps are checked like
    // any frozen year's; one that does not pass -- or a barrier that timed out -- hands the year back, from x, to the
    // launch-per-phase path below with its checkpoints and resumes (return 3: the caller restarts the year)
    // the sampled error estimates of the rows flagged in `done`, from row `from` up to (not including) row `upto` (-1: all):
    // the first row whose estimate exceeds what the recorded step was accepted with, or -1
    auto first_bad_err = [&](const double* rows, int64_t from, int64_t upto, const std::vector<char>& done) -> int64_t {
        double max_err = 0.0;
        int64_t bad_err = -1;
        for (int64_t i = from; i < n; ++i) {
            if (upto >= 0 && i >= upto) break;       // beyond a step that did not converge the state means nothing
            if (!done[(size_t)i]) continue;
            const double err = rms_from_sum(sums[3 * i + 2], n_unknowns);
            if (!(err == err)) { bad_err = i; break; }
            max_err = std::max(max_err, err);
            const double base = rows[i * NK2D_SCHED_WIDTH + 6];
            if (err > 1.5 * std::max(1.0, base) && bad_err < 0) bad_err = i;
        }
        c->st.max_err = std::max(c->st.max_err, max_err);
        return bad_err;
    };
    auto err_failure = [&](int64_t bad_err) {
        c->frozen_fallbacks++;
        return nk2d_fail(c, "nk2d_comp_fcn_frozen: the error estimate of step " + std::to_string(bad_err) + " of " +
                            std::to_string(n) + " exceeds what the recorded step was accepted with: the recorded steps "
                            "do not control the error for this state", -7);
    };
    if (!s.no_persistent && !c->stream_on) {
        std::vector<char> sampled;
        const int prc = nk2d_frozen_persistent(c, sched, n, &sampled);
        if (prc < 0) return prc;
        if (prc == 0) {
            const double* r = sched + (n - 1) * NK2D_SCHED_WIDTH;
            s.t = r[0];
            NK2D_TRY(commit_step(s, r[0], r[1]));
            NK2D_TRY(fetch_sums(0));
            if (first_unconverged(sched, 0) < 0) {
                // (not with a norm hook -- but a hooked context never takes this way)
                const int64_t bad_err = first_bad_err(sched, 0, -1, sampled);
                if (bad_err >= 0) return err_failure(bad_err);
                c->frozen_persistent_years++;
                return 0;
            }
            return 3;
        }
        if (prc == 2) { c->st.nbarrier_timeouts++; return 3; }
    }
    int64_t start = 0;
    std::vector<char> err_done((size_t)n, 0);
    for (int round = 0;; ++round) {
        ReplayLocal L;
        L.err_done = &err_done;
        std::fill(err_done.begin() + start, err_done.end(), 0);
        if (round > 0) {
            // resume at row `start` (a multiple of NK2D_CKPT_EVERY): state and collocation polynomial from the checkpoint;
            // planes, predicted stage values, Jacobian and factorisation are recomputed by the launches of a first row
            const double* buf = c->ckpt[(size_t)(start / NK2D_CKPT_EVERY)];
            NK2D_CHECK(c, hipMemcpyAsync(c->Y, buf, sizeof(double) * c->nv, hipMemcpyDeviceToDevice, nk2d_s(c)));
            NK2D_CHECK(c, hipMemcpyAsync(c->YOLD, buf + c->nv, sizeof(double) * c->nv, hipMemcpyDeviceToDevice, nk2d_s(c)));
            NK2D_CHECK(c, hipMemcpyAsync(c->ZP, buf + 2 * c->nv, sizeof(double) * 3 * c->nv, hipMemcpyDeviceToDevice, nk2d_s(c)));
            const double* r = cur + start * NK2D_SCHED_WIDTH;
            s.t = r[0];
            s.have_dense = start > 0;
            if (start > 0) {
                const double* rp = r - NK2D_SCHED_WIDTH;
                s.dense_t_old = rp[0];
                s.dense_h = rp[1] - rp[0];
            }
            s.pre_setup = false;
            s.t_jac = std::numeric_limits<double>::quiet_NaN();    // whatever Jacobian is in place is not this row's
            L.kv3_at_t = false;
            L.f_at_t = false;
        }
        NK2D_TRY(replay_rows(s, cur, n, start, true, L));
        if (c->stream_on) {
            // the year ran as a command stream: the kernel ends here; had it given up on the way, the year is handed back
            // (return 3: the caller restarts it from x, launch by launch)
            const int erc = nk2d_stream_end(c);
            if (erc == NK2D_RC_STREAM_LOST) {
                c->stream_on = 0;
                if (++c->stream_lost >= 2) c->stream_years = 0;
                c->st.nbarrier_timeouts++;
                return 3;
            }
            if (erc != 0) return erc;
        }
        // SciPy's convergence test (radau.py:120-129) on what the LAST recorded iteration of every step left, with
        // slack: the perturbed state of a finite-difference product converges like the state the schedule was
        // recorded for, give or take; a state that does not (the recorded year converged at once on a special
        // structure, say) must not be integrated with its iteration counts
        NK2D_TRY(fetch_sums(start));
        const int64_t bad = first_unconverged(cur, start);
        const std::string why = "the recorded Newton iteration count does not converge for this state";
        // the sampled error estimates (not with a norm hook: a shard sees only its own tracers' share)
        const int64_t bad_err = hooked ? -1 : first_bad_err(cur, start, bad, err_done);
        if (bad_err >= 0) return err_failure(bad_err);
        if (bad < 0) return 0;
        // one more Newton iteration at the first step that did not converge, from the checkpoint before it -- where that
        // can be done (host-launched replay, SciPy's cap of six iterations, two resumes per year)
        const int n_it = (int)cur[bad * NK2D_SCHED_WIDTH + 3];
        const bool can = c->hist_n == 0 && round < 2 && n_it < NEWTON_MAXITER &&
                         c->ckpt.size() > (size_t)(bad / NK2D_CKPT_EVERY);
        if (!can) {
            c->frozen_fallbacks++;
            return nk2d_fail(c, "nk2d_comp_fcn_frozen: " + why + " (step " + std::to_string(bad) + " of " + std::to_string(n) +
                                (round > 0 ? ", after " + std::to_string(round) + " resume(s)" : "") + ")", -7);
        }
        if (mine.empty()) { mine.assign(sched, sched + (size_t)n * NK2D_SCHED_WIDTH); cur = mine.data(); }
        mine[(size_t)bad * NK2D_SCHED_WIDTH + 3] = (double)(n_it + 1);
        start = (bad / NK2D_CKPT_EVERY) * NK2D_CKPT_EVERY;
        // a checkpoint is a place to resume only if its row computes its own Jacobian (the default mode: every row does; a
        // SciPy-mode schedule may carry one over from an earlier step, whose state is gone)
        while (start > 0 && cur[start * NK2D_SCHED_WIDTH + 4] == cur[(start - 1) * NK2D_SCHED_WIDTH + 4]) start -= NK2D_CKPT_EVERY;
        c->frozen_resumes++;
        c->st.nresumed++;
    }
}

}  // namespace

int nk2d_radau_year(nk2d_ctx* c, nk2d_vec x, nk2d_vec fx, nk2d_stats* stats, const double* replay, int64_t replay_n,
                    double* record, int64_t record_cap, int64_t* record_n, bool replay_own) {
    const auto wall0 = std::chrono::steady_clock::now();
    c->st = nk2d_stats();
    if (c->strm) nk2d_stream_part_forget(c, nullptr);     // (every partial buffer stands for itself until a command takes its name)
    // every free-running year leaves its accepted steps behind (nk2d_last_schedule): recorded into the caller's
    // buffer, or into the context's own
    int64_t own_n = 0;
    if (!replay) {
        c->last_sched.clear();
        if (!record) {
            c->own_rec.resize((size_t)NK2D_OWN_REC_CAP * NK2D_SCHED_WIDTH);
            record = c->own_rec.data();
            record_cap = NK2D_OWN_REC_CAP;
            record_n = &own_n;
        } else if (!record_n) {
            record_n = &own_n;
        }
    }
    Ctl s;
    s.c = c;
    s.t = c->d.t0;
    s.t1 = c->d.t1;
    s.max_step = (c->d.t1 - c->d.t0) * c->d.max_step_frac;
    s.n_total = (c->norm_hook && c->global_n > 0.0) ? c->global_n : (double)c->tc * c->nz * c->ny;
    s.newton_tol = std::max(10 * std::numeric_limits<double>::epsilon() / c->d.rtol, std::min(0.03, std::sqrt(c->d.rtol)));
    s.fingerprint = nk2d_fingerprint(c);
    s.has_old_h = s.has_old_err = false;
    s.h_abs_old = s.err_old = 0;
    s.have_lu = false; s.have_dense = false;
    s.h_lu = 0; s.dense_t_old = 0; s.dense_h = 0;
    s.m_real = s.m_cplx = 1;
    s.pre_setup = false;
    s.pre_jac = false;
    s.pre_t = s.pre_h = 0.0;
    // per-column partials go to pinned host memory while the host takes the decisions; the flag is
    // dropped on every way out of this function
    struct PartGuard {
        nk2d_ctx* c;
        ~PartGuard() { c->part_on_host = 0; }
    } part_guard{c};
    c->part_on_host = replay ? 0 : 1;
    // single-launch Newton iterations (one-sweep solves) swap stage buffers on the host: host-side decisions and
    // step replay only
    struct SwapGuard {
        nk2d_ctx* c;
        ~SwapGuard() { c->single_swap = 0; c->swap_updates = 0; c->part_cur = nullptr; }
    } swap_guard{c};
    c->single_swap = 1;
    c->swap_updates = (c->norm_hook_vec && c->ZS) ? 1 : 0;
    // A free-running year checks the convergence of every simplified Newton iteration on the iterates
    // themselves, inexact inner solves included.  A replayed schedule dictates the iteration counts of an
    // integrator with direct solves, so there the inner solves must not be what limits the accuracy.
    struct LinTolGuard {
        nk2d_ctx* c;
        double saved;
        ~LinTolGuard() { c->d.lin_tol = saved; }
    } lin_tol_guard{c, c->d.lin_tol};
    // (a schedule this library recorded itself under the same inner tolerance -- the frozen-controller year of a
    // finite-difference product -- repeats the recorded year's own solves: replay_own)
    if (replay && !replay_own) c->d.lin_tol = std::min(c->d.lin_tol, 1.0e-3);
    if (s.t1 > s.t) {
        // y = x; f = fun(t0, y0); first step size; J = jac(t0, y0)
        auto start_year = [&]() -> int {
            NK2D_CHECK(c, hipMemcpyAsync(c->Y, x, sizeof(double) * c->nv, hipMemcpyDeviceToDevice, nk2d_s(c)));
            NK2D_TRY(eval_kv(c, s.t, 3));
            NK2D_TRY(nk2d_k_tend(c, c->Y, c->KV[3], c->F));
            c->st.nfev++;
            if (!replay) NK2D_TRY(initial_step(s, &s.h_abs));
            NK2D_TRY(refresh_jac(s, s.t, true));
            c->st.njev = 1;
            s.current_jac = true;
            return 0;
        };
        NK2D_TRY(start_year());
        if (replay) {
            // the replayed year as a command stream (option "stream_years" bit 2; nk2d_stream.h): the launches of replay_rows
            // become commands, nothing is read back before the year's end
            struct StreamGuard {
                nk2d_ctx* c;
                ~StreamGuard() {
                    c->stream_on = 0;
                    if (c->strm && nk2d_stream_running(c)) (void)nk2d_stream_end(c);
                }
            } stream_guard{c};
            if ((c->stream_years & 2) && nk2d_stream_eligible(c) && c->hist_n == 0) {
                NK2D_TRY(nk2d_stream_ready(c));
                c->stream_on = 1;
            }
            const bool was_stream = c->stream_on != 0;
            // what a year books before it is known to stand: counters, algorithmic bytes and launch tallies
            const nk2d_stats st0 = c->st;
            const double bytes0 = c->fused_bytes_all;
            const int64_t launches0 = c->sweep_launches;
            int rrc = run_replay(s, replay, replay_n, replay_own);
            if (rrc == 3) {
                // the one-launch year of a small grid was given up: the same year from x, launch by launch.  What the
                // discarded year had booked is taken back (round-3 ADVICE: it was counted twice), the reason stays
                const int64_t timeouts = c->st.nbarrier_timeouts, resumed = c->st.nresumed;
                c->st = st0;
                c->st.nbarrier_timeouts = timeouts; c->st.nresumed = resumed;
                c->fused_bytes_all = bytes0;
                c->sweep_launches = launches0;
                s.no_persistent = true;
                s.t = c->d.t0;
                s.have_lu = false; s.have_dense = false; s.pre_setup = false;
                NK2D_TRY(start_year());
                rrc = run_replay(s, replay, replay_n, replay_own);
            }
            if (rrc != 0) return rrc;
            if (was_stream && c->stream_on) { c->stream_years_run++; c->stream_lost = 0; }
        }
        else {
            // The free-running year as a command stream (nk2d_stream.h): the launches of run_free become commands of ONE
            // resident kernel, the waits for events become waits for the partials themselves.  A kernel that gives up (a
            // wait over its time limit: a co-tenant on the chip, workgroups that did not all become resident) hands the year
            // back: the same year again from x, by launches -- counted, not failed.
            const bool as_stream = (c->stream_years & 1) && nk2d_stream_eligible(c) && c->part_on_host && c->speculate;
            struct StreamGuard {
                nk2d_ctx* c;
                ~StreamGuard() {
                    c->stream_on = 0;
                    if (c->strm && nk2d_stream_running(c)) (void)nk2d_stream_end(c);
                }
            } stream_guard{c};
            if (as_stream) NK2D_TRY(nk2d_stream_ready(c));
            c->stream_on = as_stream ? 1 : 0;
            int frc = run_free(s, record, record_cap, record_n);
            c->stream_on = 0;
            if (as_stream && (frc == 0 || frc == NK2D_RC_STREAM_LOST)) {
                const int erc = nk2d_stream_end(c);
                if (erc != 0 && erc != NK2D_RC_STREAM_LOST) return erc;
                if (erc == NK2D_RC_STREAM_LOST) frc = NK2D_RC_STREAM_LOST;
            }
            if (frc == NK2D_RC_STREAM_LOST) {
                if (++c->stream_lost >= 2) c->stream_years = 0;     // (not a third time on this context)
                nk2d_stream_part_forget(c, nullptr);
                c->st = nk2d_stats();
                c->st.nbarrier_timeouts = 1;
                s.t = c->d.t0;
                s.have_lu = false; s.have_dense = false; s.pre_setup = false;
                s.has_old_h = s.has_old_err = false;
                if (c->hist_n > 0) c->hist_next = 0;
                NK2D_TRY(start_year());
                frc = run_free(s, record, record_cap, record_n);
            } else if (frc == 0 && as_stream) {
                c->stream_years_run++;
                c->stream_lost = 0;
            }
            if (frc != 0) return frc;
        }
        NK2D_TRY(nk2d_r_final(c, (const double*)x, (double*)fx));
    } else {
        NK2D_CHECK(c, hipMemsetAsync(fx, 0, sizeof(double) * c->nv, nk2d_s(c)));
    }
    NK2D_CHECK(c, hipStreamSynchronize(nk2d_s(c)));
    NK2D_TRY(nk2d_profile_collect(c));
    if (!replay && record && record_n && *record_n <= record_cap)
        c->last_sched.assign(record, record + (size_t)(*record_n) * NK2D_SCHED_WIDTH);
    c->st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
    if (stats) *stats = c->st;
    return 0;
}
