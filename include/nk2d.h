/*
 * nk2d.h -- C ABI of libnk2d.so, the MI355X (gfx950) implementation of the
 * py_driver_2d Krylov / finite-difference-JVP hot path of
 * klindsay28/Newton-Krylov_OOC.
 *
 * Every entry point replaces one piece of the reference's Python plugin surface
 * (paths relative to the reference tree):
 *
 *   nk2d_create / nk2d_destroy      class-level set-up of the processes,
 *                                   nk_ooc/py_driver_2d/model_state.py:44-65
 *   nk2d_set_region                 region_mask / grid_weight / region mean matrix,
 *                                   nk_ooc/model_config.py:249-315
 *   nk2d_vec_*                      one tracer module's values (tracer, depth, ypos),
 *                                   nk_ooc/tracer_module_state_base.py:400-440
 *   nk2d_tend                       TracerModuleState.comp_tend + iage / forced / phosphorus.comp_tend,
 *                                   nk_ooc/py_driver_2d/tracer_module_state.py:98-108,
 *                                   py_driver_2d/iage.py:22-41, forced.py:114-154, phosphorus.py:58-89
 *   nk2d_vmix_coeff                 VertMix.mixing_coeff, py_driver_2d/vert_mix.py:44-101
 *   nk2d_jacobian_diags             comp_jacobian (five diagonals of the CSR matrix),
 *                                   py_driver_2d/tracer_module_state.py:262-270, iage.py:43-64
 *   nk2d_comp_fcn                   ModelState.comp_fcn's solve_ivp("Radau") year,
 *                                   nk_ooc/py_driver_2d/model_state.py:95-121
 *   nk2d_comp_fcn_hist              the same year with the 61-sample dense output of the history
 *                                   file, py_driver_2d/model_state.py:80-83,141-233
 *   nk2d_set_lin_state / nk2d_jacobian_apply
 *                                   comp_jacobian(time, tracer_vals) @ v for state dependent modules,
 *                                   py_driver_2d/phosphorus.py:105-172
 *   nk2d_precond_setup/apply        iage.apply_precond_jacobian, py_driver_2d/iage.py:66-93 (also
 *                                   forced.apply_precond_jacobian, forced.py:204-241)
 *   nk2d_precond_setup_states       the same with the tracer of the three time levels it reads from the
 *                                   precond file, forced.py:222-236 (file source with a sink threshold)
 *   nk2d_shift_factor/solve         sp_linalg.spsolve(mat - shift * mat_id, .) and the solves inside
 *                                   sp_linalg.eigs of phosphorus.apply_precond_jacobian,
 *                                   py_driver_2d/phosphorus.py:233-255
 *   nk2d_dot                        TracerModuleStateBase.dot_prod (weighted region mean),
 *                                   nk_ooc/tracer_module_state_base.py:379-388
 *   nk2d_axpby / nk2d_scale / nk2d_diff_scale
 *                                   state algebra with region-broadcast scalars,
 *                                   nk_ooc/tracer_module_state_base.py:200-369,502-515
 *   nk2d_lin_comb                   model_state_base.lin_comb, nk_ooc/model_state_base.py:619-624
 *   nk2d_mgs                        ModelStateBase.mod_gram_schmidt, model_state_base.py:365-377
 *   nk2d_apply_region_mask          TracerModuleStateBase.apply_region_mask, :494-500
 *
 * Conventions: plain C, no exceptions cross the boundary.  Functions return 0 on
 * success and a negative code on failure; nk2d_last_error(ctx) gives the message.
 * The caller owns every host buffer.  Host arrays are contiguous fp64, C order
 * (tracer, depth, ypos) for states and (depth, ypos) for planes.  A context owns
 * one HIP stream on one device; calls on one context are not thread safe,
 * different contexts may be driven concurrently (one per GPU / tracer module).
 * All arithmetic is IEEE fp64.
 */
#ifndef NK2D_H
#define NK2D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nk2d_ctx nk2d_ctx;
/* device-resident state vector of one tracer module (opaque device pointer) */
typedef void* nk2d_vec;

#define NK2D_MAX_TRACERS 4
#define NK2D_MAX_SHIFTS 4
/* doubles per accepted step of a schedule: t, t_new, h, n_newton, t_jac, h_lu, err, fingerprint.
   err: the error-estimate norm the step was accepted with (0 in a schedule from elsewhere);
   fingerprint: nk2d_schedule_fingerprint of the context that recorded the step (0: a schedule from elsewhere, e.g.
   SciPy's steps for the step-replay parity tests -- never checked) */
#define NK2D_SCHED_WIDTH 8

typedef struct nk2d_desc {
    int32_t nz;            /* depth levels */
    int32_t ny;            /* ypos levels */
    int32_t tc;            /* tracers in the module (iage: 2) */
    int32_t device_id;     /* HIP device ordinal */
    const double* depth_edges; /* [nz+1] */
    const double* ypos_edges;  /* [ny+1] */
    const double* vvel;        /* [nz][ny+1]  Advection.vvel */
    const double* wvel;        /* [nz+1][ny]  Advection.wvel */
    const double* hmix_coeff;  /* [nz][ny-1]  HorizMix._mixing_coeff (includes 1/dy) */
    const double* bldepth_max; /* [ny]  VertMix.bldepth's bldepth_max profile */
    double bldepth_min;        /* 35 m */
    double bld_tvals[4];       /* knots of the seasonal fraction, seconds */
    double bld_fvals[4];       /* values at the knots (0,1,1,0) */
    double vmix_log_shallow;   /* ln(10)    */
    double vmix_log_deep;      /* ln(5e-4)  */
    double vmix_half_width;    /* 20 m      */
    /* module sources (iage.py:31-39, forced.py:114-139):
       tend[tr][0][:] += surf_rate[tr]*(surf_target[tr] - c[tr][0][:]);
       tend -= decay_rate[tr]*c everywhere; tend += const_src */
    double surf_rate[NK2D_MAX_TRACERS];
    double surf_target[NK2D_MAX_TRACERS];
    double decay_rate[NK2D_MAX_TRACERS];
    double const_src;
    double t0, t1;             /* time_range, seconds */
    double rtol, atol;         /* Radau tolerances (1e-6, 1e-6) */
    double max_step_frac;      /* max_step = frac*(t1-t0) (0.01) */
    double lin_tol;            /* target relative accuracy of the inner line-relaxation solves */
    /* module kind: 0 = linear sources above (iage, forced); 2 = forced with forcing files (below);
       1 = phosphorus (po4, dop, pop;
       nk_ooc/py_driver_2d/phosphorus.py:17-172), then tc = 3 and
       phos_params = {po4_halfsat, max_uptake_rate, sigma, dop_remin_rate, pop_remin_rate,
       pop_sink_vel} (phosphorus.py:42-58), light_lim [nz][ny] (phosphorus.py:26-29) */
    int32_t module_kind;
    int32_t reserved0;
    double phos_params[6];
    const double* light_lim;
    /* module kind 2 = forced module with file-driven forcing (nk_ooc/py_driver_2d/forced.py:42-56,
       125-153,188-202): tc = 1.  The records are the file's fields already interpolated to the
       model axes (nk_ooc/utils.py:488-533); the library interpolates them linearly in time with
       linear extrapolation beyond the first / last record, as scipy's interp1d(fill_value=
       "extrapolate") does.  restore_nrec > 0: tend[0][0][:] += surf_rate[0]*(restore(t) - c[0][:])
       with restore(t) from the records instead of surf_target[0].  sms_nrec > 0: tend += sms(t),
       and with sink_thres > 0 sms is multiplied by c/sink_thres where sms < 0 and 0 < c/sink_thres < 1
       (then the Jacobian has the state-dependent diagonal of forced.py:188-202).  decay_rate[0] and
       const_src keep their meaning (forced_sms_opt = decay / const). */
    int32_t restore_nrec;
    int32_t sms_nrec;
    const double* restore_times; /* [restore_nrec], seconds, increasing */
    const double* restore_vals;  /* [restore_nrec][ny] */
    const double* sms_times;     /* [sms_nrec] */
    const double* sms_vals;      /* [sms_nrec][nz][ny] */
    double sink_thres;           /* 0: no threshold */
} nk2d_desc;

typedef struct nk2d_stats {
    int64_t nfev, njev, nlu;       /* SciPy-compatible counters */
    int64_t nsteps, nrejected;     /* accepted steps, rejected attempts */
    int64_t nnewton, nsolve;       /* Newton iterations, linear solves */
    int64_t nsweeps;               /* line-relaxation sweeps (kernel launches) */
    int64_t nlaunch;               /* total kernel launches */
    double seconds;                /* host wall time of the call */
    int64_t nresumed;              /* frozen year: times it was resumed from a checkpoint with one more Newton iteration */
    int64_t nerr_checked;          /* frozen year: steps whose error estimate was evaluated (option "frozen_err_check") */
    double max_err;                /* ... and the largest of them (SciPy accepts a step at <= 1) */
    int64_t nbarrier_timeouts;     /* 1: the one-launch year timed out at a grid barrier and was rerun under host control */
} nk2d_stats;

int nk2d_create(const nk2d_desc* desc, nk2d_ctx** out);
void nk2d_destroy(nk2d_ctx* ctx);
const char* nk2d_last_error(const nk2d_ctx* ctx);
const char* nk2d_version(void);

/* region_mask [nz][ny] (int32, 0 = outside), grid_weight [nz][ny]; nreg = max(mask) */
int nk2d_set_region(nk2d_ctx* ctx, const int32_t* mask, const double* weight, int32_t nreg);

int nk2d_vec_alloc(nk2d_ctx* ctx, nk2d_vec* out);
int nk2d_vec_free(nk2d_ctx* ctx, nk2d_vec v);
int nk2d_vec_upload(nk2d_ctx* ctx, nk2d_vec v, const double* host);   /* [tc][nz][ny] */
int nk2d_vec_download(nk2d_ctx* ctx, nk2d_vec v, double* host);
/* The same in two halves, for a caller that writes vector files on a thread of its own (what the reference does inside
 * ModelStateBase.dump, nk_ooc/model_state_base.py:93-111, here behind the next forward year): `begin` queues the copy
 * into a pinned buffer of the download's own on the context's stream and returns at once (later work on the stream,
 * including changes of v, is ordered behind it); `end` -- on any host thread -- waits for that copy alone and fills
 * host [tc][nz][ny] (NULL: only releases the ticket).  Every ticket must be ended once, before nk2d_destroy. */
int nk2d_vec_download_begin(nk2d_ctx* ctx, nk2d_vec v, void** ticket);
int nk2d_vec_download_end(nk2d_ctx* ctx, void* ticket, double* host);
int nk2d_vec_copy(nk2d_ctx* ctx, nk2d_vec dst, nk2d_vec src);
int nk2d_vec_zero(nk2d_ctx* ctx, nk2d_vec v);

/* deterministic kernels */
int nk2d_tend(nk2d_ctx* ctx, double t, nk2d_vec y, nk2d_vec f);
int nk2d_vmix_coeff(nk2d_ctx* ctx, double t, double* host_out /* [nz-1][ny] */);
/* state the stand-alone Jacobian entry points below linearise about (only state dependent
   modules, i.e. phosphorus, use it); nk2d_comp_fcn linearises about its own running state */
int nk2d_set_lin_state(nk2d_ctx* ctx, nk2d_vec y);
/* out = J(t, lin_state) v : tracer_module.comp_jacobian(...) @ v,
   py_driver_2d/tracer_module_state.py:262-270, phosphorus.py:105-172 */
int nk2d_jacobian_apply(nk2d_ctx* ctx, double t, nk2d_vec v, nk2d_vec out);
/* diags: [5][tc][nz][ny] in the order up(k-1), south(j-1), centre, north(j+1), down(k+1) */
int nk2d_jacobian_diags(nk2d_ctx* ctx, double t, double* host_out);
/* solve ((mu/h) I - J(t_jac)) x = b by line relaxation; complex when mu_im != 0.
   b_re/b_im/x_re/x_im are device vectors (b_im, x_im ignored for real systems). */
int nk2d_shifted_solve(nk2d_ctx* ctx, double t_jac, double h, double mu_re, double mu_im,
                       nk2d_vec b_re, nk2d_vec b_im, nk2d_vec x_re, nk2d_vec x_im,
                       int32_t* sweeps_out);

/* one forward year: fx = y(t1) - x, region-masked.
   replay: optional schedule [replay_n][NK2D_SCHED_WIDTH] to consume (step-replay mode; the inner
           tolerance is then min(lin_tol, 1e-3): the schedule fixes the Newton iteration counts);
   record: optional buffer [record_cap][NK2D_SCHED_WIDTH] receiving the accepted steps. */
int nk2d_comp_fcn(nk2d_ctx* ctx, nk2d_vec x, nk2d_vec fx, nk2d_stats* stats,
                  const double* replay, int64_t replay_n,
                  double* record, int64_t record_cap, int64_t* record_n);
/* The finite-difference product of the reference, (F(x + sigma v) - F(x)) / sigma with sigma = 1e-4 |x|
   (model_state_base.py:492-527), differences two years of an ADAPTIVE integrator: wherever the two controllers
   decide differently (a rejected step, another Newton iteration) the difference of their discretisation errors,
   divided by sigma, lands in the product -- measured here at 5 % ... 90 % of |w| for the first Krylov direction, in
   every controller mode including SciPy's own (tools/probe_jvp_noise.py).  Internal numerical differentiation removes
   it: the perturbed year repeats the accepted steps, Newton iteration counts, Jacobian times and factorisations of
   the year that produced F(x), which makes w the derivative of ONE discrete map.
   nk2d_comp_fcn_frozen: forward year on a schedule recorded by nk2d_comp_fcn on this context under the same options,
   with the recorded year's own inner tolerance, nothing decided and nothing read back (for the recorded x itself it
   reproduces the recorded year bit for bit).
   nk2d_set_frozen_schedule: the schedule the perturbed years of nk2d_jvp / nk2d_gmres_solve repeat from now on
   (copied; sched_n = 0 returns them to free-running years). */
int nk2d_comp_fcn_frozen(nk2d_ctx* ctx, nk2d_vec x, nk2d_vec fx, nk2d_stats* stats, const double* sched, int64_t sched_n);
int nk2d_set_frozen_schedule(nk2d_ctx* ctx, const double* sched, int64_t sched_n);
/* A frozen year checks afterwards, for every step, SciPy's Newton convergence test on the last recorded iteration
   (with a slack of 30) -- the norm partials of those iterations are kept, one reduction launch and one read-back per year.
   Where the recorded count is not enough for the state given, the year is RESUMED from the checkpoint before the first
   such step (the state is kept every 128 steps) with one more Newton iteration there, at most twice; a year that still
   fails -- or a step already at SciPy's six iterations -- makes nk2d_comp_fcn_frozen return -7, and nk2d_jvp then runs a
   free-running year instead.  Option "frozen_err_check" k > 0 (default 128: the steps that open a checkpoint interval) also evaluates SciPy's
   error estimate on every k-th step of a frozen year (one launch each, plus the tendency at the step start): a step whose estimate
   exceeds 1.5 * max(1, recorded estimate) returns -7 as well.  A schedule whose fingerprint is not this context's
   (other options, grid or library build) is refused with -8 before anything runs.
   nk2d_frozen_fallbacks: how often a frozen year was given up (-7) on this context; nk2d_frozen_resumes: how often one was
   resumed. */
int nk2d_frozen_fallbacks(nk2d_ctx* ctx, int64_t* n);
int nk2d_frozen_resumes(nk2d_ctx* ctx, int64_t* n);
/* Modules whose Jacobian is a function of time alone, grids of up to 512 levels (option "frozen_persistent_max_e", levels per
   lane, default 8; five to eight only with linear sources): a frozen year runs as ONE launch whose workgroups are all resident -- a wave or a
   four-wave team per column, one simplified-Newton iteration per phase, workgroups handing over to their lateral neighbours
   between phases -- on a cache of everything its schedule fixes besides the state: mixing planes, Jacobian planes and line
   factorisation of every step, built by two batched launches when a new schedule arrives (options "frozen_persistent" 0/1,
   "frozen_cache_gb": at most that much HBM, default 128, and never more than 85 % of what the device has free; 102 GB and 26 ms
   per schedule at 416 x 416, where the year takes 115 - 120 ms instead of 190 ms; "frozen_cache_after": frozen years of a schedule
   that run launch by launch before its cache is built, default 0; "frozen_alloc_async" 1, default: a cache above 8 GB is
   allocated by a thread of the library's own -- hipMalloc of 120 GB takes 0.03 to 3 s -- and the years of the meantime run
   launch by launch).  Same device functions, bit-identical results; a year that does not pass the Newton check, or a
   barrier that times out, is handed to the launch-per-phase path.  Counters by name: "frozen_persistent_years",
   "frozen_team_years" (of them: a four-wave team per column), "frozen_cache_bytes",
   "frozen_launch_us" (device time of those launches), "frozen_cache_pending" (1 while a thread allocates a large cache),
   "frozen_cache_builds", "frozen_fallbacks", "frozen_resumes"; of the host-side controller: "spec_launches_dropped",
   "spec_front_launches_dropped", "err_estimates_queued", "err_estimates_dropped" (work queued ahead of a verdict). */
int nk2d_get_counter(nk2d_ctx* ctx, const char* name, int64_t* out);
/* hash of everything a recorded schedule depends on besides the state: grid, module description, tolerances, the
   controller options (jac_fresh, jac_stage, lin_tol, min_sweeps, growth_cap, factor storage), the library version and a
   checksum of its sources (another build's schedules are refused);
   an integer below 2^52, never 0 */
int nk2d_schedule_fingerprint(nk2d_ctx* ctx, double* out);
/* accepted steps of the most recent free-running year of this context (whichever entry point ran it: nk2d_comp_fcn,
   nk2d_comp_fcn_hist, the perturbed year of a free-running nk2d_jvp): n rows of NK2D_SCHED_WIDTH doubles; out may be
   NULL to ask for n only */
int nk2d_last_schedule(nk2d_ctx* ctx, double* out, int64_t cap, int64_t* n);

/* the same forward year with dense output: the solution at the n_eval increasing times t_eval
   (within [t0, t1]) is written to host_hist [n_eval][tc][nz][ny], evaluated from each step's
   collocation polynomial as solve_ivp(t_eval=...) does (the 61-sample history of
   py_driver_2d/model_state.py:80-83) */
int nk2d_comp_fcn_hist(nk2d_ctx* ctx, nk2d_vec x, nk2d_vec fx, nk2d_stats* stats, int32_t n_eval,
                       const double* t_eval, double* host_hist);

/* preconditioner  M^-1 v = (I - prod_k (I - dt J(t_k)))^-1 v - v */
int nk2d_precond_setup(nk2d_ctx* ctx);
/* the same for a forced module whose Jacobian depends on the state (file source with sink_thres):
   states[i] = the tracer at the end of the i-th third of [t0, t1], as forced.apply_precond_jacobian
   reads it from the preconditioner file (py_driver_2d/forced.py:222-236) */
int nk2d_precond_setup_states(nk2d_ctx* ctx, const nk2d_vec* states);
int nk2d_precond_apply(nk2d_ctx* ctx, nk2d_vec v, nk2d_vec out);
/* shifted systems of the phosphorus preconditioner (phosphorus.py:233-255): factorise
   A_i = scale * J(t, lin_state) - shifts[i] * I (all tracers of the module coupled, nshift <=
   NK2D_MAX_SHIFTS) by block elimination over the ypos columns, replacing the reference's
   sp_linalg.spsolve(mat - shift * mat_id, .) and the solves inside sp_linalg.eigs(sigma=...);
   nk2d_shift_solve: out = A_i^-1 v.  Shares its storage with nk2d_precond_setup. */
int nk2d_shift_factor(nk2d_ctx* ctx, double t, double scale, int32_t nshift, const double* shifts);
int nk2d_shift_solve(nk2d_ctx* ctx, int32_t i, nk2d_vec v, nk2d_vec out);

/* state algebra; region scalars are host arrays [nreg], broadcast with fill 1.0 where mask<=0 */
int nk2d_dot(nk2d_ctx* ctx, nk2d_vec a, nk2d_vec b, double* out /* [nreg] */);
int nk2d_axpby(nk2d_ctx* ctx, nk2d_vec out, const double* a, nk2d_vec x, const double* b, nk2d_vec y);
int nk2d_scale(nk2d_ctx* ctx, nk2d_vec out, nk2d_vec x, const double* s);
int nk2d_diff_scale(nk2d_ctx* ctx, nk2d_vec out, nk2d_vec x, nk2d_vec y, const double* s);
int nk2d_lin_comb(nk2d_ctx* ctx, nk2d_vec out, int32_t n, const nk2d_vec* vecs,
                  const double* coef /* [n][nreg] */);
/* in-place modified Gram-Schmidt of w against n basis vectors; h_out [n][nreg] */
int nk2d_mgs(nk2d_ctx* ctx, nk2d_vec w, int32_t n, const nk2d_vec* basis, double* h_out);
int nk2d_apply_region_mask(nk2d_ctx* ctx, nk2d_vec v);

/* Measurement plumbing (bench.py): time every every_n-th WINDOW of the dominant kernel
   (k_newton_fused) with a HIP event pair on the context's stream.  A window is the launches of one
   simplified-Newton iteration that are queued back to back with nothing between them (first
   iteration of a step attempt).  every_n = 0 switches sampling off. */
int nk2d_profile_reset(nk2d_ctx* ctx, int32_t every_n);
/* avg_us: (window times - one empty-pair reading per window) / launches inside the windows;
   samples: those launches; windows: timed windows; bytes: algorithmic bytes of the launches in
   the windows; launches: all launches of the kernel since the reset; overhead_us: what an EMPTY
   event pair reads on this stream */
int nk2d_profile_read(nk2d_ctx* ctx, double* avg_us, int64_t* samples, int64_t* launches, double* bytes,
                      double* overhead_us, int64_t* windows);

/* the same counters since the last nk2d_profile_reset over ALL launches of the dominant kernel (timed
   or not): their number and their algorithmic bytes -- for the end-to-end rate bytes / wall time */
int nk2d_profile_totals(nk2d_ctx* ctx, int64_t* launches, double* bytes);
/* the launches of the dominant kernel without the factorisation since the last nk2d_profile_reset, by shape
   (0: stage + sweep + update, 1: stage + first sweep, 2: last sweep + update, 3: a middle sweep), and their
   algorithmic bytes */
int nk2d_profile_shapes(nk2d_ctx* ctx, int64_t* counts4, double* bytes4);
/* n launches of shape 0, 1 or 2 of that kernel queued back to back inside ONE event pair, on the state the last
   forward year left behind, their updates written to scratch: microseconds per launch (the hand-over between
   consecutive launches included, no event cost to subtract) and the algorithmic bytes of one such launch */
int nk2d_profile_replay(nk2d_ctx* ctx, int32_t shape, int32_t n, double* avg_us, double* bytes_per_launch);
/* a HIP event pair on the context's own stream for the caller's measurements (bench.py times the
   preconditioner apply with it): nk2d_timer_begin records the first event, nk2d_timer_end records
   the second, waits for it and returns the elapsed milliseconds between the two */
int nk2d_timer_begin(nk2d_ctx* ctx);
int nk2d_timer_end(nk2d_ctx* ctx, double* elapsed_ms);

/* ---- the Krylov loop itself on the device ------------------------------

   nk2d_jvp: ModelStateBase.comp_jacobian_fcn_state_prod for one tracer module,
   nk_ooc/model_state_base.py:492-527, in one call: sigma[r] = 1e-4 * norm(x)[r] (1 where that is 0),
   w = (F(x + sigma v) - fx) * (1 / sigma), region by region.  fx = F(x) is the caller's (the Newton
   solver has it).  perturb_fcn (optional, may be null) receives F(x + sigma v), the vector the
   reference dumps as perturb_fcn_w_raw_NN.nc; sigma_out (optional) [nreg]; stats (optional) of the
   perturbed forward year.  The perturbed year is a free-running one, as the reference's, unless a schedule is
   installed (nk2d_set_frozen_schedule): then it repeats the accepted steps of the year that produced fx, and a
   state the recorded Newton counts do not converge for gets a free-running year after all (nk2d_frozen_fallbacks). */
int nk2d_jvp(nk2d_ctx* ctx, nk2d_vec x, nk2d_vec fx, nk2d_vec v, nk2d_vec w, nk2d_vec perturb_fcn,
             double* sigma_out, nk2d_stats* stats);

/* nk2d_gmres_solve: KrylovSolver.solve + _solve0 for ONE tracer module, nk_ooc/krylov_solver.py:85-165:
   left-preconditioned GMRES (Saad alg. 9.4), zero initial guess, no restart, one Hessenberg per
   region, coefficients = argmin || beta e_1 - H c ||_2 (the reference calls np.linalg.lstsq,
   krylov_solver.py:168-181; here a Givens QR of the Hessenberg -- the same minimiser), stop when
   iteration >= min_iter and || sum_j c_j w_j + M^-1 fx || < rel_tol * beta in every region, or at
   max_iter.  Everything stays in HBM; nothing is written to disk (the Python mirror of KrylovSolver
   keeps the reference's file trail).  The preconditioner is nk2d_precond_apply: call
   nk2d_precond_setup (or _setup_states) first; the phosphorus module, whose preconditioner is
   assembled above this boundary, is refused.
   Outputs (caller-owned host buffers, zero padded): beta [nreg]; h_mat [max_iter+1][max_iter][nreg]
   (the reference's h_mat[module] with the iteration axes first); resid_norm [max_iter][nreg];
   coeff [max_iter][nreg] of the last iteration; *iters = iterations done; increment = the device
   vector the reference dumps as increment_NN.nc.  Any of beta / h_mat / resid_norm / coeff may be null. */
int nk2d_gmres_solve(nk2d_ctx* ctx, nk2d_vec x, nk2d_vec fx, double rel_tol, int32_t min_iter,
                     int32_t max_iter, nk2d_vec increment, double* beta, double* h_mat,
                     double* resid_norm, double* coeff, int32_t* iters);

/* all n region-weighted dot products <w, basis[i]> in one launch and one read-back (the fused
   multi-dot of SURVEY.md section 8(e)); out [n][nreg] */
int nk2d_multi_dot(nk2d_ctx* ctx, nk2d_vec w, int32_t n, const nk2d_vec* basis, double* out);
/* w -= sum_i bcast(h[i]) basis[i] in one launch; h [n][nreg] (classical Gram-Schmidt update).  fill: the
   broadcast value where region_mask <= 0 -- 1.0 is the reference's (every modified Gram-Schmidt projection
   subtracts basis[i] itself there); a re-orthogonalisation pass uses 0.0 so as not to subtract it twice */
int nk2d_multi_axpy(nk2d_ctx* ctx, nk2d_vec w, int32_t n, const nk2d_vec* basis, const double* h, double fill);

/* Tracers of one module sharded over several contexts / GPUs (level 2): the
   tracers' Jacobian blocks are independent, but SciPy's Radau takes its decisions from norms over the
   whole module (radau.py:118,481; common.py:63-65).  With a hook installed, every sum of squares the
   controller reads is passed through `fn` (the caller makes it an all-reduce over the shards: RCCL on
   GPUs, gloo in the CPU tests) and the RMS norms divide by global_n = the module's tc * nz * ny, so
   that all shards take identical decisions.  fn == NULL removes the hook.  (A hooked context's years are launched: the
   controller waits for the caller's all-reduces, which no resident kernel should sit through -- option "stream_years".) */
typedef double (*nk2d_norm_hook_fn)(void* user, double local_sum_of_squares);
int nk2d_set_norm_hook(nk2d_ctx* ctx, nk2d_norm_hook_fn fn, void* user, double global_n);
/* The same with several sums per call (1 <= n <= 4; replaced in place by the module-wide sums): the controller then
   BATCHES what it can -- the norm of a Newton iteration with those of up to two iterations queued speculatively behind it
   (option "hook_spec_depth", default 2; 1 = one) and with the error estimate queued behind the iteration expected to be the
   last -- and a coupled year takes well under half the collectives of the scalar hook (measured: tools/probe_pairing.py,
   bench.py's shard_e2 leg); the decisions and their order are the scalar hook's, the schedules identical.  Unwanted
   iterations and estimates are dropped.  A context has one hook: installing either kind removes the other. */
typedef void (*nk2d_norm_hook_vec_fn)(void* user, double* sums, int32_t n);
int nk2d_set_norm_hook_vec(nk2d_ctx* ctx, nk2d_norm_hook_vec_fn fn, void* user, double global_n);

/* run-time options.  ONE set of defaults since round 3: a context is created in the mode the engines, the tests of the
   benchmarked path and bench.py run -- "jac_fresh" 1, "jac_stage" 1 (and desc.lin_tol = 3e-2 is what they pass);
   SciPy's decisions step for step are "jac_fresh" 0 + "jac_stage" -1 (what the counter-parity tests set).
   "lin_tol" (relative accuracy of the inner line-relaxation solves),
   "stream_years" (bit 1, default: free-running forward years run as COMMAND STREAMS -- one resident kernel executes the
   launches of the host-controlled year as commands pushed into a ring in HBM, workgroups hand over to their two lateral
   neighbours between commands, norm partials come back through pinned host memory; same controller, same device functions,
   the same year bit for bit, csrc/nk2d_stream.h.  Bit 2: frozen / replayed years too -- measured slower than the one-launch
   year on the schedule cache for the modules that has one, equal to launches otherwise; 0: every year by launches.
   Counters "stream_years_run", "stream_commands", "stream_launches" (kernel starts), "stream_timeouts" (kernels that gave
   up: the year is rerun by launches), "stream_prof_0..11" (microseconds per workgroup waiting for commands / executing /
   waiting for neighbours, commands, and per kind of command).  NK2D_STREAM_RELAY=1 in the environment: commands through
   pinned host memory and a relay wave instead of written over the large BAR),
   "stream_two_waves" (default 1; before the context's first year: phosphorus from five levels per lane takes the flavour of
   the resident kernel that fits two waves to a SIMD -- 256 registers, the rest in scratch memory -- where that lets every
   (tracer, ypos) column be resident instead of several rounds of columns per command, 416 x 416: 1 248 columns; 2: wherever
   that flavour exists; 0: never.  Counters "stream_two_waves_kernel", "stream_columns_per_workgroup"),
   "device_ctl" (only 0: rounds 1 - 3 had device-side controllers 1, 2, 3; they lost to the command stream and are gone),
   "spec_bias" (default 1: what the host queues behind a Newton iteration it has not judged yet -- the next iteration, or the
   error estimate when the predicted convergence test value is below this many tolerances; never a decision; measured flat),
   "jac_fresh" (1: re-evaluate the
   Jacobian at every step start instead of SciPy's reuse heuristic; same ODE, same error
   control, different -- shorter -- sequence of Newton iterations), "growth_cap" (> 0: largest
   growth factor of the step size after a step whose Newton iteration failed and was repeated with
   half the step; 1.0 is the rule of Hairer & Wanner's RADAU5, 0 = SciPy, which has none),
   "min_sweeps" (1, default: a solve whose contraction bound meets lin_tol after ONE sweep runs the whole simplified
   Newton iteration as a single launch, its update written to a spare stage buffer; 2: at least two sweeps wherever
   columns couple, the round-1 rule),
   "final_fuse" (1, default: a step of a frozen year, nk2d_comp_fcn_frozen, ends in the launch of its last Newton
   iteration; 0: in a step boundary launch of its own, for A/B runs),
   "jac_stage" (0, 1 or 2, with "jac_fresh" 1: the Jacobian of a step attempt is
   taken at the time of that stage of the attempt, t + c_i h, instead of the step start -- the simplified Newton
   iteration and the error filter use ONE Jacobian for the three stages and the vertical mixing changes over a step;
   the launch that computes the stage's mixing plane derives the Jacobian planes from it.  1 is the default; -1:
   the step start, as SciPy.  Modules whose Jacobian also reads the state take the mixing plane of that time and the
   state of the step start, in a launch of their own -- option "jac_stage_state" 1, the default; 0: step start for both),
   "team" (launch shape of the Newton-iteration launches: 0 one wave per column; 1 one workgroup of four waves per
   column -- the three stages and the complex system on waves of their own; 2 a pair of waves per column -- stages and
   real system on one, complex system on the other; same arguments, bit-identical results; -1, the default: chosen per
   context where all waves fit the chip at once -- teams up to 128 columns, pairs up to 512, nk2d_team_auto),
   "pc_valu" (1: the round-1 preconditioner kernels, for A/B runs), "pc_fused" (1, default: from blocks of 1024 rows a panel
   step of the preconditioner's Gauss-Jordan inversions is ONE launch whose first workgroup also inverts the NEXT step's
   pivot block; 2: at every size; 0: the two launches of rounds 1 - 3 -- the same bits either way), "sweep_wpb",
   (closed experiments removed in round 4: "xcd_map", "prefactor" -- profiles/r03_prefactor holds their measurements),
   "frozen_err_check" (k: SciPy's error estimate on every k-th step of a frozen year, default 128; 0 off),
   "frozen_persistent" / "frozen_persistent_max_e" / "frozen_cache_gb" (the one-launch frozen year, see nk2d_get_counter),
   "frozen_team" (1, default: a four-wave team per column inside that launch up to two levels per lane; 0: a wave per
   column), "frozen_wpb" (columns per workgroup of the wave-per-column
   flavour with that hand-over, 1 .. 4, default 2: the waves of a workgroup move in lock step, a neighbour in another
   workgroup is read over the fabric), "frozen_coef_lds" (bits, default 15: what a wave of that flavour keeps in LDS for the
   year at three and more levels per lane -- 1 the static coefficients of its column, 2 W, 4 the step's mixing columns and
   vertical Jacobian diagonals, 8 the real system's pivots; bits 4 and 8 need "frozen_by_column"), "frozen_by_column" (1,
   default: from five levels per lane a workgroup is ONE ypos column with all its tracers, so that what is the same for every
   tracer of a column is shared through LDS; 2: from three levels per lane; 0: adjacent columns of one tracer),
   "hook_spec_depth" (1 or 2, default 2: whole Newton iterations a controller with a vector norm hook queues ahead of a verdict), "barrier_timeout_ms" (longest wait at a grid barrier of the one-launch years, default 2000:
   then the year is rerun launch by launch), "year_fences" (1: release / acquire fences around those barriers, validation),
   "pc_fp32" (1: the preconditioner's Schur inverses stored in single precision -- half the HBM -- and every apply refined
   "pc_refine" times, default 1, against the exact block tridiagonal operator; set before nk2d_precond_setup) */
int nk2d_set_option(nk2d_ctx* ctx, const char* name, double value);

/* block until every operation queued on the context's stream has finished */
int nk2d_sync(nk2d_ctx* ctx);
/* the context's HIP stream (hipStream_t as void*), for event timing by the caller */
void* nk2d_stream(nk2d_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* NK2D_H */
