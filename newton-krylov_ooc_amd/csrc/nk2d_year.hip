// nk2d_year.hip -- the free-running forward year of a small grid in ONE launch (option "device_ctl" 3).
#include "nk2d_bodies.h"

template <int E, int KIND>
__global__ void __launch_bounds__(NK2D_BLOCK) k_year_persistent(DevP P, YearArgs A) {
    __shared__ int lds_ok;
    const int lane = threadIdx.x & 63;
    const int wave = uni_i((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    const bool col_wave = wave < P.ncol;             // this wave owns column `wave` for the whole year
    GridBarrier bar{A.arrive, A.abort_flag, gridDim.x, 0u, &lds_ok, A.spin_ticks, A.fences};
    const double RC0 = 0.15505102572168222, RC1 = 0.6449489742783178, RC2 = 1.0;
    const double MU_REAL = 3.637834252744496, MU_CR = 2.6810828736277523, MU_CI = -3.050430199247411;
    const int NEWTON_MAXITER = 6;
    const size_t nv = (size_t)P.ncol * (E * 64);

    // buffers that swap roles: parities, the pointers are selected where they are used
    // the three stage buffers rotate: current iterate, previous step's (dense output), spare (single-launch iterations)
    int swapY = 0, swapKV = 0, zc = 0, zp = 1, zn = 2;
#define YR_ZBUF(i) (((i) == 0) ? A.Z : (((i) == 1) ? A.ZP : A.ZN))
#define YR_Y (swapY ? A.YOLD : A.Y)
#define YR_YOLD (swapY ? A.Y : A.YOLD)
#define YR_Z YR_ZBUF(zc)
#define YR_ZP YR_ZBUF(zp)
#define YR_ZN YR_ZBUF(zn)
#define YR_KV2 (swapKV ? A.KV[3] : A.KV[2])
#define YR_KV3 (swapKV ? A.KV[2] : A.KV[3])
    // controller state (identical in every wave)
    double t = A.t0, h_abs_s = A.h_abs0, h_abs_old_s = 0.0, err_old_s = 0.0;
    bool has_old_h = false, has_old_err = false, current_jac = true, have_lu = false, have_dense = false;
    double h_lu = 0.0, t_jac = A.t0, dense_t_old = 0.0, dense_h = 0.0;
    int m_real = 1, m_cplx = 1;
    bool factor_pending = false;
    bool pre_setup = false;      // the next attempt's planes and predicted stage values came with the commit phase
    double pre_h = 0.0;
    double lu_cre = 0.0, lu_ccr = 0.0, lu_cci = 0.0;
    int nfev = 0, njev = 0, nlu = 0, nsteps = 0, nrejected = 0, nnewton = 0, nsolve = 0, nsweeps = 0, nrec = 0;
    double words = 0.0;
    int status = 0;     // 0 ok, 1 barrier timeout, 2 non-finite step, 3 step too small
    // Norm partials alternate between two buffers: the waves still summing reduction n must not see the
    // partials of reduction n + 1, which a faster wave may already write (a single phase can lie between them)
    unsigned pev = 0;
#define YEAR_PART() (A.PART + (size_t)(pev & 1u) * P.ncol)
#define YEAR_SYNC() \
    if (!bar.sync()) { status = 1; goto finish; }
    const double Pc = (double)P.nz * P.ny, Ntot = Pc * P.tc;

    // Jacobian planes from the vertical mixing plane kv (tasks spread over all waves)
#define YEAR_JAC(kvp)                                                                                                   \
    for (int task = wave; task < P.ny; task += nwaves)                                                                  \
        jac_body<E, 1>(P, kvp, const_cast<double*>(A.fac.JL), const_cast<double*>(A.fac.JU), const_cast<double*>(A.fac.JS), \
                       const_cast<double*>(A.fac.JN), const_cast<double*>(A.fac.JC), nullptr, nullptr, task, lane);

    // one (stage time, ypos column) task of an attempt's planes; with option "jac_stage" the wave that computes the
    // column of that stage derives the Jacobian planes of the column from it
    const bool jac_at_stage = A.jac_stage >= 0;
    const double rc_jac = (A.jac_stage == 0) ? RC0 : ((A.jac_stage == 1) ? RC1 : RC2);
#define YEAR_PLANE(ti, fr, dst, j)                                                                                     \
    {                                                                                                                   \
        double kvc_[E];                                                                                                 \
        vmix_col_regs<E>(P, A.bldmin, A.vy0, A.vy1, A.hw, fr, j, lane, kvc_);                                           \
        store_col<E, 1>(dst, j, lane, kvc_);                                                                            \
        if ((ti) == A.jac_stage)                                                                                        \
            jac_core<E, 1>(P, kvc_, dst, const_cast<double*>(A.fac.JL), const_cast<double*>(A.fac.JU),                  \
                           const_cast<double*>(A.fac.JS), const_cast<double*>(A.fac.JN), const_cast<double*>(A.fac.JC), \
                           nullptr, nullptr, j, lane);                                                                  \
    }

    while (t < A.t1) {
        const double min_step = 10.0 * fabs(nextafter(t, INFINITY) - t);
        double h_abs, h_abs_old = 0.0, err_old = 0.0;
        bool has_h_old, has_err_old;
        if (h_abs_s > A.max_step) { h_abs = A.max_step; has_h_old = has_err_old = false; }
        else if (h_abs_s < min_step) { h_abs = min_step; has_h_old = has_err_old = false; }
        else { h_abs = h_abs_s; h_abs_old = h_abs_old_s; err_old = err_old_s; has_h_old = has_old_h; has_err_old = has_old_err; }
        if (A.jac_fresh && !current_jac && !jac_at_stage) {
            YEAR_JAC(YR_KV3)     // KV3 holds the plane at the current t
            YEAR_SYNC()
            t_jac = t; ++njev; current_jac = true; have_lu = false;
        }
        bool rejected = false, accepted = false, newton_failed = false;
        double h = 0.0, t_new = 0.0, err = 0.0, safety = 0.0, rate = 0.0;
        bool have_rate = false;
        int n_iter = 0;
        while (!accepted) {
            if (uni_b(!isfinite(h_abs))) { status = 2; goto finish; }
            if (uni_b(h_abs < min_step)) { status = 3; goto finish; }
            h = h_abs;
            t_new = t + h;
            if (t_new - A.t1 > 0) t_new = A.t1;
            h = uni_d(t_new - t);
            t_new = uni_d(t_new);
            h_abs = fabs(h);
            // stage planes at the three collocation times + predicted stage values (radau.py:445-448); the first
            // attempt of a step normally got them in the commit phase of the step before
            const bool have_setup = pre_setup && uni_b(h == pre_h);
            pre_setup = false;
            if (!have_setup) {
                const double f0 = uni_d(year_interp4(A.bld_t, A.bld_f, t + (h * RC0)));
                const double f1 = uni_d(year_interp4(A.bld_t, A.bld_f, t + (h * RC1)));
                const double f2 = uni_d(year_interp4(A.bld_t, A.bld_f, t + (h * RC2)));
                for (int task = wave; task < 3 * P.ny; task += nwaves) {
                    const int ti = task / P.ny, j = task - ti * P.ny;
                    YEAR_PLANE(ti, (ti == 0) ? f0 : ((ti == 1) ? f1 : f2), (ti == 0) ? A.KV[0] : ((ti == 1) ? A.KV[1] : YR_KV2), j)
                }
                if (col_wave) {
                    if (have_dense) {
                        PredictArgs PA;
                        PA.y = YR_Y; PA.yold = YR_YOLD; PA.zp = YR_ZP; PA.z = YR_Z; PA.w = A.W; PA.nv = nv;
                        PA.x0 = ((t + h * RC0) - dense_t_old) / dense_h;
                        PA.x1 = ((t + h * RC1) - dense_t_old) / dense_h;
                        PA.x2 = ((t + h * RC2) - dense_t_old) / dense_h;
                        predict_body<E, 1>(PA, wave, lane);
                    } else {
                        double zero[E];
#pragma unroll
                        for (int e = 0; e < E; ++e) zero[e] = 0.0;
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            store_col<E, 1>(YR_Z + i * nv, wave, lane, zero);
                            store_col<E>(A.W + i * nv, wave, lane, zero);
                        }
                    }
                }
                YEAR_SYNC()
            }
            if (jac_at_stage) { t_jac = t + (h * rc_jac); ++njev; current_jac = true; have_lu = false; }
            bool converged = false;
            while (!converged) {
                if (!have_lu) {
                    h_lu = h; have_lu = true;
                    m_real = uni_i(year_sweeps_for(A, MU_REAL / h));
                    m_cplx = uni_i(year_sweeps_for(A, MU_CR / h));
                    nlu += 2;
                    lu_cre = MU_REAL / h; lu_ccr = MU_CR / h; lu_cci = MU_CI / h;
                    factor_pending = true;
                }
                // simplified Newton iterations (radau.py:48-136)
                const double mreal = MU_REAL / h, mcr = MU_CR / h, mci = MU_CI / h;
                const int m = (m_real > m_cplx) ? m_real : m_cplx;
                double dW_norm_old = 0.0;
                bool has_old = false;
                have_rate = false; rate = 0.0;
                converged = false;
                int k = 0;
                for (k = 0; k < NEWTON_MAXITER; ++k) {
                    int src = 0;
                    for (int it = 0; it < m; ++it) {
                        const bool do_stage = it == 0, first = it == 0, do_update = it == m - 1, delta = m == 2;
                        const bool do_factor = factor_pending && it == 0;
                        FusedArgs FA = {};
                        FA.st.y = YR_Y; FA.st.z = YR_Z; FA.st.w = A.W;
                        // a single-sweep solve is ONE phase: its update writes the spare buffer (the neighbours still
                        // read the old stage values in this phase), then the buffers swap
                        FA.st.zout = (do_stage && do_update) ? YR_ZN : YR_Z;
                        FA.st.kv[0] = A.KV[0]; FA.st.kv[1] = A.KV[1]; FA.st.kv[2] = YR_KV2;
                        FA.st.br = A.BR; FA.st.bcr = A.BCR; FA.st.bci = A.BCI;
                        FA.st.nv = nv; FA.st.mreal = mreal; FA.st.mcr = mcr; FA.st.mci = mci;
                        FA.sw = A.fac;
                        FA.sw.br = A.BR; FA.sw.bcr = A.BCR; FA.sw.bci = A.BCI;
                        FA.sw.xr_old = src ? A.XR[1] : A.XR[0]; FA.sw.xcr_old = src ? A.XCR[1] : A.XCR[0];
                        FA.sw.xci_old = src ? A.XCI[1] : A.XCI[0];
                        FA.sw.xr_new = src ? A.XR[0] : A.XR[1]; FA.sw.xcr_new = src ? A.XCR[0] : A.XCR[1];
                        FA.sw.xci_new = src ? A.XCI[0] : A.XCI[1];
                        FA.sw.first = first ? 1 : 0;
                        FA.sw.cre = lu_cre; FA.sw.ccr = lu_ccr; FA.sw.cci = lu_cci;
                        FA.sw.f32 = 0;
                        FA.part = YEAR_PART();
                        FA.do_stage = do_stage ? 1 : 0; FA.do_update = do_update ? 1 : 0; FA.delta = delta ? 1 : 0;
                        if (col_wave) {
                            if (do_factor) newton_fused_body<E, KIND, 1, 1, 1>(P, FA, wave, lane);
                            else newton_fused_body<E, KIND, 0, 1, 1>(P, FA, wave, lane);
                        }
                        {   // algorithmic bytes, as nk2d_r_newton_fused counts them
                            double wd = 0.0;
                            if (do_stage) wd += 7.0 * Ntot + 7.0 * Pc + ((do_update || delta) ? 0.0 : 3.0 * Ntot);
                            wd += (first ? 2.0 : 4.0) * Pc + (3.0 * Ntot + 3.0 * 14.0 / E * Ntot);
                            if (do_factor) wd += Pc;
                            if (!do_stage && !delta) wd += 3.0 * Ntot;
                            if (!first) wd += 3.0 * Ntot;
                            if (!do_update) wd += 3.0 * Ntot;
                            if (do_update) wd += (do_stage ? 0.0 : Ntot) + 9.0 * Ntot;
                            words += wd;
                        }
                        ++nsweeps;
                        if (it == 0) factor_pending = false;
                        src = 1 - src;
                        if (do_stage && do_update) { const int tmp = zc; zc = zn; zn = tmp; }
                        YEAR_SYNC()
                    }
                    nsolve += 2; nfev += 3; ++nnewton;
                    const double sum = uni_d(year_part_sum(YEAR_PART(), P.ncol, lane));
                    ++pev;
                    const double dW_norm = sqrt(sum) / sqrt(3.0 * A.n_total);
                    if (uni_b(!(dW_norm == dW_norm))) break;
                    if (has_old) { rate = uni_d(dW_norm / dW_norm_old); have_rate = true; }
                    if (have_rate && uni_b(rate >= 1.0 || pow(rate, (double)(NEWTON_MAXITER - k)) / (1.0 - rate) * dW_norm > A.newton_tol)) break;
                    if (uni_b(dW_norm == 0.0 || (have_rate && rate / (1.0 - rate) * dW_norm < A.newton_tol))) { converged = true; break; }
                    dW_norm_old = dW_norm;
                    has_old = true;
                }
                n_iter = (k < NEWTON_MAXITER) ? k + 1 : NEWTON_MAXITER;
                if (!converged) {
                    if (current_jac) break;
                    // stale Jacobian: refresh it and repeat the iteration from the predicted stage values
                    // (radau.py:462-470: solve_collocation_system starts from Z0 again)
                    YEAR_JAC(YR_KV3)
                    if (col_wave) {
                        if (have_dense) {
                            PredictArgs PA;
                            PA.y = YR_Y; PA.yold = YR_YOLD; PA.zp = YR_ZP; PA.z = YR_Z; PA.w = A.W; PA.nv = nv;
                            PA.x0 = ((t + h * RC0) - dense_t_old) / dense_h;
                            PA.x1 = ((t + h * RC1) - dense_t_old) / dense_h;
                            PA.x2 = ((t + h * RC2) - dense_t_old) / dense_h;
                            predict_body<E, 1>(PA, wave, lane);
                        } else {
                            double zero[E];
#pragma unroll
                            for (int e = 0; e < E; ++e) zero[e] = 0.0;
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                store_col<E, 1>(YR_Z + i * nv, wave, lane, zero);
                                store_col<E>(A.W + i * nv, wave, lane, zero);
                            }
                        }
                    }
                    YEAR_SYNC()
                    t_jac = t; ++njev; current_jac = true; have_lu = false;
                }
            }
            if (!converged) {
                h_abs = uni_d(h_abs * 0.5);
                have_lu = false;
                newton_failed = true;
                continue;
            }
            // error estimate (radau.py:477-487)
            int buf = 0;
            {
                if (m_real <= 2) {
                    int src = 0;
                    for (int it = 0; it < m_real; ++it) {
                        ErrArgs EA = {};
                        EA.sw = A.fac;
                        EA.f = A.F; EA.z = YR_Z; EA.y = YR_Y; EA.nv = nv; EA.h = h; EA.part = YEAR_PART();
                        EA.sw.xr_old = src ? A.XR[1] : A.XR[0]; EA.sw.xr_new = src ? A.XR[0] : A.XR[1];
                        EA.stage = it; EA.last = (it == m_real - 1) ? 1 : 0;
                        if (col_wave) err_fused_body<E, KIND, 1>(P, EA, wave, lane);
                        src = 1 - src;
                        ++nsweeps;
                        YEAR_SYNC()
                    }
                    buf = src;
                } else {
                    if (col_wave) err_rhs_body<E, 1>(A.F, YR_Z, nv, h, A.BR, wave, lane);
                    // the first sweep reads only its own column's right-hand side: no barrier before it
                    int src = 0;
                    const int tr = wave / P.ny, j = wave - tr * P.ny;
                    for (int it = 0; it < m_real; ++it) {
                        SweepArgs SA = A.fac;
                        SA.br = A.BR; SA.nreal = P.ncol; SA.ntasks = P.ncol;
                        SA.xr_old = src ? A.XR[1] : A.XR[0]; SA.xr_new = src ? A.XR[0] : A.XR[1]; SA.first = (it == 0) ? 1 : 0;
                        if (col_wave) sweep_body<E, KIND, 1>(P, SA, j * P.tc + tr, lane);
                        src = 1 - src;
                        ++nsweeps;
                        YEAR_SYNC()
                    }
                    buf = src;
                    if (col_wave) err_norm_body<E, 1>(P, YR_Y, YR_Z + 2 * nv, buf ? A.XR[1] : A.XR[0], YEAR_PART(), wave, lane);
                    YEAR_SYNC()
                }
                ++nsolve;
            }
            {
                const double sum = uni_d(year_part_sum(YEAR_PART(), P.ncol, lane));
                ++pev;
                err = sqrt(sum) / sqrt(A.n_total);
            }
            safety = 0.9 * (2 * NEWTON_MAXITER + 1) / (2 * NEWTON_MAXITER + n_iter);
            if (rejected && uni_b(err > 1)) {
                // filtered estimate: error <- solve(fun(t, y + error) + Z^T E / h)  (radau.py:485-487)
                if (col_wave) {
                    double tmp[E];
                    load_col<E, 1>(buf ? A.XR[1] : A.XR[0], wave, lane, tmp);
                    store_col<E, 1>(A.TMP, wave, lane, tmp);
                }
                YEAR_SYNC()
                if (col_wave) err_rhs2_body<E, KIND, 1>(P, YR_Y, A.TMP, YR_KV3, YR_Z, nv, h, A.BR, wave, lane);
                ++nfev;
                int src = 0;
                const int tr = wave / P.ny, j = wave - tr * P.ny;
                for (int it = 0; it < m_real; ++it) {
                    SweepArgs SA = A.fac;
                    SA.br = A.BR; SA.nreal = P.ncol; SA.ntasks = P.ncol;
                    SA.xr_old = src ? A.XR[1] : A.XR[0]; SA.xr_new = src ? A.XR[0] : A.XR[1]; SA.first = (it == 0) ? 1 : 0;
                    if (col_wave) sweep_body<E, KIND, 1>(P, SA, j * P.tc + tr, lane);
                    src = 1 - src;
                    ++nsweeps;
                    YEAR_SYNC()
                }
                buf = src;
                ++nsolve;
                if (col_wave) err_norm_body<E, 1>(P, YR_Y, YR_Z + 2 * nv, buf ? A.XR[1] : A.XR[0], YEAR_PART(), wave, lane);
                YEAR_SYNC()
                const double sum = uni_d(year_part_sum(YEAR_PART(), P.ncol, lane));
                ++pev;
                err = sqrt(sum) / sqrt(A.n_total);
            }
            if (uni_b(err > 1)) {
                const double factor = year_predict_factor(h_abs, has_h_old, h_abs_old, err, has_err_old, err_old);
                h_abs = uni_d(h_abs * fmax(0.2, safety * factor));
                have_lu = false;
                rejected = true;
                ++nrejected;
            } else {
                accepted = true;
            }
        }
        const bool recompute_jac = uni_b(n_iter > 2 && have_rate && rate > 1e-3);
        double factor = year_predict_factor(h_abs, has_h_old, h_abs_old, err, has_err_old, err_old);
        factor = fmin(10.0, safety * factor);
        if (newton_failed && A.growth_cap > 0.0) factor = fmin(factor, A.growth_cap);
        factor = uni_d(factor);
        const double h_lu_used = h_lu;
        if (!recompute_jac && uni_b(factor < 1.2)) factor = 1;
        else have_lu = false;
        if (A.record && nrec < A.record_cap && wave == 0 && lane == 0) {
            double* r = A.record + (size_t)nrec * NK2D_SCHED_WIDTH;
            r[0] = t; r[1] = t_new; r[2] = h; r[3] = (double)n_iter; r[4] = t_jac; r[5] = h_lu_used;
            r[6] = err; r[7] = A.fingerprint;
        }
        ++nrec;
        // y_new, f_new = fun(t_new, y_new)
        const double h_abs_next = uni_d(h_abs * factor);
        const bool jac_due = recompute_jac || (A.jac_fresh != 0);
        bool fusedb = uni_b(t + h == t_new) && uni_b(t_new < A.t1);
        double h2 = 0.0;
        if (fusedb) {
            const double min_step2 = 10.0 * fabs(nextafter(t_new, INFINITY) - t_new);
            double h_abs2 = h_abs_next;
            if (h_abs2 > A.max_step) h_abs2 = A.max_step;
            else if (h_abs2 < min_step2) h_abs2 = min_step2;
            double t_new2 = t_new + h_abs2;
            if (t_new2 - A.t1 > 0) t_new2 = A.t1;
            h2 = uni_d(t_new2 - t_new);
            fusedb = uni_b(isfinite(h2) && h2 > 0.0);
        }
        if (fusedb) {
            // ONE phase for the whole boundary: commit, Jacobian at t_new where due, and the next attempt's planes and
            // predicted stage values (every piece reads what the Newton iteration left or what its own wave writes)
            const double f0 = uni_d(year_interp4(A.bld_t, A.bld_f, t_new + (h2 * RC0)));
            const double f1 = uni_d(year_interp4(A.bld_t, A.bld_f, t_new + (h2 * RC1)));
            const double f2 = uni_d(year_interp4(A.bld_t, A.bld_f, t_new + (h2 * RC2)));
            for (int task = wave; task < 3 * P.ny; task += nwaves) {
                const int ti = task / P.ny, j = task - ti * P.ny;
                YEAR_PLANE(ti, (ti == 0) ? f0 : ((ti == 1) ? f1 : f2), (ti == 0) ? A.KV[0] : ((ti == 1) ? A.KV[1] : YR_KV3), j)
            }
            if (jac_due && !jac_at_stage) { YEAR_JAC(YR_KV2) }      // the third stage plane is the plane at t_new
            if (col_wave) {
                commit_tend_body<E, KIND, 1>(P, YR_Y, YR_Z + 2 * nv, YR_KV2, YR_YOLD, A.F, wave, lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's y_new is in memory before it reads it back
                PredictArgs PA;
                PA.y = YR_YOLD; PA.yold = YR_Y; PA.zp = YR_Z; PA.z = YR_ZP; PA.w = A.W; PA.nv = nv;
                PA.x0 = ((t_new + h2 * RC0) - t) / (t_new - t);
                PA.x1 = ((t_new + h2 * RC1) - t) / (t_new - t);
                PA.x2 = ((t_new + h2 * RC2) - t) / (t_new - t);
                predict_body<E, 1>(PA, wave, lane);
            }
            swapKV ^= 1;
            swapY ^= 1;
            { const int tmp = zc; zc = zp; zp = tmp; }
            have_dense = true; dense_t_old = t; dense_h = uni_d(t_new - t);
            t = t_new;
            ++nsteps; ++nfev;
            YEAR_SYNC()
            pre_setup = true; pre_h = h2;
            if (jac_at_stage) {
                current_jac = false;     // came with the planes; booked when the attempt starts
            } else if (jac_due) {
                t_jac = t; ++njev; current_jac = true;
                if (!recompute_jac) have_lu = false;
            } else {
                current_jac = false;
            }
        } else {
            if (uni_b(t + h == t_new)) {
                swapKV ^= 1;        // the third stage plane is the plane at t_new
            } else {
                const double fr = uni_d(year_interp4(A.bld_t, A.bld_f, t_new));
                for (int j = wave; j < P.ny; j += nwaves) vmix_col<E, 1>(P, A.bldmin, A.vy0, A.vy1, A.hw, fr, YR_KV3, j, lane);
                YEAR_SYNC()
            }
            if (col_wave) commit_tend_body<E, KIND, 1>(P, YR_Y, YR_Z + 2 * nv, YR_KV3, YR_YOLD, A.F, wave, lane);
            swapY ^= 1;
            { const int tmp = zc; zc = zp; zp = tmp; }
            have_dense = true; dense_t_old = t; dense_h = uni_d(t_new - t);
            t = t_new;
            ++nsteps; ++nfev;
            YEAR_SYNC()
            if (recompute_jac) {
                YEAR_JAC(YR_KV3)
                YEAR_SYNC()
                t_jac = t; ++njev; current_jac = true;
            } else {
                current_jac = false;
            }
        }
        h_abs_old_s = h_abs_s; has_old_h = true;
        err_old_s = err; has_old_err = true;
        h_abs_s = h_abs_next;
    }
finish:
    if (wave == 0 && lane == 0) {
        double* o = A.out;
        o[0] = (double)status; o[1] = t; o[2] = (double)nfev; o[3] = (double)njev; o[4] = (double)nlu;
        o[5] = (double)nsteps; o[6] = (double)nrejected; o[7] = (double)nnewton; o[8] = (double)nsolve;
        o[9] = (double)nsweeps; o[10] = (double)nrec; o[11] = (double)swapY; o[12] = (double)(zc + 4 * zp + 16 * zn);
        o[13] = (double)swapKV; o[14] = 8.0 * words; o[15] = (double)bar.epoch; o[16] = t_jac;
    }
#undef YEAR_SYNC
#undef YEAR_JAC
#undef YEAR_PLANE
#undef YR_Y
#undef YR_YOLD
#undef YR_Z
#undef YR_ZP
#undef YR_ZN
#undef YR_ZBUF
#undef YR_KV2
#undef YR_KV3
#undef YEAR_PART
}

// host side: run the stepping loop of a forward year in the persistent kernel.  The caller has done SciPy's
// prologue (f0 in F, initial step size, Jacobian at t0 with the plane of t0 in KV[3]).  Returns 1 when the
// launch is not possible (grid not fully resident): the caller then steps under host control.
int nk2d_year_persistent(nk2d_ctx* c, double h_abs0, double newton_tol, double max_step, double n_total,
                         double* record, int64_t record_cap, int64_t* record_n) {
    if (c->kind != 0) return 1;
    const int nblk = nk2d_grid(c->ncol);
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
    const int min_sweeps = (c->min_sweeps > 1 && nk2d_has_lateral(c)) ? 2 : 1;
    if (c->yr_lin_tol != c->d.lin_tol) {
        // sweeps per shift bucket with the host's arithmetic (nk2d_sweeps_for), looked up on the device
        std::vector<int> mtab(c->rho_tab.size());
        for (size_t k = 0; k < mtab.size(); ++k)
            mtab[k] = std::max(nk2d_sweeps_for(c, c->rho_c0 * std::pow(10.0, ((double)k + 0.5) * c->rho_dlog)), min_sweeps);
        if (!mtab.empty())
            NK2D_CHECK(c, hipMemcpy(c->YR_MTAB, mtab.data(), sizeof(int) * mtab.size(), hipMemcpyHostToDevice));
        c->yr_lin_tol = c->d.lin_tol;
    }
    if (record && record_cap > c->yr_rec_cap) {
        if (c->YR_REC) NK2D_CHECK(c, hipFree(c->YR_REC));
        NK2D_CHECK(c, hipMalloc((void**)&c->YR_REC, sizeof(double) * NK2D_SCHED_WIDTH * record_cap));
        c->yr_rec_cap = record_cap;
    }
    NK2D_CHECK(c, hipMemsetAsync(c->YR_SYNC, 0, yr_sync_bytes(c), nk2d_s(c)));
    YearArgs A = {};
    A.Y = c->Y; A.YOLD = c->YOLD; A.F = c->F; A.Z = c->Z; A.ZP = c->ZP; A.ZN = c->ZN; A.W = c->W;
    A.BR = c->BR; A.BCR = c->BCR; A.BCI = c->BCI;
    for (int i = 0; i < 2; ++i) { A.XR[i] = c->XR[i]; A.XCR[i] = c->XCR[i]; A.XCI[i] = c->XCI[i]; }
    A.TMP = c->TMP;
    for (int i = 0; i < 4; ++i) A.KV[i] = c->KV[i];
    fill_factor_args(c, A.fac);
    A.fac.f32 = 0;
    A.PART = c->YR_PART;
    A.t0 = c->d.t0; A.t1 = c->d.t1; A.h_abs0 = h_abs0; A.max_step = max_step; A.newton_tol = newton_tol;
    A.n_total = n_total; A.growth_cap = c->growth_cap; A.jac_fresh = c->jac_fresh; A.jac_stage = c->jac_fresh ? c->jac_stage : -1;
    for (int i = 0; i < 4; ++i) { A.bld_t[i] = c->d.bld_tvals[i]; A.bld_f[i] = c->d.bld_fvals[i]; }
    A.bldmin = c->d.bldepth_min; A.vy0 = c->d.vmix_log_shallow; A.vy1 = c->d.vmix_log_deep; A.hw = c->d.vmix_half_width;
    A.m_tab = c->YR_MTAB; A.n_tab = (int)c->rho_tab.size(); A.rho_c0 = c->rho_c0; A.rho_dlog = c->rho_dlog;
    A.arrive = (unsigned*)c->YR_SYNC; A.abort_flag = (int*)((char*)c->YR_SYNC + 4096);
    A.out = c->YR_OUT;
    A.record = record ? c->YR_REC : nullptr;
    A.record_cap = record ? record_cap : 0;
    A.fingerprint = nk2d_fingerprint(c);
    A.spin_ticks = (long long)(c->barrier_timeout_ms * 1.0e5);
    A.fences = c->year_fences;
    DevP P = make_devp(c);
    P.guard = nullptr;
    void* args[2] = {&P, &A};
    hipError_t rc = hipErrorInvalidValue;
    NK2D_CHECK(c, hipEventRecord(c->yr_ev[0], nk2d_s(c)));
    {
        std::lock_guard<std::mutex> coop(coop_launch_mutex());
        NK2D_DISPATCH_E(c->E, rc = hipLaunchCooperativeKernel((const void*)k_year_persistent<EE, 0>, dim3(nblk), dim3(NK2D_BLOCK),
                                                               args, 0, nk2d_s(c)));
    }
    if (rc == hipErrorCooperativeLaunchTooLarge) { (void)hipGetLastError(); return 1; }
    NK2D_CHECK(c, rc);
    NK2D_CHECK(c, hipEventRecord(c->yr_ev[1], nk2d_s(c)));
    NK2D_CHECK(c, hipMemcpyAsync(c->hYR_OUT, c->YR_OUT, sizeof(double) * 32, hipMemcpyDeviceToHost, nk2d_s(c)));
    NK2D_CHECK(c, hipStreamSynchronize(nk2d_s(c)));
    const double* o = c->hYR_OUT;
    const int status = (int)o[0];
    // the buffers swapped roles on the device an odd or even number of times
    if ((int)o[11]) std::swap(c->Y, c->YOLD);
    {   // the three stage buffers in the roles the device left them in
        double* bufs[3] = {c->Z, c->ZP, c->ZN};
        const int code = (int)o[12];
        c->Z = bufs[code & 3]; c->ZP = bufs[(code >> 2) & 3]; c->ZN = bufs[(code >> 4) & 3];
    }
    if ((int)o[13]) std::swap(c->KV[2], c->KV[3]);
    c->st.nfev += (int64_t)o[2]; c->st.njev += (int64_t)o[3]; c->st.nlu += (int64_t)o[4];
    c->st.nsteps += (int64_t)o[5]; c->st.nrejected += (int64_t)o[6]; c->st.nnewton += (int64_t)o[7];
    c->st.nsolve += (int64_t)o[8]; c->st.nsweeps += (int64_t)o[9]; c->st.nlaunch += 1;
    float ms = 0.f;
    NK2D_CHECK(c, hipEventElapsedTime(&ms, c->yr_ev[0], c->yr_ev[1]));
    if (c->prof_every > 0) {
        // profile window = the whole-year kernel: one "launch", its algorithmic bytes, its duration
        c->prof_ms_sum += ms; c->prof_windows += 1; c->prof_cnt += 1;
        c->sweep_launches += 1; c->sweep_bytes += o[14]; c->fused_bytes_all += o[14];
    }
    // a grid barrier timed out (a co-tenant held the chip, say): the input is intact (the caller re-copies it), the year
    // reruns under host control -- counted, not failed
    if (status == 1) return 2;
    if (status == 2) return nk2d_fail(c, "Radau: step size is not finite (non-finite state or tendency)", -3);
    if (status == 3) return nk2d_fail(c, "Radau: required step size is less than spacing between numbers", -3);
    const int64_t nrec = (int64_t)o[10];
    if (record && nrec > 0) {
        const int64_t ncopy = std::min<int64_t>(nrec, record_cap);
        NK2D_CHECK(c, hipMemcpy(record, c->YR_REC, sizeof(double) * NK2D_SCHED_WIDTH * ncopy, hipMemcpyDeviceToHost));
    }
    if (record_n) *record_n = nrec;
    if (record && nrec > record_cap && record != c->own_rec.data())
        return nk2d_fail(c, "nk2d_comp_fcn: schedule record buffer too small", -4);
    return 0;
}
