// The projected L-BFGS step of the outer iteration as ONE WAVE (gfx950, wave64, fp64) -- the form the one-launch solves use
// (tiny.hpp: one workgroup, <= 64 tokens, E = 1; E = 2 served the cooperating-workgroups solve of DESIGN.md's "tried" table).
//
// Lane L owns tokens / group variables L + 64 e, e < E: every vector of the update is E registers per lane, every reduction
// one DPP butterfly (no barrier, no LDS round trip, no memory), the history pairs sit in LDS rows of 64 E.  The solver
// state lives in registers from the first iteration of a solve to the last; global memory sees it at the start and at
// the end.  Same iteration as update_generic_body (kernels.hpp) and oracle/cfmm_oracle.c:oracle_step -- price ties,
// bounds, any memory <= MAX_MEMORY --, different layout.                          reference: arbitrage.py:82 (prob.solve())
#pragma once
#include "kernels.hpp"

namespace cfmm {

__device__ __forceinline__ int wuni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double wuni(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ double lane_value(double v, int l)          // l wave-uniform
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

template <int E>
struct WaveUpdate {
    static constexpr int W = 64 * E;             // row width of the LDS vectors (history rows, group sums)
    DevState st;
    double s[E], s_t[E], Gs[E], d[E], Ds[E], glo[E], ghi[E];      // group variables r = L + 64 e
    double cj[E], hj[E], offj[E], nuj[E], psi_a[E], nu_a[E];      // tokens j = L + 64 e
    int ct[E], grp[E];
    double rho_l;                                // 1 / s'y of history slot L
    bool tin[E], gin[E], ties;
    int L, n, ng;

    template <class F> __device__ __forceinline__ double wsum(F f) const
    {
        double v = 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) v += f(e);
        return wave_allsum(v);
    }
    template <class F> __device__ __forceinline__ double wmax(F f) const
    {
        double v = f(0);
#pragma unroll
        for (int e = 1; e < E; ++e) v = fmax(v, f(e));
        return wave_allmax(v);
    }

    // the state a solve starts from (start_kernel has written it)
    __device__ __forceinline__ void load(const UpdArgs &a, int lane, int n_, int ng_)
    {
        L = lane; n = n_; ng = ng_; ties = ng != n; rho_l = 0.0;
        st = load_state(a.st);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int j = L + 64 * e;
            tin[e] = j < n; gin[e] = j < ng;
            s[e] = s_t[e] = Gs[e] = d[e] = Ds[e] = glo[e] = ghi[e] = 0.0;
            cj[e] = hj[e] = offj[e] = nuj[e] = psi_a[e] = nu_a[e] = 0.0; ct[e] = 0; grp[e] = 0;
            if (gin[e]) { s[e] = a.s[j]; s_t[e] = a.s_t[j]; d[e] = a.d[j]; glo[e] = a.glo[j]; ghi[e] = a.ghi[j]; }
            if (tin[e]) { cj[e] = a.c[j]; hj[e] = a.h[j]; offj[e] = a.off[j]; ct[e] = a.ctype[j]; grp[e] = a.grp[j]; nuj[e] = a.nu[j]; nu_a[e] = a.nu_acc[j]; }
        }
        st.status = wuni(st.status); st.evals = wuni(st.evals); st.iters = wuni(st.iters); st.first = wuni(st.first);
        st.hist = wuni(st.hist); st.head = wuni(st.head); st.nrej = wuni(st.nrej);
        st.f = wuni(st.f); st.t_step = wuni(st.t_step);
    }

    // One step from the evaluation at the trial prices: psi[e] / dg[e] = net trade / diagonal-metric entry of this lane's
    // tokens, fpools = sum_i arb_i.  Sh / Yh: history rows [MAX_MEMORY][W] in LDS; q, q2: [W] LDS scratch (group sums, the
    // exchange of the trial point).  The next trial prices are left in nuj (and, for the lane's tokens, written to nu_out).
    __device__ __forceinline__ void step(const UpdArgs &a, int M, const double (&psi)[E], const double (&dgin)[E], double fpools,
                                         double *Sh, double *Yh, double *q, double *q2, double *nu_out)
    {
        const bool first = st.first != 0;
        // ---- A. residuals, group gradient at the trial point -----------------------------------------------------------
        double rj[E], Gs_t[E];
#pragma unroll
        for (int e = 0; e < E; ++e) rj[e] = psi[e] + hj[e];
        const double f_t = wsum([&](int e) { return tin[e] ? (nuj[e] - cj[e]) * hj[e] : 0.0; }) + fpools;
        const double gapv = wsum([&](int e) { return tin[e] ? (nuj[e] - cj[e]) * rj[e] : 0.0; });
        const double viol = wmax([&](int e) { return !tin[e] ? 0.0 : (ct[e] == 0 ? fmax(-rj[e], 0.0) : (ct[e] == 1 ? fabs(rj[e]) : 0.0)); });
        const double scale = wmax([&](int e) { return tin[e] ? fmax(fabs(psi[e]), fabs(hj[e])) : 0.0; });
        if (ties) {                                  // group sums through LDS (one wave: its LDS operations execute in order)
#pragma unroll
            for (int e = 0; e < E; ++e) { q[L + 64 * e] = 0.0; q2[L + 64 * e] = 0.0; }
#pragma unroll
            for (int e = 0; e < E; ++e) if (tin[e]) { unsafeAtomicAdd(&q[grp[e]], nuj[e] * rj[e]); if (first) unsafeAtomicAdd(&q2[grp[e]], dgin[e]); }
#pragma unroll
            for (int e = 0; e < E; ++e) { Gs_t[e] = gin[e] ? q[L + 64 * e] : 0.0; if (first) Ds[e] = gin[e] ? q2[L + 64 * e] : 0.0; }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) { Gs_t[e] = tin[e] ? nuj[e] * rj[e] : 0.0; if (first) Ds[e] = dgin[e]; }
        }
        st.evals += 1;

        // ---- B. accept test ------------------------------------------------------------------------------------------------
        double ds[E];
#pragma unroll
        for (int e = 0; e < E; ++e) ds[e] = gin[e] ? s_t[e] - s[e] : 0.0;
        bool accept = first;
        if (!first) {
            const double dd0 = wsum([&](int e) { return Gs[e] * ds[e]; }), dd1 = wsum([&](int e) { return Gs_t[e] * ds[e]; });
            accept = lbfgs::accept(f_t, st.f, a.armijo, dd0, dd1);
        }
        if (!accept) {
            lbfgs::reject(st);
        } else {
            // ---- C. curvature pair, move the accepted point ------------------------------------------------------------
            if (!first) {
                double y1[E];
#pragma unroll
                for (int e = 0; e < E; ++e) { y1[e] = gin[e] ? Gs_t[e] - Gs[e] : 0.0; Sh[st.head * W + L + 64 * e] = ds[e]; Yh[st.head * W + L + 64 * e] = y1[e]; }
                const double sy = wsum([&](int e) { return ds[e] * y1[e]; }), ss = wsum([&](int e) { return ds[e] * ds[e]; }),
                             yy = wsum([&](int e) { return y1[e] * y1[e]; });
                if (lbfgs::pair_ok(sy, ss, yy)) {
                    if (L == st.head) rho_l = rcp_nr(sy);
                    st.head = (st.head + 1) % M;
                    if (st.hist < M) st.hist += 1;
                }
                st.iters += 1;
            }
#pragma unroll
            for (int e = 0; e < E; ++e) { s[e] = s_t[e]; Gs[e] = Gs_t[e]; psi_a[e] = psi[e]; nu_a[e] = nuj[e]; }
            st.first = 0;
            lbfgs::certify(st, f_t, gapv, viol, scale, wsum([&](int e) { return gin[e] ? lbfgs::pg_entry(Gs[e], s[e], glo[e], ghi[e]) : 0.0; }));
            if (lbfgs::converged(st, a.pg_rule, a.tol_gap, a.tol_infeas)) {
                st.status = 1;
            } else {
                // ---- D. two-loop recursion with the diagonal metric ----------------------------------------------------
                bool active[E];
                double qv[E], H[E];
#pragma unroll
                for (int e = 0; e < E; ++e) { active[e] = !gin[e] || is_active(s[e], glo[e], ghi[e], Gs[e]); qv[e] = active[e] ? 0.0 : Gs[e]; }
                const double gp = wsum([&](int e) { return qv[e] * qv[e]; });
                double alpha[MAX_MEMORY];
#pragma unroll
                for (int k = 0; k < MAX_MEMORY; ++k) {
                    alpha[k] = 0.0;
                    if (k < st.hist) {
                        const int i = (st.head - 1 - k + 2 * M) % M;
                        const double al = lane_value(rho_l, i) * wsum([&](int e) { return Sh[i * W + L + 64 * e] * qv[e]; });
                        alpha[k] = al;
#pragma unroll
                        for (int e = 0; e < E; ++e) qv[e] -= al * Yh[i * W + L + 64 * e];
                    }
                }
#pragma unroll
                for (int e = 0; e < E; ++e) { H[e] = Ds[e] + fmax(Gs[e], 0.0); qv[e] = (gin[e] && H[e] > 0.0) ? qv[e] / H[e] : 0.0; }
#pragma unroll
                for (int k = MAX_MEMORY - 1; k >= 0; --k) {
                    if (k < st.hist) {
                        const int i = (st.head - 1 - k + 2 * M) % M;
                        const double beta = lane_value(rho_l, i) * wsum([&](int e) { return Yh[i * W + L + 64 * e] * qv[e]; });
#pragma unroll
                        for (int e = 0; e < E; ++e) qv[e] += Sh[i * W + L + 64 * e] * (alpha[k] - beta);
                    }
                }
                double dv[E];
#pragma unroll
                for (int e = 0; e < E; ++e) dv[e] = active[e] ? 0.0 : -qv[e];
                const double dsum = wsum([&](int e) { return dv[e] * Gs[e]; });
                double dmx = wmax([&](int e) { return fabs(dv[e]); });
                if (!(dsum < 0.0) && gp > 0.0) {        // not a descent direction: restart
                    st.hist = 0;
#pragma unroll
                    for (int e = 0; e < E; ++e) dv[e] = (active[e] || !(H[e] > 0.0)) ? 0.0 : -Gs[e] / H[e];
                    dmx = wmax([&](int e) { return fabs(dv[e]); });
                }
#pragma unroll
                for (int e = 0; e < E; ++e) d[e] = dv[e];
                st.t_step = lbfgs::step_cap(dmx, a.max_step);
            }
        }
        // ---- E. next trial point -----------------------------------------------------------------------------------------------
        if (st.status == 0) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                double v = s[e] + st.t_step * d[e];
                v = fmax(v, glo[e]);
                v = fmin(v, ghi[e]);
                s_t[e] = gin[e] ? v : 0.0;
                q[L + 64 * e] = s_t[e];              // (same wave: ordered)
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                nuj[e] = tin[e] ? exp(q[grp[e]] + offj[e]) : 0.0;
                if (tin[e]) nu_out[L + 64 * e] = nuj[e];
            }
            if (st.evals >= a.max_evals) st.status = 3;
        }
        st.status = wuni(st.status); st.hist = wuni(st.hist); st.head = wuni(st.head);
    }

    // the state reaches global memory once, when the solve ends
    __device__ __forceinline__ void store(const UpdArgs &a)
    {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int j = L + 64 * e;
            if (tin[e]) { a.nu_acc[j] = nu_a[e]; a.psi_acc[j] = psi_a[e]; a.nu[j] = nuj[e]; }
            if (gin[e]) { a.s[j] = s[e]; a.s_t[j] = s_t[e]; a.Gs[j] = Gs[e]; a.d[j] = d[e]; a.Ds[j] = Ds[e]; }
        }
        if (L == 0) {
            if (st.status == 0) st.status = 3;           // (the launch's budget is the solve's)
            store_state(a.st, st); a.nu[n] = 1.0; report_progress(a, st);
        }
    }
};

}  // namespace cfmm
