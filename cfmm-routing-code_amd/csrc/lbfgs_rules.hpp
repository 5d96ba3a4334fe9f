// The decision rules of the projected L-BFGS outer iteration, in ONE place.
//
// The step exists in several LAYOUTS, because what bounds it differs with the size of the problem and with where it runs --
//   update_generic_body   any token count, one workgroup, vectors in global memory            (kernels.hpp)
//   update_reg_kernel     <= 2048 tokens, history in registers, sequential two-loop recursion  (kernels.hpp)
//   update_gram_kernel    <= 1024 tokens, Gram form: the recursion on scalars                  (kernels.hpp)
//   iter_kernel           the Gram form inside the evaluation launch, one-wave scalar section  (iterate.hpp)
//   WaveUpdate<E>         one wave, state in registers: the one-launch solves                  (onewave.hpp)
// -- but every one of them takes the SAME decisions from the same reduced scalars: whether the trial point is accepted,
// whether the curvature pair enters the history, the certificates and the stopping rule, which pairs are in the window,
// the two-loop recursion on Gram scalars, the cap on the step.  Those live here and nowhere else; the layouts only
// produce their inputs.  oracle/cfmm_oracle.c:oracle_step restates the same rules independently (it is the checker).
//                                                                        reference: arbitrage.py:82 (prob.solve())
#pragma once
#include "pool_math.hpp"

namespace cfmm {
namespace lbfgs {

// ---- the utility table (utilities beyond linear-plus-box) ------------------------------------------------
// The reference's utilities are linear with a box: U(Psi) = c'Psi over Psi + h >= 0 (or = 0, or free), whose conjugate is the
// linear term (nu - c)'h over the bounds nu >= c that the projected iteration carries.  A SEPARABLE concave utility
// u_j(Psi_j) per token enters the same dual through its own conjugate ubar_j(nu_j) = sup_P (u_j(P) - nu_j P), attained at
// P*_j(nu_j) = (u_j')^-1(nu_j):   g(nu) = sum_j ubar_j(nu_j) + sum_i arb_i(nu),   dg/dnu_j = psi_j - P*_j(nu_j),
// no bound on nu_j, and by Fenchel-Young the duality gap of the pair (nu, psi) is sum_j [ubar_j(nu_j) + nu_j psi_j - u_j(psi_j)]
// (>= 0, zero iff psi_j = P*_j) -- for the linear-box entry that IS (nu_j - c_j)(psi_j + h_j).  One entry = these five numbers.
// Entries (ctype of the token; the token's c and h carry the entry's two parameters):
//   CFMM_ULOG  = 3   u(P) = c log(P + h),  c > 0, h >= 0:   P* = c / nu - h,   ubar = c log(c / nu) - c + nu h
//   CFMM_UQUAD = 4   u(P) = c P - P^2 / (2 h),  h > 0:       P* = h (c - nu),   ubar = h (c - nu)^2 / 2
// Served by the generic (two-launch) outer iteration only: update_generic_body and oracle/cfmm_oracle.c:oracle_step.
struct UtilityTerm { double pstar, ubar, uval, viol, curv; };        // curv: d^2/ds^2 of ubar(e^s), for the diagonal metric
__host__ __device__ inline bool smooth_utility(int ctype) { return ctype >= 3; }
__device__ __forceinline__ UtilityTerm utility_term(int ctype, double c, double h, double nu, double psi)
{
    UtilityTerm t;
    if (ctype == 3) {
        t.pstar = c / nu - h;
        t.ubar = c * log(c / nu) - c + nu * h;
        const double arg = psi + h;
        t.viol = fmax(-arg, 0.0);
        t.uval = c * log(fmax(arg, 1e-300));
        t.curv = nu * h;
    } else {
        const double dlt = c - nu;
        t.pstar = h * dlt;
        t.ubar = 0.5 * h * dlt * dlt;
        t.viol = 0.0;
        t.uval = c * psi - 0.5 * psi * psi / h;
        t.curv = h * nu * (2.0 * nu - c);
    }
    return t;
}

// the trial point is accepted: sufficient decrease of the dual value (Armijo on the projected move), or -- below the
// rounding noise of the value -- the approximate Wolfe condition on the directional derivatives
//   f_t, f: value at the trial / accepted point; gds = G's, gtds = G_t's with s the move, G / G_t the group gradients
__device__ __forceinline__ bool accept(double f_t, double f, double armijo, double gds, double gtds)
{
    return (f_t == f_t) && ((f_t <= f + armijo * gds) || (f_t <= f + 1e-11 * fmax(1.0, fabs(f)) && gtds <= 0.8 * fabs(gds)));
}

// a rejected trial point: halve the step; a step below 1e-9 means the line search has stalled (status 2)
template <class State>
__device__ __forceinline__ void reject(State &st)
{
    st.t_step *= 0.5;
    st.nrej += 1;
    if (st.t_step < 1e-9) st.status = 2;
}

// the curvature pair (s, y) enters the history iff s'y > 1e-12 |s| |y|  (written without the square roots)
__device__ __forceinline__ bool pair_ok(double sy, double ss, double yy) { return sy > 0.0 && sy * sy > 1e-24 * ss * yy; }

// certificates of the accepted point: relative duality gap |(nu - c)'(psi + h)| / max(1, |g|), relative
// infeasibility, the primal value c'psi = g - (nu - c)'(psi + h), and the projected-gradient value
template <class State>
__device__ __forceinline__ void certify(State &st, double f_t, double gapv, double viol, double scale, double pg_sum)
{
    st.f = f_t;
    const double rf = rcp_nr(fmax(1.0, fabs(f_t)));      // (v_rcp_f64 + two Newton steps: ~1 ulp, a fifth of the IEEE division's chain)
    st.gap = fabs(gapv) * rf;
    st.infeas = viol * rcp_nr(fmax(scale, 1e-300));
    st.primal = f_t - gapv;
    st.pg = pg_sum * rf;
}
template <class State>
__device__ __forceinline__ bool converged(const State &st, int pg_rule, double tol_gap, double tol_infeas)
{
    return pg_rule ? (st.pg <= tol_gap) : (st.gap <= tol_gap && st.infeas <= tol_infeas);
}

// |projected gradient| of one group variable at s with bounds [lo, hi] and gradient G
__device__ __forceinline__ double pg_entry(double G, double s, double lo, double hi)
{
    double v = G;
    if (lo == hi) v = 0.0;
    else if (s <= lo + 1e-14) v = fmin(G, 0.0);
    else if (s >= hi - 1e-14) v = fmax(G, 0.0);
    return fabs(v);
}

// how many of the STORED pairs stay in the window behind the (possibly) new one
__device__ __forceinline__ int keep_old(bool was_first, bool new_pair, int old_hist, int M)
{
    return was_first ? 0 : (new_pair ? (old_hist < M ? old_hist : M - 1) : old_hist);
}

// The two-loop recursion on Gram scalars, P pairs newest first (pair 0 = the new one), rho[k] = 1 / s_k'y_k or 0 for a
// pair outside the window.  U(k) = s_k'q0, V(k) = y_k'H0 q0, SY(k, j) = s_k'y_j (k > j), YHY(k, j) = y_k'H0 y_j (k >= j).
// On return d = -(H0 (q0 - sum al_k y_k) + sum ga_k s_k).
template <int P, class FU, class FV, class FSY, class FYHY>
__device__ __forceinline__ void gram_two_loop(const double (&rho)[P], FU U, FV V, FSY SY, FYHY YHY, double (&al)[P], double (&ga)[P])
{
#pragma unroll
    for (int k = 0; k < P; ++k) {
        double t = U(k);
#pragma unroll
        for (int j = 0; j < k; ++j) t -= al[j] * SY(k, j);
        al[k] = rho[k] * t;
    }
#pragma unroll
    for (int k = P - 1; k >= 0; --k) {
        double t = V(k);
#pragma unroll
        for (int j = 0; j < P; ++j) t -= al[j] * (j > k ? YHY(j, k) : YHY(k, j));
#pragma unroll
        for (int j = k + 1; j < P; ++j) t += ga[j] * SY(j, k);
        ga[k] = al[k] - rho[k] * t;
    }
}

// the step along d is capped at max_step in log-price (max |d| from the reduction)
__device__ __forceinline__ double step_cap(double dmax, double max_step) { return dmax > max_step ? max_step * rcp_nr(dmax) : 1.0; }

}  // namespace lbfgs
}  // namespace cfmm
