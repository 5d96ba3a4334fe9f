// The K-asset trading-function table (the half round 3 left open): a K-asset trading function other than the
// weighted geometric mean is ONE struct PhiK<KIND> here, and its pools ride in the generic K-asset bucket (columns idx, R,
// fee, param: include/cfmm.h, cfmm_upload_poolsG).  "A pool is whatever constraint line is written" (arbitrage.py:63-74);
// the shipped Balancer pool is 4-asset (arbitrage.py:65).
//
// One pool's arbitrage subproblem (arbitrage.py:51-52, 60, 63-74) in the new reserves x = R + gamma D - L:
//     minimise  sum_j c_j(x_j)   s.t.  phi(x) >= phi(R),      c_j(x) = nu_j (x - R_j) / gamma  (x >= R_j: deposit)  |  nu_j (x - R_j)  (withdraw)
// KKT with multiplier m > 0:  m phi_j(x) = nu_j / gamma on deposited legs, = nu_j on withdrawn legs, in between on untouched ones.
//
// The SMOOTH family the generic search serves: the gradient depends on x through the leg's own reserve and ONE coupling scalar,
//     phi_j(x) = f(x_j, s),   s = S(x),   f decreasing in x_j
// more precisely through s / x_j:  phi_j = F(s / x_j)  (n-asset stableswap  sum x - alpha / prod x :  s = alpha / prod x,
// F(u) = 1 + u;  the weighted geometric mean has the same shape with s = phi, F_j(u) = w_j u).  Given (m, s) every leg is in
// closed form,  x_j(m, s) = clip(R_j;  s / G(nu_j / (gamma m)),  s / G(nu_j / m)),  G = F^-1   [deposit level <= withdraw level]
// and two scalar equations remain:  S(x(m, s)) = s  (inner, monotone in s)  and  phi(x(m, s(m))) = phi(R)  (outer, monotone in m).
// pool_generic_k<KIND, K> solves them by nested bisection in (log m, log s): safeguarded by construction -- no derivative, no
// starting point; in log space the inner residual is piecewise linear, so its 64 steps cost adds and compares only -- which
// is the price of "a new function is one table entry": the search needs `prep`, `coupling_log`, `log_G`, `marginal` and `value_minus`
// only.  A function outside the family (the piecewise-linear constant sum: its levels jump) overrides `solve` with its own
// closed form and still shares bucket, kernels, tenders and host.
//
// First-order path only (exact pool solutions; the diagonal metric by a price perturbation of the same solver).  The
// second-order path refuses networks that hold such pools (cfmm_hip.hip: newton_supported): their generalised Hessian block
// needs the implicit derivative of the two-level root, which is not built.
#pragma once
#include "kernels.hpp"

namespace cfmm {

constexpr int N_KINDSK = 2;           // CFMM_POOLK_KINDS: 0 stableswap (sum x - alpha / prod x), 1 constant sum
constexpr int GK_THREADS = 256;

struct BucketG {
    long long m;
    const int *idx;                   // pool-major legs: leg j of pool i at [i K + j]
    const double *R;
    const double *fee, *param;        // per pool
    // derived on the device behind the upload (gk_derive_kernel, round 5): 1 / fee; the coupling at the pool's own reserves
    // s_R = alpha / prod R (stableswap); and the warm start of the evaluation tiles' root search (the iterate theta the previous
    // evaluation ended on; NaN = none), rewritten by every evaluation outside the reproducible mode
    const double *ifee, *sR;
    double *ws;
};
__global__ void __launch_bounds__(256) gk_derive_kernel(BucketG b, int K, double *ifee, double *sR, double *ws)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= b.m) return;
    ifee[i] = 1.0 / b.fee[i];
    if (b.param) { double pr = 1.0; for (int j = 0; j < K; ++j) pr *= b.R[i * K + j]; sR[i] = b.param[i] / pr; } else sR[i] = 0.0;
    ws[i] = __builtin_nan("");
}

template <int KIND> struct PhiK;

// ---- n-asset stableswap  phi(x) = sum_j x_j - alpha / prod_j x_j   (the paper's concave form; K = 2 is CFMM_POOL_CURVE2) --------
//      s = alpha / prod x,   phi_j = 1 + s / x_j = F(s / x_j),   F(u) = 1 + u
template <> struct PhiK<0> {
    static constexpr bool SMOOTH = true;
    // log S(x) from the sum of the legs' log-reserves and the prepared parameter (log alpha): log alpha - sum log x_j
    static __host__ __device__ inline double prep(double al) { return log(al); }
    static __host__ __device__ inline double coupling_log(double sum_lx, double lal) { return lal - sum_lx; }
    // log G(q), G = F^-1: the marginal F(s / x) equals q at x = s / G(q).  q <= 1: no reserve is that cheap (-inf: level +inf)
    static __host__ __device__ inline double log_G(double q) { return q > 1.0 ? log(q - 1.0) : -1.7976931348623157e308; }
    // the marginal at the pool's own reserves (the diagonal metric's no-trade prices)
    static __host__ __device__ inline double marginal(double s_over_x) { return 1.0 + s_over_x; }
    // phi(x) - phi(R) from the leg differences and the two coupling values, formed without cancelling the big sums
    static __host__ __device__ inline double value_minus(double sum_dx, double s, double sR) { return sum_dx - (s - sR); }
};

// ---- n-asset constant sum  phi(x) = sum_j x_j,  x >= 0   (arbitrage.py:73-74 with more than two tokens) ---------------------------
// an LP: tender the cheapest token, drain every token worth more than it after the fee (a bang-bang vertex; ties are kinks of
// the dual exactly as in the two-asset case -- an optimum that ends ON one needs the host's active-set loop, which knows
// two-asset pools only: such an instance ends without its certificates and says so)
template <> struct PhiK<1> {
    static constexpr bool SMOOTH = false;
    template <int K> static __host__ __device__ inline void solve(const double (&R)[K], const double (&nu)[K], double g, double, double (&y)[K])
    {
        int lo = 0;
        for (int j = 1; j < K; ++j) if (nu[j] < nu[lo]) lo = j;
        double paid = 0.0;
        for (int j = 0; j < K; ++j) {
            y[j] = 0.0;
            if (j != lo && g * nu[j] > nu[lo]) { y[j] = R[j]; paid += R[j]; }
        }
        y[lo] = -paid / g;
    }
};

// The smooth family in log space: with the marginal a function of s / x alone (phi_j = F_j(s / x_j)), a leg's deposit and
// withdraw levels are  log x = log s - log G(nu_j / (gamma m))  and  log s - log G(nu_j / m):  per value of m two numbers per
// leg (gd >= gw), and log x_j(m, s) = clip(log R_j; log s - gd_j, log s - gw_j) is piecewise linear in log s -- the inner
// search costs a handful of adds and compares per step, no transcendental.  Returns sum_j log x_j and the number of legs
// that move with s (off their reserve).
template <int K>
__host__ __device__ inline double gk_sum_lx(const double (&lR)[K], const double (&gd)[K], const double (&gw)[K], double ls, double (&lx)[K], int &moving)
{
    double t = 0.0;
    moving = 0;
    for (int j = 0; j < K; ++j) {
        const double ld = ls - gd[j], lw = ls - gw[j];
        lx[j] = lR[j] < ld ? ld : (lR[j] > lw ? lw : lR[j]);
        moving += lx[j] != lR[j] ? 1 : 0;
        t += lx[j];
    }
    return t;
}

// y = L - D per leg (negative = tendered, positive = received), as every pool function of pool_math.hpp returns it
template <int KIND, int K>
__host__ __device__ inline void pool_generic_k(const double (&R)[K], const double (&nu)[K], double g, double prm, double (&y)[K])
{
    if constexpr (!PhiK<KIND>::SMOOTH) {
        PhiK<KIND>::template solve<K>(R, nu, g, prm, y);
        return;
    } else {
        constexpr double INF = 1.7976931348623157e308;
        double lR[K], lx[K], gd[K], gw[K];
        double slR = 0.0;
        for (int j = 0; j < K; ++j) { lR[j] = log(R[j]); slR += lR[j]; }
        const double lprm = PhiK<KIND>::prep(prm);
        const double lsR = PhiK<KIND>::coupling_log(slR, lprm), sR = exp(lsR);
        // The bracket of the multiplier comes from the pool's own reserves: leg j stays untouched at x = R iff
        // nu_j / phi_j(R) <= m <= nu_j / (gamma phi_j(R)).  With mA = max_j nu_j / phi_j(R) (from there on no leg is withdrawn)
        // and mB = min_j nu_j / (gamma phi_j(R)) (up to there none is deposited):  mA <= mB is the no-trade band -- nothing to
        // solve --, otherwise phi(x) - phi(R) <= 0 at mB, >= 0 at mA: the root lies in [mB, mA], a few per cent wide for a pool
        // a few per cent off the market (the search used to start from ninety units of log m: ~25 outer steps, now ~8).
        double mA = 0.0, mB = INF;
        for (int j = 0; j < K; ++j) {
            const double fj = PhiK<KIND>::marginal(sR / R[j]);
            mA = fmax(mA, nu[j] / fj); mB = fmin(mB, nu[j] / (g * fj));
        }
        if (mA <= mB) { for (int j = 0; j < K; ++j) y[j] = 0.0; return; }
        double ls_warm = lsR;
        // One outer step: the inner root log s(m) of the piecewise-linear, growing residual
        //     r(ls) = ls - log S(x(m, ls)),        slope 1 + (legs off their reserve)
        // by Newton steps kept inside a bracket (a piecewise-linear function: exact once the step starts on the root's piece;
        // bisection otherwise), then phi(x) - phi(R) there.  Exactly 0 when no leg has left its reserve (the no-trade band);
        // +1 when a deposit level is infinite (m at / beyond its upper end).
        auto phi_gap = [&](double lm) -> double {
            const double im = exp(-lm);
            bool open = false;
            for (int j = 0; j < K; ++j) {
                gd[j] = PhiK<KIND>::log_G(nu[j] * im / g); gw[j] = PhiK<KIND>::log_G(nu[j] * im);
                open |= gd[j] <= -INF;
            }
            if (open) return 1.0;
            double a = lsR - 90.0, b = lsR + 90.0, ls = ls_warm;         // (warm: the previous outer step's root)
            int moving = 0;
            for (int it = 0; it < 100; ++it) {
                const double r = ls - PhiK<KIND>::coupling_log(gk_sum_lx<K>(lR, gd, gw, ls, lx, moving), lprm);
                if (r > 0.0) b = ls; else a = ls;
                if (r == 0.0 || b - a <= 4e-16 * fmax(1.0, fabs(ls))) break;
                double nx = ls - r / (1.0 + moving);
                if (!(nx > a && nx < b)) nx = 0.5 * (a + b);
                if (nx == ls) break;
                ls = nx;
            }
            gk_sum_lx<K>(lR, gd, gw, ls, lx, moving);
            ls_warm = ls;
            if (moving == 0) return 0.0;
            double dx = 0.0;
            for (int j = 0; j < K; ++j) dx += lx[j] == lR[j] ? 0.0 : exp(lx[j]) - R[j];
            return PhiK<KIND>::value_minus(dx, exp(ls), sR);
        };
        // outer root in log m of the growing phi_gap.  The bracket [mB, mA] is checked (and widened, doubling, where the other
        // legs' moves have pushed the root outside it); then false position with the Illinois correction, every fourth step
        // a bisection.  phi(x) = phi(R) met to a few dozen roundings of its own sums is the root (every leg's exp carries an
        // ulp or two: asking for less sends the stragglers through dozens of bisections -- 2 ms per launch for a handful of lanes).
        double Rsum = 0.0, mmax = nu[0];
        for (int j = 0; j < K; ++j) { Rsum += R[j]; mmax = fmin(mmax, nu[j]); }
        const double ftol = 8e-15 * Rsum, ltop = log(mmax / g);          // (beyond ltop a leg's deposit level is infinite)
        double lo = log(mB), hi = fmin(log(mA), ltop), w = fmax(hi - lo, 1e-6);
        double flo = phi_gap(lo), fhi = 0.0;
        bool done = fabs(flo) <= ftol, khi = false;
        if (done) hi = lo;
        for (int it = 0; it < 60 && !done && flo > 0.0; ++it) {          // (the root is below: walk down)
            hi = lo; fhi = flo; khi = true;
            lo -= w; w *= 2.0;
            flo = phi_gap(lo);
            if (fabs(flo) <= ftol) { hi = lo; done = true; }
        }
        if (!done && !khi) {
            fhi = phi_gap(hi);
            if (fabs(fhi) <= ftol) done = true;
            for (int it = 0; it < 60 && !done && fhi < 0.0; ++it) {      // (the root is above: walk up, to the open end at most)
                lo = hi; flo = fhi;
                hi = fmin(hi + w, ltop); w *= 2.0;
                fhi = phi_gap(hi);
                if (fabs(fhi) <= ftol) done = true;
            }
        }
        int side = 0;
        for (int it = 0; it < 120 && !done; ++it) {
            if (hi - lo <= 4e-16 * fmax(1.0, fabs(hi))) break;
            double lm = 0.5 * (lo + hi);
            if (fhi != 1.0 && (it & 3) != 3) {                                   // (1.0: the open end's marker, not a value)
                const double t = (lo * fhi - hi * flo) / (fhi - flo);
                if (t > lo && t < hi) lm = t;
            }
            const double f = phi_gap(lm);
            if (fabs(f) <= ftol) { hi = lm; done = true; break; }
            if (f < 0.0) {
                lo = lm; flo = f;
                if (side == -1 && fhi != 1.0) fhi *= 0.5;
                side = -1;
            } else {
                hi = lm; fhi = f;
                if (side == 1) flo *= 0.5;
                side = 1;
            }
        }
        // (the upper end: phi(x) >= phi(R) holds there, and inside the no-trade band -- where the level set is met over a
        //  whole interval of m -- every leg sits exactly on its reserve: lx_j == lR_j, y_j = 0)
        phi_gap(hi);
        for (int j = 0; j < K; ++j) {
            const double d = lx[j] == lR[j] ? 0.0 : R[j] - exp(lx[j]);
            y[j] = d > 0.0 ? d : d / g;
        }
    }
}

// =====================================================================================================================================
// Round 5: the table's pools as WAVE-TILES of the evaluation.  Round 4's kernel solved one pool per lane through
// stride-K loads, libm log / exp inside a nested bracket search of up to 100 x 120 steps, and K + 1 global atomics per pool: 0.4-0.8 ms
// per evaluation for a few thousand pools.  Now: LEG PER LANE like the geometric-mean tiles (K consecutive lanes = one pool, 64 / K
// pools per wave-tile, every column load coalesced), psi through the workgroup's LDS tile and the ordinary flush, and for the
// stableswap entry a search that knows the function:
//
//   * INNER equation  s prod_j x_j(m, s) = alpha  with  x_j = clip(R_j; s / Gd_j, s / Gw_j),  Gw_j = nu_j / m - 1,  Gd_j = nu_j / (gamma m) - 1:
//     h(s) = s prod x_j grows with s and is piecewise a power of s with kinks at R_j Gw_j (below: leg j withdrawn) and R_j Gd_j (above:
//     deposited).  As in tilen every lane evaluates h at its OWN two kinks (K terms each, in the reserves' own units: products,
//     no logarithm) and learns its leg's side of the root; on the root's piece  s^(1 + nA) = alpha prod_A G_j / prod_U R_j  is closed form.
//     No iteration.
//   * OUTER equation  Phi(m) = sum_j (x_j - R_j) - (s - s_R) = 0  (the level set phi(x) = phi(R) along the inner solution).  Phi has
//     poles at m = nu_j / c_j just above the root (a near-peg pool trades until its marginal prices meet the fee-adjusted market: x_j =
//     s / G_j with G_j small), and Newton in m crawls away from a pole by halving (27 steps measured in the NumPy prototype).  In
//     theta = log(m_top / m - 1), m_top = min_j nu_j / gamma the first pole, with the residual in logarithmic form
//     F = log((Phi + B) / B),  B = sum_A R_j - s_R  (Phi + B = s (sum_A 1 / G_j - 1) is a product of powers of the gaps), F is close
//     to LINEAR: safeguarded Newton on F(theta) takes 5.2 evaluations on average from a cold start (maximum 6 over the prototype's
//     3000 random pools of 2..8 assets, fees 0.99..1, prices 0.3 %..20 % off), 2-3 from the previous evaluation's root, which the
//     tiles keep per pool (BucketG::ws, 8 B; not in the reproducible mode, where a pool's result must be a function of the pool and
//     the prices alone).
// The diagonal metric of a solve's first evaluation is the closed-form diagonal of the pool's Hessian block at its own no-trade
// prices (the same block the second-order path assembles: table_newton_kernel) instead of K + 1 perturbed solves.
// pool_generic_k above stays as the function-agnostic reference: cfmm_selftest checks the fast search against it on the device.
// =====================================================================================================================================
constexpr int GT_THREADS = 512;           // 8 waves per workgroup (the search keeps ~60 doubles per lane alive: 256-VGPR budget)
constexpr int GT_STRIP = 128;             // double2 per wave: two 64-entry exchange strips

// ---- one pool, serially (tenders, the second-order path's blocks, the self-test): the same search as the tiles' ----------------------
struct StableSol { double m, s, i1; int nA; };         // multiplier, coupling s = alpha / prod x, 1 / (1 + active legs)
// evaluation at the multiplier m: sides (+1 withdrawn, -1 deposited, 0 untouched), new reserves, Phi, d Phi / d log m, B.  false: m is
// beyond an open end (a deposit level would be infinite)
template <int K>
__host__ __device__ inline bool stable_eval_serial(const double (&R)[K], const double (&nu)[K], double ifee, double al, double sR, double m,
                                                   double (&x)[K], int (&side)[K], double &Phi, double &dPhi, double &B, StableSol &sol)
{
    double Gw[K], Gd[K], iGw[K], iGd[K], qw[K];
    const double im = 1.0 / m;
    for (int j = 0; j < K; ++j) {
        qw[j] = nu[j] * im; Gw[j] = qw[j] - 1.0; Gd[j] = qw[j] * ifee - 1.0;
        if (!(Gd[j] > 0.0)) return false;
        iGw[j] = Gw[j] > 0.0 ? 1.0 / Gw[j] : 1e300; iGd[j] = 1.0 / Gd[j];
    }
    double F = al; int nA = 0;
    for (int j = 0; j < K; ++j) {
        const double Kw = Gw[j] > 0.0 ? R[j] * Gw[j] : 0.0, Kd = R[j] * Gd[j];
        double P1 = Kw, P2 = Kd;
        for (int k = 0; k < K; ++k) { P1 *= fmin(fmax(R[k], Kw * iGd[k]), Kw * iGw[k]); P2 *= fmin(fmax(R[k], Kd * iGd[k]), Kd * iGw[k]); }
        side[j] = (Gw[j] > 0.0 && P1 > al) ? 1 : (P2 < al ? -1 : 0);
        F *= side[j] > 0 ? Gw[j] : (side[j] < 0 ? Gd[j] : 1.0 / R[j]);
        nA += side[j] != 0;
    }
    const double i1 = 1.0 / (1.0 + nA), s = exp(log(F) * i1);
    double Sdx = 0.0, SQ = 0.0, SxQ = 0.0, SxA = 0.0;
    for (int j = 0; j < K; ++j) {
        x[j] = side[j] > 0 ? s * iGw[j] : (side[j] < 0 ? s * iGd[j] : R[j]);
        const double Q = side[j] > 0 ? qw[j] * iGw[j] : (side[j] < 0 ? qw[j] * ifee * iGd[j] : 0.0);
        if (side[j]) { Sdx += x[j] - R[j]; SQ += Q; SxQ += x[j] * Q; SxA += x[j]; }
    }
    Phi = Sdx - (s - sR);
    dPhi = SxQ - (SxA - s) * SQ * i1;
    B = (SxA - Sdx) - sR;
    sol.m = m; sol.s = s; sol.i1 = i1; sol.nA = nA;
    return true;
}
// the search; false = the pool does not trade (x = R).  theta0: a warm start (NaN: none)
template <int K>
__host__ __device__ inline bool pool_stable_k(const double (&R)[K], const double (&nu)[K], double g, double al, double (&x)[K], int (&side)[K],
                                              StableSol &sol, double theta0 = __builtin_nan(""))
{
    const double INF = __builtin_inf(), ifee = 1.0 / g;
    double pr = 1.0, Rsum = 0.0;
    for (int j = 0; j < K; ++j) { pr *= R[j]; Rsum += R[j]; x[j] = R[j]; side[j] = 0; }
    const double sR = al / pr;
    double mA = 0.0, mB = INF, mtop = INF;
    for (int j = 0; j < K; ++j) { const double pj = nu[j] / (1.0 + sR / R[j]); mA = fmax(mA, pj); mB = fmin(mB, pj * ifee); mtop = fmin(mtop, nu[j] * ifee); }
    if (mA <= mB) return false;
    double th = theta0 == theta0 ? theta0 : log(mtop / fmin(mB, mtop * (1.0 - 1e-3)) - 1.0);
    double lo = -INF, hi = INF;
    const double ftol = 1e-9 * Rsum;
    double xn[K]; int sn[K];
    for (int it = 0; it < 64; ++it) {
        const double e = exp(th), m = mtop / (1.0 + e);
        double Phi, dPhi, B;
        if (!stable_eval_serial<K>(R, nu, ifee, al, sR, m, xn, sn, Phi, dPhi, B, sol)) { lo = th; th = hi < INF ? 0.5 * (lo + hi) : th + 1.0; continue; }
        for (int j = 0; j < K; ++j) { x[j] = xn[j]; side[j] = sn[j]; }
        if (fabs(Phi) <= ftol && dPhi > 0.0) {                 // the last Newton step on the solution itself (see tileg_stable)
            const double step = -Phi / dPhi;
            double SQ = 0.0, Q[K];
            for (int j = 0; j < K; ++j) { const double qj = nu[j] / m * (side[j] < 0 ? ifee : 1.0); Q[j] = side[j] ? qj / (qj - 1.0) : 0.0; SQ += Q[j]; }
            for (int j = 0; j < K; ++j) if (side[j]) x[j] = xn[j] * (1.0 + (Q[j] - SQ * sol.i1) * step);
            sol.m = m * (1.0 + step); sol.s = sol.s * (1.0 - SQ * sol.i1 * step);
            break;
        }
        if (Phi > 0.0) lo = th; else hi = th;
        if (lo > -INF && hi < INF && hi - lo <= 1e-15 * fmax(1.0, fabs(hi))) break;
        const double dth = -dPhi * e / (1.0 + e);            // d Phi / d theta  (log m = log m_top - log(1 + e^theta))
        double tn = __builtin_nan("");
        if (dth < 0.0) tn = (B > 0.0 && Phi + B > 0.0) ? th - log1p(Phi / B) * (Phi + B) / dth : th - Phi / dth;
        if (!(tn > lo && tn < hi)) tn = (lo > -INF && hi < INF) ? 0.5 * (lo + hi) : (Phi > 0.0 ? th + 1.0 : th - 1.0);
        if (fabs(tn - th) <= 1e-15 * fmax(1.0, fabs(th))) break;
        th = tn;
    }
    return true;
}
// y = L - D per leg from the new reserves (the interface of pool_generic_k)
template <int KIND, int K>
__host__ __device__ inline void pool_table_k(const double (&R)[K], const double (&nu)[K], double g, double prm, double (&y)[K])
{
    if constexpr (KIND == 0) {
        double x[K]; int side[K]; StableSol sol;
        pool_stable_k<K>(R, nu, g, prm, x, side, sol);
        for (int j = 0; j < K; ++j) y[j] = side[j] > 0 ? R[j] - x[j] : (side[j] < 0 ? (R[j] - x[j]) / g : 0.0);
    } else pool_generic_k<KIND, K>(R, nu, g, prm, y);
}

// ---- a wave-tile of the stableswap entry: 64 / K pools, leg per lane ---------------------------------------------------------------
// xs: this wave's GT_STRIP double2 of LDS.  warm: read / write the bucket's warm-start column
// NEWT (the second-order path's evaluation, table_newton_kernel below): 1 = the tenders carry the first-order response to the low-order
// log-prices `nw.slo_s` (smooth.hpp: apply_slo; null: none) and the pool's value p'y is summed into nw.vsum; 2 = also the pool's exact
// Hessian block in log-prices (stable_block above), leg pair (j, k <= j) by lane j, through global atomics into nw.H
struct TileNewt { const double *slo_s; double *H; int ldh; double *vsum; };
template <int K, bool WITH_D, bool DET, int NEWT = 0>
__device__ __forceinline__ void tileg_stable(const BucketG &b, int tb, int lane, const double *nu_s, const Scatter<DET> &psi_s, const Scatter<DET> &diag_s,
                                             double2 *xs, bool warm, double ftol_rel, const TileNewt &nw = TileNewt{nullptr, nullptr, 0, nullptr})
{
    static_assert(!NEWT || (!WITH_D && !DET), "the second-order tile: plain accumulation, no metric");
    constexpr int P = 64 / K;
    constexpr double INF = 1.7976931348623157e308;
    // (opaque copy, as in kernels.hpp: tilen -- otherwise lane / K, lane % K and the strip addresses of all seven instantiations are hoisted
    //  out of the tile loop and stay live across it: table_newton_kernel<true> sat at 256 VGPRs WITH three dwords spilled, round 5)
#ifndef CFMM_NO_OPAQUE_LANE
    asm volatile("" : "+v"(lane));
#endif
    const int g = lane / K, j = lane - g * K;
    const unsigned pool = (unsigned)tb * P + g;
    const bool live = g < P && pool < (unsigned long long)b.m;
    const unsigned leg = live ? pool * K + j : 0u, pl = live ? pool : 0u;
    const int tok = b.idx[leg];
    const double R = b.R[leg], ifee = b.ifee[pl], al = b.param[pl], sR = b.sR[pl];
    const double wsv = (!DET && warm) ? b.ws[pl] : __builtin_nan("");
    const double nu = nu_s[tok];
    const int gb = (g < P ? g : 0) * K;
    double2 *xt = xs + 64;
    const double iR = rcp_nr(R), phiR = fma(sR, iR, 1.0);
    const double pj = nu * rcp_nr(phiR);
    // the pool's no-trade band [mB, mA] in the multiplier, its first pole m_top, the sum of its reserves
    xs[lane] = make_double2(pj, nu); xt[lane] = make_double2(R, R * phiR);
    __builtin_amdgcn_wave_barrier();
    double mA = 0.0, mnp = INF, mnn = INF, Rsum = 0.0, Rq = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) { const double2 q = xs[gb + k], r = xt[gb + k]; mA = fmax(mA, q.x); mnp = fmin(mnp, q.x); mnn = fmin(mnn, q.y); Rsum += r.x; Rq += r.y; }
    __builtin_amdgcn_wave_barrier();
    const double mB = mnp * ifee, mtop = mnn * ifee;
    if constexpr (WITH_D) {
        // the closed-form diagonal of the Hessian block at the pool's own no-trade prices q_j = phi_j(R), no fee (all K legs active,
        // x = R, m = 1, s = s_R):  B_jj = (q_j^2 / s_R) (N_jj - u_j^2 / (q' u)),  N = diag(R^2) - R R' / (1 + K),  u = N q;  the metric
        // entry is nu_j B_jj / q_j (the pool's share of -d y_j / d log nu_j, priced at the trial prices like every other bucket's)
        const double u = R * R * phiR - R * Rq * (1.0 / (1.0 + K));
        xs[lane] = make_double2(phiR * u, 0.0);
        __builtin_amdgcn_wave_barrier();
        double qu = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) qu += xs[gb + k].x;
        __builtin_amdgcn_wave_barrier();
        const double dj = nu * phiR * (R * R * (K / (1.0 + K)) - u * u * rcp_nr(qu)) * rcp_nr(sR);
        if (live && dj > 0.0 && dj < INF) diag_s.add(tok, dj);
    }
    const bool trade = live && mA > mB;
    if (!__any(trade)) return;
    const double imtop = rcp_nr(mtop);
    double th = wsv == wsv ? wsv : log_pos(mtop * rcp_nr(fmin(mB, mtop * (1.0 - 1e-3))) - 1.0);
    double lo = -INF, hi = INF, x = R;
    double m_f = 0.0, s_f = 0.0, i1_f = 1.0, nA_f = 0.0;      // NEWT: multiplier, coupling s = alpha / prod x, 1 / (1 + active legs), active legs -- of the point x is at
    bool done = !trade, W = false, D = false;
    const double ftol = ftol_rel * Rsum;
    for (int it = 0; it < 64; ++it) {
        if (!__any(!done)) break;
        const double e = exp(th), im = (1.0 + e) * imtop;
        const double qw = nu * im, qd = qw * ifee, Gw = qw - 1.0, Gd = qd - 1.0;
        const bool okw = Gw > 0.0, opn = !(Gd > 0.0);
        const double iGw = okw ? rcp_nr(Gw) : 1e300, iGd = rcp_nr(opn ? 1.0 : Gd);
        const double Kw = okw ? R * Gw : 0.0, Kd = R * Gd;
        xs[lane] = make_double2(R, iGd); xt[lane] = make_double2(iGw, opn ? 1.0 : 0.0);
        __builtin_amdgcn_wave_barrier();
        double P1 = Kw, P2 = Kd, nopen = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double2 q = xs[gb + k], r = xt[gb + k];
            P1 *= fmin(fmax(q.x, Kw * q.y), Kw * r.x);
            P2 *= fmin(fmax(q.x, Kd * q.y), Kd * r.x);
            nopen += r.y;
        }
        __builtin_amdgcn_wave_barrier();
        const bool Wn = okw && P1 > al, Dn = !Wn && P2 < al, act = Wn || Dn;
        xs[lane] = make_double2(Wn ? Gw : (Dn ? Gd : iR), act ? 1.0 : 0.0);
        __builtin_amdgcn_wave_barrier();
        double F = al, nA = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) { const double2 q = xs[gb + k]; F *= q.x; nA += q.y; }
        __builtin_amdgcn_wave_barrier();
        const double i1 = rcp_nr(1.0 + nA);
        const double s = exp(log_pos(fmax(F, 1e-300)) * i1);
        const double xn = Wn ? s * iGw : (Dn ? s * iGd : R);
        const double Q = Wn ? qw * iGw : (Dn ? qd * iGd : 0.0);
        xs[lane] = make_double2(act ? xn - R : 0.0, Q); xt[lane] = make_double2(xn * Q, act ? xn : 0.0);
        __builtin_amdgcn_wave_barrier();
        double Sdx = 0.0, SQ = 0.0, SxQ = 0.0, SxA = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) { const double2 q = xs[gb + k], r = xt[gb + k]; Sdx += q.x; SQ += q.y; SxQ += r.x; SxA += r.y; }
        __builtin_amdgcn_wave_barrier();
        if (!done) {
            if (nopen > 0.0) {                           // beyond an open end: Phi = +inf there
                lo = th; th = hi < INF ? 0.5 * (lo + hi) : th + 1.0;
            } else {
                const double Phi = Sdx - (s - sR);
                const double dlm = SxQ - (SxA - s) * SQ * i1;              // d Phi / d log m  (> 0)
                x = xn; W = Wn; D = Dn;
                if constexpr (NEWT != 0) { m_f = rcp_nr(im); s_f = s; i1_f = i1; nA_f = nA; }
                if (fabs(Phi) <= ftol && dlm > 0.0) {
                    // close enough for the LAST Newton step to be taken on the solution itself instead of re-evaluated (the iteration
                    // converges quadratically: the step from |Phi| <= 1e-9 sum R lands below the rounding of x): d log x_j = (Q_j - SQ / (1 + nA)) d log m
                    // on the active legs.  One evaluation less per pool, and tenders accurate to rounding -- which the outer iterations need:
                    // stopping at |Phi| <= 1e-13 sum R left 3e-8 of noise in the dual value of 1000 pools, above the last Newton decrements
                    const double step = -Phi * rcp_nr(dlm);
                    if (act) x = xn * fma(Q - SQ * i1, step, 1.0);
                    if constexpr (NEWT != 0) { m_f *= 1.0 + step; s_f *= 1.0 - SQ * i1 * step; }      // (pool_stable_k's sol after its last step)
                    th -= step * (1.0 + e) * rcp_nr(e);
                    done = true;
                } else {
                    if (Phi > 0.0) lo = th; else hi = th;
                    if (lo > -INF && hi < INF && hi - lo <= 1e-15 * fmax(1.0, fabs(hi))) done = true;
                    else {
                        const double dth = -dlm * e * rcp_nr(1.0 + e);
                        const double B = (SxA - Sdx) - sR;
                        double tn = __builtin_nan("");
                        if (dth < 0.0) tn = (B > 0.0 && Phi + B > 0.0) ? th - log1p_wave<true>(Phi * rcp_nr(B)) * (Phi + B) * rcp_nr(dth) : th - Phi * rcp_nr(dth);
                        if (!(tn > lo && tn < hi)) tn = (lo > -INF && hi < INF) ? 0.5 * (lo + hi) : (Phi > 0.0 ? th + 1.0 : th - 1.0);
                        if (fabs(tn - th) <= 1e-15 * fmax(1.0, fabs(th))) done = true; else th = tn;
                    }
                }
            }
        }
    }
    if (!DET && warm && trade && j == 0) b.ws[pl] = th;
    if constexpr (NEWT != 0) {
        // the pool of the second-order path, leg per lane: pools that trade on at least two legs enter with tenders, value and block; the others
        // not at all (a single active leg cannot move along the level set)
        const bool emit = trade && nA_f > 1.5;
        const bool act = emit && (W || D);
        const double pk = act ? (D ? nu * ifee : nu) : 0.0;
        xs[lane] = make_double2(x * pk, 0.0);
        __builtin_amdgcn_wave_barrier();
        double xp = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) xp += xs[gb + k].x;
        __builtin_amdgcn_wave_barrier();
        const double v = act ? x * x * pk - x * xp * i1_f : 0.0;
        const double sl = (nw.slo_s && act) ? nw.slo_s[tok] : 0.0;
        xs[lane] = make_double2(pk * v, x * pk * sl); xt[lane] = make_double2(v * pk * sl, 0.0);
        __builtin_amdgcn_wave_barrier();
        double pv = 0.0, xps = 0.0, vps = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) { const double2 q = xs[gb + k]; pv += q.x; xps += q.y; vps += xt[gb + k].x; }
        __builtin_amdgcn_wave_barrier();
        const double coef = emit ? rcp_nr(m_f * s_f) : 0.0, ipv = pv > 0.0 ? rcp_nr(pv) : 0.0;
        if (act) {
            double y = D ? (R - x) * ifee : R - x;
            if (nw.slo_s) y += coef * pk * (x * x * pk * sl - x * i1_f * xps - v * vps * ipv) * rcp_nr(nu);
            psi_s.add(tok, y);
            *nw.vsum += nu * y;
        }
        if constexpr (NEWT == 2) {
            xs[lane] = make_double2(x, pk); xt[lane] = make_double2(v, __hiloint2double(0, tok));
            __builtin_amdgcn_wave_barrier();
            if (act) {
                for (int k = 0; k <= j; ++k) {
                    const double2 q = xs[gb + k], r = xt[gb + k];
                    if (q.y == 0.0) continue;              // (leg k does not trade)
                    const int tk = __double2loint(r.y);
                    const double Njk = (k == j ? x * x : 0.0) - x * q.x * i1_f;
                    const double hv = coef * pk * q.y * (Njk - v * r.x * ipv);
                    const int row = tok > tk ? tok : tk, col = tok > tk ? tk : tok;
                    unsafeAtomicAdd(&nw.H[(size_t)col * nw.ldh + row], hv);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    if (live) {
        const double y = W ? R - x : (D ? (R - x) * ifee : 0.0);
        if (y != 0.0) psi_s.add(tok, y);
    }
}

// ---- a wave-tile of the constant-sum entry (arbitrage.py:73-74 over K tokens): the LP vertex, leg per lane -------------------------------
// tender the cheapest token, drain every token worth more than it after the fee.  flags (per leg, or null): legs the host's
// active-set loop has TIED to the pool's cheapest token (gamma nu_j = nu_lo: a kink of the dual) are left out here -- the host
// adds their partial fill (cfmm/problem.py)
template <int K, bool DET>
__device__ __forceinline__ void tileg_sum(const BucketG &b, const int *flags, int tb, int lane, const double *nu_s, const Scatter<DET> &psi_s, double2 *xs)
{
    constexpr int P = 64 / K;
    const int g = lane / K, j = lane - g * K;
    const unsigned pool = (unsigned)tb * P + g;
    const bool live = g < P && pool < (unsigned long long)b.m;
    const unsigned leg = live ? pool * K + j : 0u, pl = live ? pool : 0u;
    const int tok = b.idx[leg];
    const double R = b.R[leg], fee = b.fee[pl], ifee = b.ifee[pl];
    const bool tied = flags && flags[leg] != 0;
    const double nu = nu_s[tok];
    const int gb = (g < P ? g : 0) * K;
    xs[lane] = make_double2(nu, 0.0);
    __builtin_amdgcn_wave_barrier();
    double mn = 1.7976931348623157e308; int lo = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) { const double v = xs[gb + k].x; if (v < mn) { mn = v; lo = k; } }      // (ties: the first, as PhiK<1>::solve)
    __builtin_amdgcn_wave_barrier();
    const double yj = (j != lo && !tied && fee * nu > mn) ? R : 0.0;
    xs[lane] = make_double2(yj, 0.0);
    __builtin_amdgcn_wave_barrier();
    double paid = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) paid += xs[gb + k].x;
    __builtin_amdgcn_wave_barrier();
    const double y = j == lo ? -paid * ifee : yj;
    if (live && y != 0.0) psi_s.add(tok, y);
}

// ---- the table's evaluation launch: every bucket of the table in ONE launch, behind the main evaluation ----------------------------
struct TableArgs {
    BucketG bs[7], bq[7];             // stableswap / constant-sum buckets of 2..8 assets (index k - 2)
    const int *qflags[7];             // per-leg tie flags of the constant-sum buckets (or null)
    int tile_end[14];                 // cumulative wave-tiles: stableswap 2..8, then constant sum 2..8
    int ntiles, n, nslices, warm;
    const double *nu;                 // [n + 1]: prices, then the stop flag
    double *acc;
    unsigned long long *acc_l;        // reproducible mode: the limbs (kernels.hpp: Scatter<true>)
    double det_scale, det_scale_d;
    double ftol;                      // the stableswap search takes its last Newton step on the solution itself once |phi(x) - phi(R)| <= ftol x the pool's reserves
};
__host__ __device__ inline size_t table_lds_bytes(int n, bool with_d, bool det, int waves)
{
    return (size_t)(((with_d ? 2 : 1) * eval_tile_doubles(n, det) + n + 2 + 16 + 2 + 1) & ~1) * sizeof(double) + (size_t)waves * GT_STRIP * sizeof(double2);
}
#ifndef GT_WAVES_PER_SIMD
#define GT_WAVES_PER_SIMD 2
#endif
template <bool WITH_D, bool DET>
__global__ void __launch_bounds__(GT_THREADS, GT_WAVES_PER_SIMD)
table_eval_kernel(TableArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = a.n, tile = eval_tile_doubles(n, DET);
    double *psi_t = lds, *diag_t = lds + tile;
    double *nu_s = lds + (WITH_D ? 2 : 1) * tile;      // [n + 1]
    double *fpart = nu_s + n + 2;                      // [16]
    int *next_tile = reinterpret_cast<int *>(fpart + 16);
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    double2 *xs = reinterpret_cast<double2 *>(lds + ((((WITH_D ? 2 : 1) * tile + n + 2 + 16 + 2 + 1) & ~1))) + GT_STRIP * wib;
    for (int j = threadIdx.x; j <= n; j += blockDim.x) nu_s[j] = a.nu[j];
    for (int j = threadIdx.x; j < (WITH_D ? 2 : 1) * tile; j += blockDim.x) lds[j] = 0.0;
    // this workgroup's contiguous share of the tile space; its waves draw from an LDS ticket counter
    const int t0 = (int)(((long long)blockIdx.x * a.ntiles) / gridDim.x), t1 = (int)(((long long)(blockIdx.x + 1) * a.ntiles) / gridDim.x);
    if (threadIdx.x == 0) *next_tile = t0 + nw;
    __syncthreads();
    if (nu_s[n] != 0.0) return;
    const Scatter<DET> psi_s{psi_t, n, a.det_scale}, diag_s{diag_t, n, a.det_scale_d};
    int ticket = t0 + wib;
    for (;;) {
        const int t = __builtin_amdgcn_readfirstlane(ticket);
        if (t >= t1) break;
        if (lane == 0) ticket = atomicAdd(next_tile, 1);
        int q = 0, first = 0;                            // (13 scalar compares against the kernel arguments: constant indices, no copy of the table)
#pragma unroll
        for (int i = 0; i < 13; ++i) if (t >= a.tile_end[i]) { q = i + 1; first = a.tile_end[i]; }
        const int tb = t - first;
        switch (q) {
#define GT_S(KK) case KK - 2: tileg_stable<KK, WITH_D, DET>(a.bs[KK - 2], tb, lane, nu_s, psi_s, diag_s, xs, a.warm != 0, a.ftol); break;
#define GT_Q(KK) case 7 + KK - 2: tileg_sum<KK, DET>(a.bq[KK - 2], a.qflags[KK - 2], tb, lane, nu_s, psi_s, xs); break;
        GT_S(2) GT_S(3) GT_S(4) GT_S(5) GT_S(6) GT_S(7) GT_S(8)
        GT_Q(2) GT_Q(3) GT_Q(4) GT_Q(5) GT_Q(6) GT_Q(7) GT_Q(8)
#undef GT_S
#undef GT_Q
        default: break;
        }
    }
    __syncthreads();
    // the flush: as eval_tiles_and_flush's (kernels.hpp)
    if constexpr (DET) {
        const unsigned long long *pl = reinterpret_cast<const unsigned long long *>(psi_t), *dl = reinterpret_cast<const unsigned long long *>(diag_t);
        for (int j = threadIdx.x; j < 3 * n; j += blockDim.x) {
            const unsigned long long v = pl[j];
            if (v) atomicAdd(&a.acc_l[j], v);
            if (WITH_D) { const unsigned long long dv = dl[j]; if (dv) atomicAdd(&a.acc_l[3 * n + j], dv); }
        }
        return;
    }
    double *base = a.acc + (size_t)(blockIdx.x % a.nslices) * acc_stride(n);
    double fw = 0.0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const double v = psi_t[j];
        if (v != 0.0) { unsafeAtomicAdd(&base[j], v); fw = fma(nu_s[j], v, fw); }
        if (WITH_D) { const double dv = diag_t[j]; if (dv != 0.0) unsafeAtomicAdd(&base[acc_diag(n) + j], dv); }
    }
    fw = wave_sum(fw);
    if (lane == 0) fpart[wib] = fw;
    __syncthreads();
    if (threadIdx.x == 0) {
        double f = 0.0;
        for (int w = 0; w < nw; ++w) f += fpart[w];
        if (f != 0.0) unsafeAtomicAdd(&base[acc_arb(n)], f);
    }
}

// ---- the table's stableswap buckets inside the second-order path: ONE launch, the same wave-tiles (round 5) -------------------------------
// (Round 5's first form, one pool per lane and one launch per bucket, served as the reference the tiles were tested against entry by entry;
//  round 6 pinned the tiles' Hessian block against finite differences of the NumPy restatement instead -- tests/test_gpu_table.py -- and dropped it.)  Exact
// tenders + first-order response to the low-order log-prices into out[0 .. n), the pools' value into out[n] and out[n + 1], HESS: the
// exact K x K blocks into H.  LDS: psi tile | nu_s[n] | slo_s[n] | wave partials | ticket | strips.
__host__ __device__ inline size_t table_newton_lds_bytes(int n, int waves)
{
    return (size_t)((3 * n + 16 + 2 + 1) & ~1) * sizeof(double) + (size_t)waves * GT_STRIP * sizeof(double2);
}
template <bool HESS>
__global__ void __launch_bounds__(GT_THREADS, GT_WAVES_PER_SIMD)
table_newton_kernel(TableArgs a, const double *__restrict__ slo, double *__restrict__ out, double *__restrict__ H, int ldh)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = a.n;
    double *psi_t = lds, *nu_s = lds + n, *slo_s = lds + 2 * n, *fpart = lds + 3 * n;
    int *next_tile = reinterpret_cast<int *>(fpart + 16);
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    double2 *xs = reinterpret_cast<double2 *>(lds + ((3 * n + 16 + 2 + 1) & ~1)) + GT_STRIP * wib;
    for (int j = threadIdx.x; j < n; j += blockDim.x) { nu_s[j] = a.nu[j]; psi_t[j] = 0.0; if (slo) slo_s[j] = slo[j]; }
    const int ntiles = a.tile_end[6];                     // the stableswap buckets' share of the tile space
    const int t0 = (int)(((long long)blockIdx.x * ntiles) / gridDim.x), t1 = (int)(((long long)(blockIdx.x + 1) * ntiles) / gridDim.x);
    if (threadIdx.x == 0) *next_tile = t0 + nw;
    __syncthreads();
    const Scatter<false> psi_s{psi_t, n, 0.0};
    double vsum = 0.0;
    const TileNewt tn{slo ? slo_s : nullptr, H, ldh, &vsum};
    int ticket = t0 + wib;
    for (;;) {
        const int t = __builtin_amdgcn_readfirstlane(ticket);
        if (t >= t1) break;
        if (lane == 0) ticket = atomicAdd(next_tile, 1);
        int q = 0, first = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) if (t >= a.tile_end[i]) { q = i + 1; first = a.tile_end[i]; }
        const int tb = t - first;
        switch (q) {
#define GT_N(KK) case KK - 2: tileg_stable<KK, false, false, HESS ? 2 : 1>(a.bs[KK - 2], tb, lane, nu_s, psi_s, psi_s, xs, a.warm != 0, a.ftol, tn); break;
        GT_N(2) GT_N(3) GT_N(4) GT_N(5) GT_N(6) GT_N(7) GT_N(8)
#undef GT_N
        default: break;
        }
    }
    vsum = wave_sum(vsum);
    if (lane == 0) fpart[wib] = vsum;
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const double v = psi_t[j];
        if (v != 0.0) unsafeAtomicAdd(&out[j], v);
    }
    if (threadIdx.x == 0) {
        double f = 0.0;
        for (int w = 0; w < nw; ++w) f += fpart[w];
        if (f != 0.0) { unsafeAtomicAdd(&out[n], f); unsafeAtomicAdd(&out[n + 1], f); }
    }
}

// ---- the stableswap entry in the second-order path (CFMM_METHOD_NEWTON) --------------------------------------------
// Like the K-asset geometric-mean pools these are strictly curved and enter UNSMOOTHED: exact solution + exact generalised Hessian
// block.  With p_j = nu_j (withdrawn) / nu_j / gamma (deposited) on the active legs A, the KKT system  m grad phi_A(x) = p_A,
// phi(x) = phi(R)  differentiates to  dx_A = (m H)^-1 (dp_A - dm grad phi_A),  grad phi_A' dx_A = 0,  and the Hessian of phi on A is
// H = -s (diag(1 / x^2) + (1/x)(1/x)'),  (m H)^-1 = -(1 / (m s)) N,  N = diag(x^2) - x x' / (1 + nA)  (Sherman-Morrison).  The block of
// the dual's Hessian in log-prices (the term delta_jk nu_j y_j aside, which the caller's diagonal carries) is
//     B_jk = nu_j nu_k d y_j / d nu_k = (p_j p_k / (m s)) (N_jk - v_j v_k / (p' v)),   v = N p
// -- symmetric, positive semidefinite, its null vector the common scaling of the prices (checked against finite differences of
// the solver in the NumPy prototype: 1e-9).  Low-order log-prices (smooth.hpp: s_lo) move the tenders by dy_j = sum_k B_jk s_lo,k / nu_j.
template <int K>
__device__ __forceinline__ void stable_block(const double (&x)[K], const double (&p)[K], const int (&side)[K], double g, const StableSol &sol,
                                             double (&pk)[K], double (&v)[K], double &coef, double &ipv)
{
    double xp = 0.0;
    for (int j = 0; j < K; ++j) { pk[j] = side[j] ? (side[j] < 0 ? p[j] / g : p[j]) : 0.0; xp += x[j] * pk[j]; }
    double pv = 0.0;
    for (int j = 0; j < K; ++j) { v[j] = side[j] ? x[j] * x[j] * pk[j] - x[j] * xp * sol.i1 : 0.0; pv += pk[j] * v[j]; }
    coef = 1.0 / (sol.m * sol.s);
    ipv = pv > 0.0 ? 1.0 / pv : 0.0;
}
// dy_j = sum_k B_jk s_k / nu_j for the low-order log-prices s (by leg)
template <int K>
__device__ __forceinline__ void stable_slo(const double (&x)[K], const double (&p)[K], const int (&side)[K], double g, const StableSol &sol,
                                           const double (&sl)[K], double (&dy)[K])
{
    double pk[K], v[K], coef, ipv;
    stable_block<K>(x, p, side, g, sol, pk, v, coef, ipv);
    double xps = 0.0, vps = 0.0;
    for (int k = 0; k < K; ++k) { xps += x[k] * pk[k] * sl[k]; vps += v[k] * pk[k] * sl[k]; }
    for (int j = 0; j < K; ++j) dy[j] = side[j] ? coef * pk[j] * (x[j] * x[j] * pk[j] * sl[j] - x[j] * sol.i1 * xps - v[j] * vps * ipv) / p[j] : 0.0;
}
// ---- the constant-sum entry in the second-order path: barrier-smoothed like the two-asset pools (round 5) ------------------------------
// arbitrage.py:73-74 over K tokens is an LP per pool; its kinks in the dual -- a leg drained only partly, two tokens tied for cheapest --
// are what the first-order path's active-set loop chases (cfmm/problem.py) and, on small networks whose tokens differ in value, often
// does not catch (tools/fuzz_table.py).  The second-order path puts the path's own log barrier (weight mu) on the LP's sign constraints,
// as smooth.hpp does for the two-asset pools:
//     arb_mu(p) = max  p'(L - D) + mu sum_j [log L_j + log(R_j - L_j)] + mu sum_k log D_k    s.t.  gamma sum D = sum L
// (receipts L_j in (0, R_j), payments D_k > 0; the pool's constraint sum x >= sum R holds with equality, x = R + gamma D - L > 0).  With
// the multiplier tau of the constraint everything separates:  D_k = mu / (p_k - gamma tau)  (tau < min p / gamma),  L_j the root in
// (0, R_j) of  (p_j - tau) + mu / L - mu / (R_j - L) = 0  -- closed form -- and tau is the root of the increasing
// F(tau) = gamma sum D - sum L: a safeguarded Newton iteration on ONE scalar per pool, carried as the GAP  delta = min p / gamma - tau > 0
// (the cheapest token's payment is mu / (gamma delta): with tau itself that difference would lose every digit at the small weights the
// path ends on).  The optimum is interior, so arb_mu is smooth, its gradient y = L - D is a strictly FEASIBLE tender for every mu
// (partly drained legs and split payments come out of it without an active set), arb - p'y <= 3 K mu (one mu per barrier term: the pool
// counts 3 K of them in the path's gap bound), and the Hessian in prices is  M = diag(d) - c c' / F',  d_j = kappa_j + D_j^2 / mu,
// c_j = kappa_j + gamma D_j^2 / mu,  kappa_j = 1 / (mu / L_j^2 + mu / (R_j - L_j)^2),  F' = sum (kappa_j + gamma^2 D_j^2 / mu).
// (A first version smoothed in PRICE space -- softplus per leg, soft minimum for the payer -- was feasible and convex too, but not
// self-concordant: Newton's quadratic model held only within mu of a kink, and the path crawled: 200 steps on mid-size networks.)
template <int K> struct SumSmooth { double L[K], D[K], kap[K], d[K], c[K], val, trade, Fp; };
template <int K>
__host__ __device__ inline void sum_smooth_k(const double (&R)[K], const double (&p)[K], double g, double mu, SumSmooth<K> &o)
{
    double pmin = p[0], Rs = 0.0;
    for (int j = 0; j < K; ++j) { pmin = fmin(pmin, p[j]); Rs += R[j]; }
    const double tmax = pmin / g, ig = 1.0 / g;
    // at the gap delta:  p_k - gamma tau = (p_k - pmin) + gamma delta,   a_j = p_j - tau = (p_j - tmax) + delta
    auto eval = [&](double dl, double &F, double &Fp) {
        F = 0.0; Fp = 0.0;
        for (int j = 0; j < K; ++j) {
            const double D = mu / ((p[j] - pmin) + g * dl);
            const double a = (p[j] - tmax) + dl, aR = a * R[j], S = sqrt(aR * aR + 4.0 * mu * mu);
            const double L = a > 0.0 ? (aR - 2.0 * mu + S) / (2.0 * a) : 2.0 * mu * R[j] / (S + 2.0 * mu - aR);
            const double Lc = fmin(fmax(L, 1e-300), R[j] * (1.0 - 1e-16));
            const double kap = 1.0 / (mu / (Lc * Lc) + mu / ((R[j] - Lc) * (R[j] - Lc)));
            o.L[j] = Lc; o.D[j] = D; o.kap[j] = kap;
            F += g * D - Lc; Fp += kap + g * g * D * D / mu;
        }
    };
    // F decreases in delta (F(0+) = +inf, F(inf) = -sum R): Newton in log delta, bracketed
    double lo = 0.0, hi = 1.7976931348623157e308, dl = 2.0 * K * mu / (g * Rs), F, Fp;
    for (int it = 0; it < 200; ++it) {
        eval(dl, F, Fp);
        if (F > 0.0) lo = dl; else hi = dl;
        if (fabs(F) <= 1e-13 * Rs) break;
        // d F / d delta = -F' (tau = tmax - delta); the step in log delta keeps delta > 0
        double dn = dl * exp(fmin(fmax(F / (Fp * dl), -3.0), 3.0));
        if (!(dn > lo && dn < hi)) dn = hi < 1e308 ? (lo > 0.0 ? sqrt(lo * hi) : 0.5 * hi) : 8.0 * dl;
        if (fabs(dn - dl) <= 1e-15 * dl) break;
        dl = dn;
    }
    eval(dl, F, Fp);
    (void)ig;
    double val = 0.0, tr = 0.0;
    for (int j = 0; j < K; ++j) {
        o.d[j] = o.kap[j] + o.D[j] * o.D[j] / mu; o.c[j] = o.kap[j] + g * o.D[j] * o.D[j] / mu;
        const double y = o.L[j] - o.D[j];
        tr += p[j] * y;
        val += p[j] * y + mu * (log(o.L[j]) + log(R[j] - o.L[j]) + log(o.D[j]));
    }
    o.val = val; o.trade = tr; o.Fp = Fp;
}
// M_jk of the pool (price space)
template <int K>
__host__ __device__ inline double sum_smooth_hess(const SumSmooth<K> &o, double g, int j, int k)
{
    (void)g;
    return (j == k ? o.d[j] : 0.0) - o.c[j] * o.c[k] / o.Fp;
}
template <int K, bool HESS>
__global__ void __launch_bounds__(256)
gk_sum_newton_kernel(BucketG b, const double *__restrict__ nu, const double *__restrict__ slo, double mu, double *__restrict__ out, int n,
                     double *__restrict__ H, int ldh)
{
    double vsum = 0.0, tsum = 0.0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < b.m; i += (long long)gridDim.x * blockDim.x) {
        double R[K], p[K];
        int tok[K];
        for (int j = 0; j < K; ++j) { tok[j] = b.idx[i * K + j]; R[j] = b.R[i * K + j]; p[j] = nu[tok[j]]; }
        const double g = b.fee[i];
        SumSmooth<K> o;
        sum_smooth_k<K>(R, p, g, mu, o);
        for (int j = 0; j < K; ++j) {
            double y = o.L[j] - o.D[j];
            if (slo) for (int k = 0; k < K; ++k) y += sum_smooth_hess<K>(o, g, j, k) * p[k] * slo[tok[k]];      // first-order response to the low-order log-prices
            unsafeAtomicAdd(&out[tok[j]], y); tsum += p[j] * y;
        }
        vsum += o.val;
        if (HESS) {
            for (int j = 0; j < K; ++j)
                for (int k = 0; k <= j; ++k) {
                    const double h = p[j] * p[k] * sum_smooth_hess<K>(o, g, j, k);
                    const int row = tok[j] > tok[k] ? tok[j] : tok[k], col = tok[j] > tok[k] ? tok[k] : tok[j];
                    unsafeAtomicAdd(&H[(size_t)col * ldh + row], h);
                }
        }
    }
    vsum = wave_allsum(vsum); tsum = wave_allsum(tsum);
    if ((threadIdx.x & 63) == 0) { if (vsum != 0.0) unsafeAtomicAdd(&out[n], vsum); if (tsum != 0.0) unsafeAtomicAdd(&out[n + 1], tsum); }
}

// tenders at the accepted prices, slot-major [K][m] as the C-ABI hands them out (two-asset.py:94,98); flags: as tileg_sum;
// slo: the low-order log-prices a second-order solve ended with (the tenders must be those of the same point as psi)
template <int KIND, int K>
__global__ void __launch_bounds__(GK_THREADS)
tradesg_kernel(BucketG b, const int *flags, const double *__restrict__ nu, const double *__restrict__ slo, double mu, double *__restrict__ delta, double *__restrict__ lambda)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.m) return;
    double R[K], p[K], y[K];
    for (int j = 0; j < K; ++j) { R[j] = b.R[i * K + j]; p[j] = nu[b.idx[i * K + j]]; }
    if (KIND == 1 && mu > 0.0) {
        // behind a second-order solve: the smoothed pool's own tenders (gk_sum_newton_kernel), GROSS -- a leg may both receive its
        // R_j sigma_j and pay its share w_j P / gamma (arbitrage.py:51-52 allows both; netting them would claim a fee that was not paid)
        const double g = b.fee[i];
        SumSmooth<K> o;
        sum_smooth_k<K>(R, p, g, mu, o);
        // first-order response to the low-order log-prices, of the receipts and of the payments SEPARATELY (their difference is the
        // M dp of gk_sum_newton_kernel; split like this the pool's constraint gamma sum dD = sum dL holds to rounding):
        // dtau = c'dp / F',  dL_j = kappa_j (dp_j - dtau),  dD_k = -(D_k^2 / mu) (dp_k - gamma dtau)
        double dp[K], dtau = 0.0;
        for (int k = 0; k < K; ++k) { dp[k] = slo ? p[k] * slo[b.idx[i * K + k]] : 0.0; dtau += o.c[k] * dp[k]; }
        dtau /= o.Fp;
        for (int j = 0; j < K; ++j) {
            const double dl = o.L[j] + o.kap[j] * (dp[j] - dtau);
            const double dd = o.D[j] - o.D[j] * o.D[j] / mu * (dp[j] - g * dtau);
            delta[(size_t)j * b.m + i] = fmax(dd, 0.0); lambda[(size_t)j * b.m + i] = fmax(dl, 0.0);
        }
        return;
    }
    if (KIND == 0 && slo) {
        double x[K], sl[K], dy[K]; int side[K]; StableSol sol;
        const double g = b.fee[i];
        const bool tr = pool_stable_k<K>(R, p, g, b.param[i], x, side, sol) && sol.nA >= 2;
        for (int j = 0; j < K; ++j) { sl[j] = slo[b.idx[i * K + j]]; dy[j] = 0.0; }
        if (tr) stable_slo<K>(x, p, side, g, sol, sl, dy);
        for (int j = 0; j < K; ++j) y[j] = (tr && side[j]) ? (side[j] > 0 ? R[j] - x[j] : (R[j] - x[j]) / g) + dy[j] : 0.0;
    } else
    pool_table_k<KIND, K>(R, p, b.fee[i], b.param ? b.param[i] : 0.0, y);
    if (KIND == 1 && flags) {                          // tied legs: their fill is the host's (theta R_j, paid for by the cheapest token)
        int lo = 0;
        for (int j = 1; j < K; ++j) if (p[j] < p[lo]) lo = j;
        for (int j = 0; j < K; ++j) if (j != lo && flags[i * K + j] && y[j] != 0.0) { y[lo] += y[j] / b.fee[i]; y[j] = 0.0; }
    }
    for (int j = 0; j < K; ++j) {
        delta[(size_t)j * b.m + i] = fmax(-y[j], 0.0);
        lambda[(size_t)j * b.m + i] = fmax(y[j], 0.0);
    }
}

// self-test (cfmm_selftest): the stableswap entry's own search against the function-agnostic two-level search above, on the device
__global__ void __launch_bounds__(64)
selftest_table_kernel(int *out)
{
    int bad = 0;
    unsigned long long st = 0xD1B54A32D192ED03ull * (threadIdx.x + 1);
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) * (1.0 / 9007199254740992.0); };
    for (int rep = 0; rep < 6; ++rep) {
        double R[4], nu[4], ya[4], yb[4];
        const double val = exp(6.0 * rnd() + 1.0), g = 1.0 - 0.01 * rnd(), off = 0.002 + 0.1 * rnd() * rnd();
        double pr = 1.0, mean = 0.0;
        for (int j = 0; j < 4; ++j) { const double price = exp(0.6 * rnd() - 0.3); R[j] = val / price * exp(0.1 * rnd() - 0.05); nu[j] = price * exp(off * (2.0 * rnd() - 1.0)); pr *= R[j]; mean += 0.25 * R[j]; }
        const double al = pr * mean / (5.0 + 400.0 * rnd());
        pool_table_k<0, 4>(R, nu, g, al, ya);
        pool_generic_k<0, 4>(R, nu, g, al, yb);
        for (int j = 0; j < 4; ++j) if (!(fabs(ya[j] - yb[j]) <= 1e-10 * (R[0] + R[1] + R[2] + R[3]))) { ++bad; break; }
    }
    atomicAdd(out, bad);
}

}  // namespace cfmm
