// The trading-function table: a two-asset trading function is ONE struct here, on BOTH paths.
//
//   * the second-order path (smooth.hpp) reaches a function only through Phi2<KIND>: the barrier-smoothed per-direction
//     solve, its Hessian term and the interior tenders are written once, against fwd / marginal0 / ratio / level / start;
//   * the first-order path (kernels.hpp: tile2, trades2_kernel) evaluates the constant-product, weighted and stableswap
//     buckets through their hand-tuned closed forms (pool_math.hpp) -- and EVERY OTHER KIND through pool_generic2<KIND>
//     below: the exact arbitrage subproblem  max nu_out L(D) - nu_in D  over D >= 0  solved by a safeguarded Newton
//     iteration on  A(D) = nu_out L'(D) - nu_in  (decreasing: L is concave), from the same fwd.  Its diagonal-metric
//     entry comes from fwd(0) too.  A new function needs no closed form, no new tile code and no new upload case: its
//     pools ride in the generic two-asset bucket (columns Ra, Rb, fee, param, ia, ib: include/cfmm.h).
//
// Shipped through that route end to end: CFMM_POOL_POW2, the power-sum invariant  x^(1-t) + y^(1-t)  (YieldSpace's
// constant-power-sum curve: t -> 0 is the constant sum of arbitrage.py:73-74, t -> 1 approaches the constant product of
// arbitrage.py:68-70).  cfmm_selftest runs pool_generic2 on the three closed-form kinds as well and compares.
//                                          reference: arbitrage.py:63-74 ("a pool is whatever constraint line is written")
#pragma once
#include "pool_math.hpp"

namespace cfmm {

struct Fwd { double L, L1, L2; };       // L(D), L'(D), L''(D)

// (reciprocals and square roots through rcp_nr / rsqrt_nr, pool_math.hpp: <= ~1 ulp, a third of the
//  instructions of the IEEE sequences -- this path is fp64-issue bound like the exact evaluation)

__device__ __forceinline__ double curve_y_stable(double x, double ix, double C, double al)
{
    const double b = C - x, q = 4.0 * al * ix;
    const double sq = sqrt_nr(fma(b, b, q));
    return b >= 0.0 ? 0.5 * (b + sq) : 0.5 * q * rcp_nr(sq - b);
}

// ---- the trading-function table of the second-order path ---------------------------------------
// A two-asset trading function enters the smoothed evaluation, its Hessian and the interior tenders ONLY through the five
// members below: a new function is one more Phi2<KIND> (plus its bucket in the upload layer and its closed form -- or a
// generic root search on the same L' -- in pool_math.hpp for the exact first-order evaluation).
//   fwd(D, Rin, Rout, g, r, C)   L(D), L'(D), L''(D) of the forward exchange function L(D) = R_out - Y(R_in + g D) on the
//                                pool's level set (arbitrage.py:60,63-74), formed without cancellation
//   marginal0(Rin, Rout, g, r)   L'(0), without a curve solve
//   ratio(prm, a_to_b)           the direction's parameter r from the pool's stored parameter
//   level(Ra, Rb, prm)           a per-pool constant C handed to fwd (0 where the function needs none)
//   start(...)                   a starting tender on the trade side: the exact mu = 0 root where it is closed form, an
//                                estimate otherwise, <= 0 for "none" (the barrier model at D = 0 is used then)
template <int KIND> struct Phi2;

template <> struct Phi2<0> {                       // constant product (arbitrage.py:68-70)
    static __device__ __forceinline__ Fwd fwd(double D, double Rin, double Rout, double g, double, double)
    {
        Fwd o;
        const double ix = rcp_nr(fma(g, D, Rin));
        const double gy = g * Rout * ix;                   // L = gamma D R_out / x  (no cancellation)
        o.L = D * gy;
        o.L1 = gy * Rin * ix;                              // gamma k / x^2
        o.L2 = -2.0 * g * o.L1 * ix;
        return o;
    }
    static __device__ __forceinline__ double marginal0(double Rin, double Rout, double g, double) { return g * Rout * rcp_nr(Rin); }
    static __device__ __forceinline__ double ratio(double, bool) { return 0.0; }
    static __device__ __forceinline__ double level(double, double, double) { return 0.0; }
    static __device__ __forceinline__ double start(double Rin, double Rout, double g, double, double, double ni, double no)
    {
        return (sqrt_nr(g * no * Rin * Rout * rcp_nr(ni)) - Rin) * rcp_nr(g);
    }
};

template <> struct Phi2<1> {                       // weighted geometric mean, r = w_in / w_out (arbitrage.py:65 with two tokens)
    static __device__ __forceinline__ Fwd fwd(double D, double Rin, double Rout, double g, double r, double)
    {
        Fwd o;
        const double ix = rcp_nr(fma(g, D, Rin));
        const double lq = -r * log1p(g * D * rcp_nr(Rin)); // log (R_in / x)^r
        const double q = exp(lq);
        o.L = -Rout * expm1(lq);
        o.L1 = g * Rout * r * q * ix;
        o.L2 = -g * (r + 1.0) * o.L1 * ix;
        return o;
    }
    static __device__ __forceinline__ double marginal0(double Rin, double Rout, double g, double r) { return g * r * Rout * rcp_nr(Rin); }
    static __device__ __forceinline__ double ratio(double wa, bool a_to_b) { return a_to_b ? wa / (1.0 - wa) : (1.0 - wa) / wa; }
    static __device__ __forceinline__ double level(double, double, double) { return 0.0; }
    static __device__ __forceinline__ double start(double Rin, double Rout, double g, double r, double, double ni, double no)
    {
        return Rin * expm1(log(g * no * Rout * r / (ni * Rin)) / (r + 1.0)) / g;
    }
};

template <> struct Phi2<3> {                       // stableswap  x + y - alpha / (x y),  r = alpha, C = the pool's level
    static __device__ __forceinline__ Fwd fwd(double D, double Rin, double Rout, double g, double al, double C)
    {
        Fwd o;
        const double x = fma(g, D, Rin);
        const double ix = rcp_nr(x);
        const double y = curve_y_stable(x, ix, C, al);
        const double iy = rcp_nr(y);
        const double t = al * ix * iy;                     // alpha / (x y)
        const double fx = fma(t, ix, 1.0), fy = fma(t, iy, 1.0);
        const double ify = rcp_nr(fy);
        const double y1 = -fx * ify;
        const double fxx = -2.0 * t * ix * ix, fxy = -t * ix * iy, fyy = -2.0 * t * iy * iy;
        const double y2 = -(fxx + 2.0 * fxy * y1 + fyy * y1 * y1) * ify;
        o.L = Rout - y;
        o.L1 = -g * y1;
        o.L2 = -g * g * y2;
        return o;
    }
    static __device__ __forceinline__ double marginal0(double Rin, double Rout, double g, double al)
    {
        const double t = al * rcp_nr(Rin * Rout);
        return g * fma(t, rcp_nr(Rin), 1.0) * rcp_nr(fma(t, rcp_nr(Rout), 1.0));
    }
    static __device__ __forceinline__ double ratio(double al, bool) { return al; }
    static __device__ __forceinline__ double level(double Ra, double Rb, double al) { return Ra + Rb - al / (Ra * Rb); }
    // where the marginal price m = phi_x / phi_y has dropped to rho = nu_in / (gamma nu_out): for y << x,
    // 1 - m ~ alpha rho / (x y^2) with x ~ C - y (three fixed-point sweeps) -- a few per cent off the root at the 80/20
    // imbalance such trades end at, from where the iteration converges in 5-6 steps (from D = 0 it first overshoots the
    // knee and needs 12-16)
    static __device__ __forceinline__ double start(double Rin, double Rout, double g, double al, double C, double ni, double no)
    {
        if (!(no * marginal0(Rin, Rout, g, al) - ni > 0.0)) return 0.0;       // no-trade side
        const double rho = ni * rcp_nr(g * no);
        if (!(rho < 1.0)) return 0.0;
        const double k = al * rho * rcp_nr(1.0 - rho);
        double y = sqrt_nr(k * rcp_nr(C));
        y = sqrt_nr(k * rcp_nr(C - y));
        y = sqrt_nr(k * rcp_nr(C - y));
        const double D0 = (C - y - Rin) * rcp_nr(g);
        return (D0 > 0.0 && D0 < 1e300) ? D0 : 0.0;
    }
};

template <> struct Phi2<4> {                       // power sum  x^q + y^q,  q = 1 - t,  r = t in (0, 1): no closed form is USED on the exact path
    static __device__ __forceinline__ Fwd fwd(double D, double Rin, double Rout, double g, double t, double)
    {
        Fwd o;
        const double q = 1.0 - t;
        const double x = fma(g, D, Rin);
        const double lx = log1p(g * D / Rin);              // log(x / R_in)
        const double lr = log(Rin / Rout);
        const double z = expm1(q * lx) * exp(q * lr);      // (x^q - R_in^q) / R_out^q: the share of R_out^q paid out
        if (!(z < 1.0)) { o.L = Rout; o.L1 = 0.0; o.L2 = -1e-300; return o; }      // (the pool is drained at a finite tender)
        const double ly = log1p(-z) / q;                   // log(y / R_out)
        const double y = Rout * exp(ly);
        o.L = -Rout * expm1(ly);                           // R_out - y, without cancellation
        o.L1 = g * exp(t * (ly - lr - lx));                // gamma (y / x)^t
        o.L2 = -t * o.L1 * (o.L1 / y + g / x);
        return o;
    }
    static __device__ __forceinline__ double marginal0(double Rin, double Rout, double g, double t) { return g * exp(t * log(Rout / Rin)); }
    static __device__ __forceinline__ double ratio(double t, bool) { return t; }
    static __device__ __forceinline__ double level(double, double, double) { return 0.0; }
    // (the second-order path's cold start only -- pool_generic2 is told not to use it: (y / x)^t = nu_in / (gamma nu_out) on the level set)
    static __device__ __forceinline__ double start(double Rin, double Rout, double g, double t, double, double ni, double no)
    {
        const double q = 1.0 - t, rho = ni / (g * no);
        if (!(no * marginal0(Rin, Rout, g, t) - ni > 0.0)) return 0.0;
        const double K = exp(q * log(Rin)) + exp(q * log(Rout));
        const double x = exp(log(K / (1.0 + exp((q / t) * log(rho)))) / q);
        const double D0 = (x - Rin) / g;
        return (D0 > 0.0 && D0 < 1e300) ? D0 : 0.0;
    }
};

// ---- the generic exact pool: any Phi2<KIND>, no closed form --------------------------------------------------------------
// One direction (tender `in`, receive `out`): trades iff nu_out L'(0) > nu_in; then the root of A(D) = nu_out L'(D) - nu_in.
// Newton on A (A' = nu_out L'' < 0), every step kept inside the bracket [lo: A > 0, hi: A < 0]; while no upper end is
// known a rejected step doubles D; converged when the step is below 2e-15 of D.  USE_START: begin at Phi2::start (the exact
// root where the function has one in closed form) instead of the Newton step from D = 0.
template <int KIND, bool USE_START>
__device__ __forceinline__ bool generic_dir(double Rin, double Rout, double g, double r, double C, double ni, double no,
                                            double &yin, double &yout)
{
    const double A0 = no * Phi2<KIND>::marginal0(Rin, Rout, g, r) - ni;
    if (!(A0 > 0.0)) return false;
    double lo = 0.0, hi = 1.7976931348623157e308;
    double D = USE_START ? Phi2<KIND>::start(Rin, Rout, g, r, C, ni, no) : 0.0;
    if (!(D > 0.0 && D < 1e300)) {
        const Fwd f0 = Phi2<KIND>::fwd(0.0, Rin, Rout, g, r, C);
        D = A0 / fmax(-no * f0.L2, 1e-300);
        if (!(D > 0.0 && D < 1e300)) D = Rin;
    }
    for (int it = 0; it < 200; ++it) {
        const Fwd f = Phi2<KIND>::fwd(D, Rin, Rout, g, r, C);
        const double A = no * f.L1 - ni, A1 = fmin(no * f.L2, -1e-300);
        if (A > 0.0) lo = D; else hi = D;
        if (fabs(A) <= 1e-15 * ni) break;              // (the price condition met to rounding: pool_math.hpp, curve_dir)
        double Dn = D - A / A1;
        if (!(Dn > lo && Dn < hi)) Dn = hi < 1e308 ? 0.5 * (lo + hi) : 2.0 * D;
        const bool done = fabs(Dn - D) <= 2e-15 * fmax(Dn, D);
        D = Dn;
        if (done) break;
        SCHED_FENCE();
    }
    yin = -D;
    yout = Phi2<KIND>::fwd(D, Rin, Rout, g, r, C).L;
    return true;
}

template <int KIND, bool USE_START = false>
__device__ __forceinline__ Y2 pool_generic2(double Ra, double Rb, double g, double prm, double pa, double pb)
{
    const double C = Phi2<KIND>::level(Ra, Rb, prm);
    Y2 y; y.ya = 0.0; y.yb = 0.0;
    if (generic_dir<KIND, USE_START>(Ra, Rb, g, Phi2<KIND>::ratio(prm, true), C, pa, pb, y.ya, y.yb)) return y;
    generic_dir<KIND, USE_START>(Rb, Ra, g, Phi2<KIND>::ratio(prm, false), C, pb, pa, y.yb, y.ya);
    return y;
}

// the pool's share of the static diagonal metric (d (nu_k y_k) / d log nu_k at the no-trade point, fee aside): with the
// pool's own marginal price for the price ratio,  nu_in L'(0) / |L''(0)|  per direction (constant product: nu_a R_a / 2)
template <int KIND>
__device__ __forceinline__ void generic_diag(double Ra, double Rb, double prm, double pa, double pb, double &da, double &db)
{
    const double C = Phi2<KIND>::level(Ra, Rb, prm);
    const Fwd fa = Phi2<KIND>::fwd(0.0, Ra, Rb, 1.0, Phi2<KIND>::ratio(prm, true), C);
    const Fwd fb = Phi2<KIND>::fwd(0.0, Rb, Ra, 1.0, Phi2<KIND>::ratio(prm, false), C);
    da = pa * fa.L1 / fmax(-fa.L2, 1e-300);
    db = pb * fb.L1 / fmax(-fb.L2, 1e-300);
}

}  // namespace cfmm
