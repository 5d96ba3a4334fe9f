// Per-pool arbitrage subproblems, fp64, device side.  gfx950 only.
//
// For local prices p > 0 each pool independently solves
//     arb_i(p) = max p'(L - D)  s.t.  phi_i(R + gamma D - L) >= phi_i(R),  D, L >= 0
// i.e. one pool's share of the reference model: variables /root/reference/arbitrage.py:51-52,
// post-trade reserves :60, trading-function constraint :63-74.  Every function returns
// y = L - D per leg (negative = tendered, positive = received); arb = p'y.
#pragma once
#include <hip/hip_runtime.h>

namespace cfmm {


struct Y2 { double ya, yb; };

// scheduling fence: keeps the compiler from interleaving independent library-routine expansions (log, expm1, ...),
// which multiplies their temporaries; the evaluation kernel is register-budget sensitive
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// 1/x and 1/sqrt(x) from the quarter-rate hardware seeds plus two Newton steps (<= ~1 ulp for
// normal positive inputs: reserves, prices, fees); ~6-8 instructions instead of the 11-20 of the
// IEEE sequences, which matters because the evaluation kernel is fp64-issue bound
__device__ __forceinline__ double rcp_nr(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double rsqrt_nr(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = fma(0.5 * y, fma(-x * y, y, 1.0), y);
    y = fma(0.5 * y, fma(-x * y, y, 1.0), y);
    return y;
}

// log x for positive, finite x (reserves x prices / weights): argument reduction x = m 2^e with m in [sqrt(1/2), sqrt(2)) from
// the hardware's frexp pair, then log m = log(1 + f) through s = f / (2 + f) and the classical degree-14 minimax polynomial in
// s (the scheme of Sun's fdlibm e_log; error < 1 ulp with an exact division, ~1 ulp with rcp_nr).  ~36 fp64-pipe
// instructions against the ~95 of the library routine, which also handles zeros, negatives, infinities and NaNs -- a
// quarter of a K-asset wave-tile's instructions went into that one call.
__device__ __forceinline__ double log_pos(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);                     // [0.5, 1)  (subnormals included)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double f = m - 1.0;
    const double s = f * rcp_nr(2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return fma(dk, 6.93147180369123816490e-01, -((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
}

// log1p and expm1 for the small arguments that dominate here (a pool a few per cent off the market):
// short series, taken only when EVERY active lane of the wave is in range (wave-uniform branch, no
// divergence); otherwise the library routines.  Relative error < 2e-16 inside the ranges below.
// PURE = true (reproducible mode): the choice is made per lane, so that a pool's result is a function of the pool alone and
// not of which other pools share its wave (a different sharding of the pools must reproduce every contribution bit for bit)
template <bool PURE = false>
__device__ __forceinline__ double log1p_wave(double d)        // log(1 + d)
{
    if (PURE ? (fabs(d) < 0.125) : (bool)__all(fabs(d) < 0.125)) {
        const double s = d * rcp_nr(2.0 + d), z = s * s;           // log(1+d) = 2 atanh(d / (2 + d))
        double p = 1.0 / 13.0;
        p = fma(p, z, 1.0 / 11.0); p = fma(p, z, 1.0 / 9.0); p = fma(p, z, 1.0 / 7.0);
        p = fma(p, z, 1.0 / 5.0); p = fma(p, z, 1.0 / 3.0);
        return 2.0 * fma(s * z, p, s);
    }
    return log1p(d);
}
template <bool PURE = false>
__device__ __forceinline__ double expm1_wave(double x)
{
    if (PURE ? (fabs(x) < 0.0625) : (bool)__all(fabs(x) < 0.0625)) {  // Taylor to x^11 / 11!
        double p = 1.0 / 39916800.0;
        p = fma(p, x, 1.0 / 3628800.0); p = fma(p, x, 1.0 / 362880.0); p = fma(p, x, 1.0 / 40320.0);
        p = fma(p, x, 1.0 / 5040.0); p = fma(p, x, 1.0 / 720.0); p = fma(p, x, 1.0 / 120.0);
        p = fma(p, x, 1.0 / 24.0); p = fma(p, x, 1.0 / 6.0); p = fma(p, x, 0.5);
        return fma(x * x, p, x);
    }
    return expm1(x);
}

// Constant product sqrt(xy)  (Uniswap v2: arbitrage.py:68-70, equal-weight cp.geo_mean).
// With v = p R (the value of a reserve) the pool tenders `in` for `out` iff
//     rho = gamma v_out / v_in > 1,
// the new reserve of the tendered token is R_in sqrt(rho), so
//     y_in = -R_in (sqrt(rho) - 1) / gamma,     y_out = R_out (1 - 1/sqrt(rho)).
// One rsqrt serves both; gamma <= 1 makes the two directions exclusive.
__device__ __forceinline__ Y2 pool_cp2(double Ra, double Rb, double g, double pa, double pb)
{
    const double va = pa * Ra, vb = pb * Rb;
    const bool ab = g * vb > va;
    const bool ba = g * va > vb;
    const double vin = ab ? va : vb, vout = ab ? vb : va;
    const double Rin = ab ? Ra : Rb, Rout = ab ? Rb : Ra;
    const double rho = g * vout * rcp_nr(vin);
    const double r = rsqrt_nr(rho);
    const double yin = -Rin * fma(rho, r, -1.0) * rcp_nr(g);
    const double yout = Rout * (1.0 - r);
    Y2 y;
    y.ya = ab ? yin : (ba ? yout : 0.0);
    y.yb = ab ? yout : (ba ? yin : 0.0);
    return y;
}

// The same solution in directed form, for the evaluation tiles: which way the pool trades, y of the tendered and of the
// received token, and the pool's arbitrage profit  nu_in y_in + nu_out y_out = v_out (1 - 1/sqrt(rho)) - v_in (sqrt(rho) - 1) / gamma
// straight from the values (the tile scatters y_in / y_out to the tokens it picks with `ab`: ten selects fewer per pool
// than building (y_a, y_b) and their price-weighted sum).
struct Y2dir { double yin, yout, arb; bool ab, active; };
__device__ __forceinline__ Y2dir pool_cp2_dir(double Ra, double Rb, double g, double pa, double pb)
{
    const double va = pa * Ra, vb = pb * Rb;
    Y2dir o;
    o.ab = g * vb > va;
    o.active = o.ab || (g * va > vb);
    const double vin = o.ab ? va : vb, vout = o.ab ? vb : va;
    const double Rin = o.ab ? Ra : Rb, Rout = o.ab ? Rb : Ra;
    const double rho = g * vout * rcp_nr(vin);
    const double r = rsqrt_nr(rho);
    const double X = fma(rho, r, -1.0), ig = rcp_nr(g);         // sqrt(rho) - 1,  1 / gamma
    const double om = 1.0 - r;
    o.yin = -Rin * X * ig;                                      // (the same operations in the same order as pool_cp2: the
    o.yout = Rout * om;                                         //  reproducible mode compares the two forms bit for bit)
    o.arb = fma(vout, om, -vin * X * ig);
    return o;
}

// Weighted geometric mean x^wa y^(1-wa) (2-asset Balancer: arbitrage.py:65 with two tokens).
// With weighted values va = w_b p_a R_a, vb = w_a p_b R_b the pool tenders `in` iff
//     rho = gamma v_out / v_in > 1,   and with L = log rho:
//     x / R_in = rho^{w_out}:  y_in = -R_in expm1(w_out L) / gamma,   y_out = -R_out expm1(-w_in L).
// One log and two expm1, all on the well-conditioned quantity L (no x - R_in cancellation).
template <bool PURE = false>
__device__ __forceinline__ Y2 pool_w2(double Ra, double Rb, double g, double wa, double pa, double pb)
{
    const double wb = 1.0 - wa;
    const double va = wb * pa * Ra, vb = wa * pb * Rb;
    const bool ab = g * vb > va;
    const bool ba = g * va > vb;
    Y2 y; y.ya = 0.0; y.yb = 0.0;
    if (ab | ba) {
        const double vin = ab ? va : vb, vout = ab ? vb : va;
        const double Rin = ab ? Ra : Rb, Rout = ab ? Rb : Ra;
        const double win = ab ? wa : wb, wout = ab ? wb : wa;
        // (scheduling fences: interleaving the three library routines costs ~30 VGPRs and the kernel sits on a register cliff)
        const double L = log1p_wave<PURE>(fma(g, vout, -vin) * rcp_nr(vin));      // log rho; rho - 1 formed without cancellation
        SCHED_FENCE();
        const double yin = -Rin * expm1_wave<PURE>(wout * L) * rcp_nr(g);
        SCHED_FENCE();
        const double yout = -Rout * expm1_wave<PURE>(-win * L);
        SCHED_FENCE();
        y.ya = ab ? yin : yout;
        y.yb = ab ? yout : yin;
    }
    return y;
}

// Constant sum x + y with x, y >= 0 (arbitrage.py:73-74): bang-bang LP.
__device__ __forceinline__ Y2 pool_sum2(double Ra, double Rb, double g, double pa, double pb)
{
    Y2 r; r.ya = 0.0; r.yb = 0.0;
    if (g * pb > pa)      { r.ya = -Rb / g; r.yb = Rb; }
    else if (g * pa > pb) { r.yb = -Ra / g; r.ya = Ra; }
    return r;
}

// Curve-style x + y - alpha/(xy) (not in the reference; BASELINE config 5).  y(x) on the level
// set is the positive root of x y^2 + (x^2 - C x) y - alpha = 0; the optimum equates the
// marginal price m = phi_x/phi_y with p_in/(gamma p_out): safeguarded Newton in x.
// (reciprocals / square roots from the hardware seeds, rcp_nr / rsqrt_nr: the IEEE sequences made this bucket 224 us
//  for 5e5 pools)
__device__ __forceinline__ double sqrt_nr(double v) { return v > 0.0 ? v * rsqrt_nr(v) : 0.0; }

__device__ __forceinline__ double curve_y(double x, double ix, double C, double al)
{
    const double b = C - x, q = 4.0 * al * ix;
    const double sq = sqrt_nr(fma(b, b, q));
    return b >= 0.0 ? 0.5 * (b + sq) : 0.5 * q * rcp_nr(sq - b);        // (no cancellation past the knee, x > C)
}

// marginal price m(x) = phi_x / phi_y on the level set minus rho, and its derivative along the curve
__device__ __forceinline__ void curve_price(double x, double C, double al, double rho, double &h, double &dh, double &y)
{
    const double ix = rcp_nr(x);
    y = curve_y(x, ix, C, al);
    const double iy = rcp_nr(y);
    const double t = al * ix * iy;                       // alpha / (x y)
    const double fx = fma(t, ix, 1.0), fy = fma(t, iy, 1.0);
    const double ify = rcp_nr(fy);
    const double m = fx * ify;
    h = m - rho;
    const double dfx = -t * ix * fma(-iy, m, 2.0 * ix);  // y' = -m
    const double dfy = -t * iy * fma(-2.0 * iy, m, ix);
    dh = (dfx * fy - fx * dfy) * ify * ify;
}

__device__ __forceinline__ bool curve_dir(double Rin, double Rout, double g, double al, double C,
                                          double pin, double pout, double &yin, double &yout)
{
    const double rho = pin * rcp_nr(g * pout);
    {   // the direction test at the pool's own reserves: y = Rout there, no curve solve
        const double ix = rcp_nr(Rin), iy = rcp_nr(Rout), t = al * ix * iy;
        if (!(fma(t, ix, 1.0) * rcp_nr(fma(t, iy, 1.0)) - rho > 0.0)) return false;
    }
    // Safeguarded Newton in x on [lo, hi), hi unknown until an iterate lands beyond the root (doubling while it is).  Start:
    // where the marginal price has dropped to rho on the imbalanced branch, 1 - m ~ alpha rho / (x y^2) with x ~ C - y
    // (three fixed-point sweeps: a few per cent off the root of the trades that end past the knee, 5-6 steps from there);
    // from x = Rin the flat part of the curve sends the first step far past the knee and the bracket then closes by
    // bisection (12-16 steps) -- 134 -> 7x us for 5e5 pools 1 % off their peg.
    double lo = Rin, hi = 1.7976931348623157e308, x = Rin;
    if (rho < 1.0) {
        const double k = al * rho * rcp_nr(1.0 - rho);
        double ye = sqrt_nr(k * rcp_nr(C));
        ye = sqrt_nr(k * rcp_nr(C - ye));
        ye = sqrt_nr(k * rcp_nr(C - ye));
        const double x0 = C - ye;
        if (x0 > Rin && x0 < 1e300) x = x0;
    }
    double h, dh, y;
    for (int it = 0; it < 200; ++it) {
        curve_price(x, C, al, rho, h, dh, y);
        if (h > 0.0) lo = x; else hi = x;
        // (the price condition met to rounding ends the search: m and rho carry ~4 ulp each, below that the sign of h is noise.
        //  Near its peg the curve is flat -- dh x / rho ~ 1e-3 .. 1e-5 -- so x is only defined to 1e-13 .. 1e-11 relative: without
        //  this test such a lane jitters at that level, never meets the step test below, and is finished by ~50 bisection steps
        //  of the bracket, with its whole wave waiting)
        if (fabs(h) <= 1e-15 * rho) break;
        double xn = x - h * rcp_nr(dh);
        if (!(xn > lo && xn < hi)) xn = hi < 1e308 ? 0.5 * (lo + hi) : 2.0 * lo;
        const bool done = fabs(xn - x) <= 2e-15 * x;
        x = xn;
        if (done) break;
        SCHED_FENCE();
    }
    yin = -(x - Rin) * rcp_nr(g);
    yout = Rout - curve_y(x, rcp_nr(x), C, al);
    return true;
}

__device__ __forceinline__ Y2 pool_curve2(double Ra, double Rb, double g, double al, double pa, double pb)
{
    const double C = Ra + Rb - al / (Ra * Rb);
    Y2 r; r.ya = 0.0; r.yb = 0.0;
    if (curve_dir(Ra, Rb, g, al, C, pa, pb, r.ya, r.yb)) return r;
    curve_dir(Rb, Ra, g, al, C, pb, pa, r.yb, r.ya);
    return r;
}

// d y_k / d log p_k at the no-trade point: the pool's share of the static diagonal metric
__device__ __forceinline__ void curve_diag(double Ra, double Rb, double al, double pa, double pb,
                                           double &da, double &db)
{
    const double x = Ra, yy = Rb;
    const double fx = 1.0 + al / (x * x * yy), fy = 1.0 + al / (x * yy * yy);
    const double yp = -fx / fy;
    const double dfx = -2.0 * al / (x * x * x * yy) - al / (x * x * yy * yy) * yp;
    const double dfy = -al / (x * x * yy * yy) - 2.0 * al / (x * yy * yy * yy) * yp;
    const double dm = (dfx * fy - fx * dfy) / (fy * fy);
    const double dxdl = (fx / fy) / fabs(dm);
    da = pa * dxdl;
    db = pb * dxdl * (fx / fy);
}

// K-asset weighted geometric mean (Balancer: arbitrage.py:65, liquidation.py:65, two-asset.py:74).
// KKT: x_j(mu) = clip(R_j, mu gamma w_j/p_j, mu w_j/p_j) with sum_j w_j log x_j = sum_j w_j log R_j.
// In t = log mu, a_j = log(R_j p_j / w_j), lg = log gamma the residual
//     F(t) = sum_j w_j f(t - a_j),   f(u) = u (u<0) | 0 (0<=u<=-lg) | u + lg (u>-lg)
// is piecewise linear and non-decreasing: evaluate it at its 2K breakpoints, keep the bracketing
// pair, interpolate.  O(K^2) flops, K logs, 1 exp, no sort, no data-dependent loop.
template <int K, class PriceOf>
__device__ __forceinline__ void pool_geomean_n(const double (&R)[K], const double (&w)[K], double g,
                                               PriceOf price, double (&y)[K])
{
    double a[K];
    const double lg = log(g);
#pragma unroll
    for (int j = 0; j < K; ++j) { a[j] = log(R[j] * price(j) / w[j]); SCHED_FENCE(); }
    double tL = -1.7976931348623157e308, fL = 0.0, tR = 1.7976931348623157e308, fR = 0.0;
#pragma unroll
    for (int b = 0; b < 2 * K; ++b) {
        const double t = (b < K) ? a[b % K] : a[b % K] - lg;
        double f = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const double u = t - a[j];
            f += w[j] * (u < 0.0 ? u : (u > -lg ? u + lg : 0.0));
        }
        if (f <= 0.0 && t > tL) { tL = t; fL = f; }
        if (f >= 0.0 && t < tR) { tR = t; fR = f; }
        SCHED_FENCE();
    }
    double t;
    if (fL == 0.0) t = tL;
    else if (fR == 0.0) t = tR;
    else t = tL - fL * (tR - tL) / (fR - fL);
    const double mu = exp(t);
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const double hi = mu * w[j] / price(j), lo = g * hi;
        const double x = R[j] < lo ? lo : (R[j] > hi ? hi : R[j]);
        y[j] = (x < R[j]) ? (R[j] - x) : (R[j] - x) / g;
        SCHED_FENCE();
    }
}

}  // namespace cfmm
