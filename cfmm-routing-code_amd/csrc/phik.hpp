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
#include "pool_math.hpp"

namespace cfmm {

constexpr int N_KINDSK = 2;           // CFMM_POOLK_KINDS: 0 stableswap (sum x - alpha / prod x), 1 constant sum
constexpr int GK_THREADS = 256;

struct BucketG {
    long long m;
    const int *idx;                   // pool-major legs: leg j of pool i at [i K + j]
    const double *R;
    const double *fee, *param;        // per pool
};

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

// ---- the kernels: one pool per lane, psi through global atomics into accumulator slice 0 (these buckets are small next to the
// tile space of eval_kernel; their launch follows it on the same stream, like the stableswap bucket's) ---------------------------
// acc layout: kernels.hpp (acc_arb / acc_diag).  nu[n] != 0: the solve has ended, nothing to evaluate.
template <int KIND, int K, bool WITH_D>
__global__ void __launch_bounds__(GK_THREADS)
evalg_kernel(BucketG b, int n, const double *__restrict__ nu, double *__restrict__ acc, int arb_at, int diag_at)
{
    if (nu[n] != 0.0) return;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double arb = 0.0;
    if (i < b.m) {
        double R[K], p[K], y[K];
        int tok[K];
        for (int j = 0; j < K; ++j) { tok[j] = b.idx[i * K + j]; R[j] = b.R[i * K + j]; p[j] = nu[tok[j]]; }
        const double g = b.fee[i], prm = b.param ? b.param[i] : 0.0;
        pool_generic_k<KIND, K>(R, p, g, prm, y);
        for (int j = 0; j < K; ++j) if (y[j] != 0.0) { unsafeAtomicAdd(&acc[tok[j]], y[j]); arb += p[j] * y[j]; }
        if constexpr (WITH_D && PhiK<KIND>::SMOOTH) {
            // the diagonal metric: -d y_j / d log nu_j at the pool's own no-trade prices, fee aside -- by the solver itself: the
            // pool's marginal prices at R (m = 1), one leg's price raised by eps, gamma = 1
            double q[K], yy[K], slR = 0.0;
            for (int j = 0; j < K; ++j) slR += log(R[j]);
            const double sR = exp(PhiK<KIND>::coupling_log(slR, PhiK<KIND>::prep(prm)));
            for (int j = 0; j < K; ++j) q[j] = PhiK<KIND>::marginal(sR / R[j]);
            const double eps = 1e-4;
            for (int j = 0; j < K; ++j) {
                const double keep = q[j];
                q[j] = keep * (1.0 + eps);
                pool_generic_k<KIND, K>(R, q, 1.0, prm, yy);
                q[j] = keep;
                unsafeAtomicAdd(&acc[diag_at + tok[j]], p[j] * fabs(yy[j]) / eps);
            }
        }
    }
    // sum arb: one atomic per wave
    for (int off = 32; off > 0; off >>= 1) arb += __shfl_down(arb, off);
    if ((threadIdx.x & 63) == 0 && arb != 0.0) unsafeAtomicAdd(&acc[arb_at], arb);
}

// tenders at the accepted prices, slot-major [K][m] as the C-ABI hands them out (two-asset.py:94,98)
template <int KIND, int K>
__global__ void __launch_bounds__(GK_THREADS)
tradesg_kernel(BucketG b, const double *__restrict__ nu, double *__restrict__ delta, double *__restrict__ lambda)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.m) return;
    double R[K], p[K], y[K];
    for (int j = 0; j < K; ++j) { R[j] = b.R[i * K + j]; p[j] = nu[b.idx[i * K + j]]; }
    pool_generic_k<KIND, K>(R, p, b.fee[i], b.param ? b.param[i] : 0.0, y);
    for (int j = 0; j < K; ++j) {
        delta[(size_t)j * b.m + i] = fmax(-y[j], 0.0);
        lambda[(size_t)j * b.m + i] = fmax(y[j], 0.0);
    }
}

}  // namespace cfmm
