// Barrier-smoothed pool subproblems and the dense dual Hessian: the second-order outer iteration
// for networks of near-linear pools (constant sum, stableswap near its peg -- BASELINE config 5),
// where the dual is almost piecewise linear and the projected quasi-Newton iteration crawls.
// gfx950 only.
//
// The reference hands the whole program to a primal-dual interior-point solver (cp.Problem.solve(),
// /root/reference/arbitrage.py:81-82).  Here the same log-barrier is put on the sign constraints
// Delta, Lambda >= 0 (arbitrage.py:51-52) *inside* the dual decomposition: for prices nu and a
// barrier weight mu every two-asset pool splits into two one-directional trades, each the 1-D problem
//
//     arb_mu(nu) = max_{D > 0}  nu_out L(D) - nu_in D + mu log D          L = forward exchange function,
//
// L(D) = R_out - Y(R_in + gamma D) on the pool's level set (arbitrage.py:60,63-74).  Its optimum is
// interior, so arb_mu is smooth in nu, with gradient (-D, L) and the rank-one Hessian
//     kappa (1, -L')(1, -L')',   kappa = 1 / (mu / D^2 - nu_out L''(D)).
// One launch solves both directions of every pool, scatter-adds psi_mu = sum (L - D) through an LDS
// tile (as the exact evaluation kernel does) and, when asked, the n x n Hessian in log-prices through
// global fp64 atomics (three per pool: two diagonal entries and the lower off-diagonal one).
//
//   smooth_kernel<HESS>        the smoothed evaluation of the four two-asset buckets (warm-started roots)
//   gn_newton_kernel<K, HESS>  k-asset geo-mean pools: exact solution + exact generalised Hessian (not smoothed)
//   apply_slo                  low-order log-prices: Newton steps below the fp64 resolution of log nu
//   smooth_trades_kernel       the interior tenders cfmm_get_trades2 returns after a second-order solve
//   hess_finish_kernel         diagonal terms, pinned tokens, padding, right-hand side -> chol.hpp's layout
#pragma once
#include "kernels.hpp"       // (phi2.hpp: the trading-function table, shared with the exact evaluation)

namespace cfmm {

template <int KIND>
__device__ __forceinline__ Fwd fwd2(double D, double Rin, double Rout, double g, double r, double C) { return Phi2<KIND>::fwd(D, Rin, Rout, g, r, C); }

struct Branch { double D, L, L1, kappa, val; };
#ifdef CFMM_SMOOTH_HIST
__device__ unsigned long long g_smooth_hist[128];      // tuning builds: iterations per direction solve
__device__ double g_smooth_samples[64 * 12];
__device__ unsigned int g_smooth_nsamples;
__device__ unsigned long long g_smooth_eff[4];         // lane-iterations | 64 x wave-maximum, summed over the direction solves: [0], [1] in-loop; the ratio is the SIMT efficiency of the loop
#endif

// root of  a D^2 + b D + mu = 0  (a < 0, mu > 0) in D > 0, without cancellation
__device__ __forceinline__ double barrier_root(double a, double b, double mu)
{
    const double rt = sqrt_nr(fma(b, b, -4.0 * a * mu));
    return b > 0.0 ? (b + rt) * rcp_nr(-2.0 * a) : 2.0 * mu * rcp_nr(rt - b);
}

// one direction of a two-asset pool: tender `in`, receive `out`.  Safeguarded iteration on
//     F(D) = A(D) + mu / D = 0,   A(D) = nu_out L'(D) - nu_in  (decreasing),
// each step linearises A only and keeps the barrier term exact (one step is exact whenever A is locally
// linear, in particular deep inside the no-trade band where D ~ mu / |A|); a step is taken only if it
// lands inside the bracket and at least halves the previous one, otherwise the bracket is bisected
// (the stableswap A is flat, then falls off a knee: plain Newton cycles across it).
// starting point without (or against) a warm start.  Trade side (A(0) > 0): the exact mu = 0 root where it is closed
// form; for the stableswap curve an estimate of where the marginal price m = phi_x / phi_y has dropped to
// rho = nu_in / (gamma nu_out): for y << x,  1 - m ~ alpha rho / (x y^2)  with  x ~ C - y  (three fixed-point sweeps) --
// a few per cent off the root at the 80/20 imbalance such trades end at, from where the iteration converges in 5-6
// steps (from D = 0 it first overshoots the knee and needs 12-16).  No-trade side: the root of the model at D = 0.
template <int KIND>
__device__ __forceinline__ double cold_start(double Rin, double Rout, double g, double r, double C, double ni, double no, double mu)
{
    const double De = Phi2<KIND>::start(Rin, Rout, g, r, C, ni, no);
    if (De > 0.0 && De < 1e300) return De;
    const Fwd f0 = Phi2<KIND>::fwd(0.0, Rin, Rout, g, r, C);
    return barrier_root(fmin(no * f0.L2, -1e-300), no * f0.L1 - ni, mu);
}

// `Dws` > 0: the root found by the previous evaluation of this direction, used as the starting point (prices and
// barrier weight move little between consecutive evaluations of the outer iteration); if the first step from it wants
// to move by more than 8x either way the direction has switched regime (trade <-> no-trade) and the cold start is better.
template <int KIND>
__device__ __forceinline__ Branch smooth_branch(double Rin, double Rout, double g, double r, double C,
                                                double ni, double no, double mu, double Dws)
{
    Branch o;
    bool warm = Dws > 0.0 && Dws < 1e300;
    if (warm) {
        // L'(0) needs no curve solve: in the no-trade regime (A(0) < 0) the root is at most mu / |A(0)|; a warm start
        // far above that is left over from the trade regime
        const double A0 = no * Phi2<KIND>::marginal0(Rin, Rout, g, r) - ni;
        if (A0 < 0.0 && Dws * -A0 > 4.0 * mu) warm = false;
    }
    double D = warm ? Dws : cold_start<KIND>(Rin, Rout, g, r, C, ni, no, mu);
    double lo = 0.0, hi = 1.7976931348623157e308, dprev = 1.7976931348623157e308;
    double Flo = 0.0, Fhi = 0.0;                 // F at the bracket ends (0: not known yet)
    int side = 0;                                // Illinois: which end the last false-position step kept
#ifdef CFMM_SMOOTH_HIST
    int nit_hist = 0;
#endif
    for (int it = 0; it < 120; ++it) {
#ifdef CFMM_SMOOTH_HIST
        nit_hist = it + 1;
#endif
        const Fwd f = fwd2<KIND>(D, Rin, Rout, g, r, C);
        const double A = no * f.L1 - ni, A1 = fmin(no * f.L2, -1e-300);
        const double bD = mu * rcp_nr(D), F = bD + A;
        // (the stationarity condition met to rounding: below ~4 ulp of its largest term the sign of F is noise -- a lane on the
        //  flat part of a stableswap curve would otherwise jitter until its bracket has been closed step by step)
        if (fabs(F) <= 1e-15 * fmax(ni, bD)) {
#ifdef CFMM_SMOOTH_HIST
            atomicAdd(&g_smooth_hist[it < 127 ? it : 127], 1ULL);
#endif
            break;
        }
        if (F > 0.0) { lo = D; Flo = F; } else { hi = D; Fhi = F; }
        double Dn = barrier_root(A1, A - A1 * D, mu);
        if (warm && (Dn > 8.0 * D || 8.0 * Dn < D)) {          // regime switch since the last evaluation: start over
            warm = false;
            D = cold_start<KIND>(Rin, Rout, g, r, C, ni, no, mu);
            lo = 0.0; hi = 1.7976931348623157e308; Flo = 0.0; Fhi = 0.0;
            continue;
        }
        warm = false;
        const double stepc = fabs(Dn - D);
        const bool conv = stepc <= 1e-13 * fmax(Dn, D);
        const bool ok = conv || (Dn > lo && Dn < hi && stepc < 0.5 * dprev);
        if (!ok) {
            if (!(hi < 1e308)) Dn = 2.0 * D;
            else if (Flo > 0.0 && Fhi < 0.0) {
                // false position on the bracket (F is decreasing), Illinois-damped, kept off the very ends
                const double wl = (side == 1) ? 0.5 * Flo : Flo, wh = (side == -1) ? 0.5 * Fhi : Fhi;
                const double xf = (lo * (-wh) + hi * wl) * rcp_nr(wl - wh);
                const double w = hi - lo;
                Dn = fmin(fmax(xf, lo + 0.02 * w), hi - 0.02 * w);
                side = (F > 0.0) ? 1 : -1;       // the end just updated is the one this step keeps
            } else Dn = 0.5 * (lo + hi);
        } else side = 0;
        dprev = fabs(Dn - D);
        D = Dn;
#ifdef CFMM_SMOOTH_HIST
        if (conv || dprev <= 1e-13 * D) {
            atomicAdd(&g_smooth_hist[it < 127 ? it : 127], 1ULL);
            if (it >= 12 && Dws > 0.0) {
                const unsigned k = atomicAdd(&g_smooth_nsamples, 1u);
                if (k < 64) { double *q = g_smooth_samples + 12 * k; q[0] = KIND; q[1] = Rin; q[2] = Rout; q[3] = g; q[4] = r; q[5] = ni; q[6] = no; q[7] = mu; q[8] = Dws; q[9] = D; q[10] = it; q[11] = C; }
            }
            break;
        }
#else
        if (conv || dprev <= 1e-13 * D) break;
#endif
    }
#ifdef CFMM_SMOOTH_HIST
    {
        const double mine = (double)nit_hist, mx = wave_allmax(mine), sm = wave_allsum(mine);
        if ((threadIdx.x & 63) == 0) { atomicAdd(&g_smooth_eff[0], (unsigned long long)sm); atomicAdd(&g_smooth_eff[1], (unsigned long long)(64.0 * mx)); }
    }
#endif
    const Fwd f = fwd2<KIND>(D, Rin, Rout, g, r, C);
    o.D = D; o.L = f.L; o.L1 = f.L1;
    { const double iD = rcp_nr(D); o.kappa = rcp_nr(fma(mu * iD, iD, -no * f.L2)); }
    o.val = no * f.L - ni * D + mu * log(D);
    return o;
}

// constant sum (arbitrage.py:73-74): L = gamma D with D <= R_out / gamma, a barrier on both ends:
//     max  s D + mu log D + mu log(cap - D),  s = gamma nu_out - nu_in      closed form
__device__ __forceinline__ Branch smooth_branch_sum(double Rout, double g, double ni, double no, double mu)
{
    Branch o;
    const double cap = Rout / g, s = g * no - ni;
    const double sc = s * cap;
    const double disc = sqrt(fma(sc, sc, 4.0 * mu * mu));
    const double b1 = sc - 2.0 * mu, b2 = -sc - 2.0 * mu;
    const double D = b1 > 0.0 ? (b1 + disc) / (2.0 * s) : 2.0 * mu * cap / (disc - b1);
    const double E = b2 > 0.0 ? (b2 + disc) / (-2.0 * s) : 2.0 * mu * cap / (disc - b2);     // cap - D
    o.D = D; o.L = g * D; o.L1 = g;
    o.kappa = 1.0 / (mu / (D * D) + mu / (E * E));
    o.val = s * D + mu * (log(D) + log(E));
    return o;
}

struct SmoothArgs {
    Bucket2 b2[N_KINDS2];
    double *ws[N_KINDS2];       // per kind: [2][m] roots of the previous evaluation (warm start), or null
    int tile_end[N_KINDS2];     // cumulative wave-tiles (64 pools) in the order curve2, pow2, w2, cp2, sum2
    int ntiles, n;
    const double *nu;           // [n] prices
    const double *slo;          // [n] low-order part of the log-prices (see smooth_tile), or null
    double mu;
    double *out;                // [n] psi_mu | [n] sum of branch values | [n + 1] sum nu'(L - D)   (zeroed by the host)
    double *H;                  // [n x n] column-major, lower triangle gets the pools' part (zeroed by the host); may be null
    int ldh;
};

template <int KIND>
__device__ __forceinline__ void smooth_pool(const Bucket2 &b, long long i, double pa, double pb, double mu,
                                            Branch &ab, Branch &ba, double *ws = nullptr)
{
    const double Ra = b.Ra[i], Rb = b.Rb[i], g = b.fee[i];
    if (KIND == 2) {
        ab = smooth_branch_sum(Rb, g, pa, pb, mu);
        ba = smooth_branch_sum(Ra, g, pb, pa, mu);
    } else {
        constexpr int K = KIND == 2 ? 0 : KIND;
        const double prm = KIND == 0 ? 0.0 : b.param[i];
        const double C = Phi2<K>::level(Ra, Rb, prm);
        const double rab = Phi2<K>::ratio(prm, true), rba = Phi2<K>::ratio(prm, false);
        ab = smooth_branch<K>(Ra, Rb, g, rab, C, pa, pb, mu, ws ? ws[i] : 0.0);          // tender a, receive b
        ba = smooth_branch<K>(Rb, Ra, g, rba, C, pb, pa, mu, ws ? ws[b.m + i] : 0.0);    // tender b, receive a
        if (ws) { ws[i] = ab.D; ws[b.m + i] = ba.D; }
    }
}

// Low-order log-prices.  Near-linear pools react to price changes far below the fp64 resolution of log nu (a
// partially filled constant-sum pool sets its fill through a price difference of ~1e-13), so the last Newton steps
// cannot be added to the log-prices themselves.  They are carried in a separate vector s_lo instead and enter every
// pool direction through its own exact first-order response  dD = -kappa (u, v).(s_lo_in, s_lo_out),  dL = L' dD.
__device__ __forceinline__ void apply_slo(Branch &br, double u, double v, double din, double dout)
{
    const double dD = -br.kappa * fma(u, din, v * dout);
    br.D += dD;
    br.L = fma(br.L1, dD, br.L);
}

// The Hessian terms of a workgroup, collected in LDS before they go to the n x n array in HBM.  Three global fp64 atomics per
// pool were two thirds of smooth_kernel<true> (104 of 138 us at config 5): 1.65 M atomics on 1000 diagonal addresses and --
// stableswap pools concentrate on few pairs: 5e5 pools on 1500 of them -- ~330 per off-diagonal address.  Now the diagonal
// is an LDS tile like psi (n atomics per workgroup at the flush), and the off-diagonal entries go through a small
// open-addressed table in LDS keyed by the entry's index (two probes, claim by compare-and-swap; a miss on both falls
// through to the global atomic): with the workgroup walking a CONTIGUOUS range of every bucket and the pools ordered by
// token blocks (reorder.hpp) its few thousand pools touch one or two hundred pairs.
struct HessCache {
    static constexpr int SLOTS = 2048;           // 8 KB of keys + 16 KB of values
    double *diag;                                // [n]
    double *val;                                 // [SLOTS]
    int *key;                                    // [SLOTS], -1 = free
    __device__ __forceinline__ void add(int k, double v, double *H) const
    {
        unsigned h = ((unsigned)k * 2654435761u) >> 21;
#pragma unroll
        for (int probe = 0; probe < 2; ++probe) {
            int t = key[h];
            if (t == -1) t = atomicCAS(&key[h], -1, k), t = (t == -1) ? k : t;
            if (t == k) { unsafeAtomicAdd(&val[h], v); return; }
            h = (h + 1) & (SLOTS - 1);
        }
        unsafeAtomicAdd(&H[k], v);
    }
};

template <int KIND, bool HESS>
__device__ __forceinline__ void smooth_tile(const Bucket2 &b, long long i0, int lane, const double *nu_s, double *psi_s,
                                            const SmoothArgs &a, const HessCache &hc, double &vsum, double &tsum)
{
    long long i = i0 + lane;
    const bool live = i < b.m && !(KIND == 2 && b.flags && b.flags[i]);
    i = i < b.m ? i : b.m - 1;
    const int ia = b.ia[i], ib = b.ib[i];
    const double pa = nu_s[ia], pb = nu_s[ib];
    Branch ab, ba;
    smooth_pool<KIND>(b, i, pa, pb, a.mu, ab, ba, (KIND != 2 && i0 + lane < b.m) ? a.ws[KIND] : nullptr);
    if (!live) return;
    if (a.slo) {
        const double da = a.slo[ia], db = a.slo[ib];
        apply_slo(ab, pa, -ab.L1 * pb, da, db);
        apply_slo(ba, pb, -ba.L1 * pa, db, da);
    }
    const double ya = ba.L - ab.D, yb = ab.L - ba.D;
    unsafeAtomicAdd(&psi_s[ia], ya);
    unsafeAtomicAdd(&psi_s[ib], yb);
    vsum += ab.val + ba.val;
    tsum += pa * ya + pb * yb;
    if (HESS) {
        // in log-prices: branch ab moves (a, b) along (pa, -L1 pb), branch ba along (-L1 pa, pb)
        const double u1 = pa, v1 = -ab.L1 * pb, u2 = -ba.L1 * pa, v2 = pb;
        const double haa = ab.kappa * u1 * u1 + ba.kappa * u2 * u2;
        const double hbb = ab.kappa * v1 * v1 + ba.kappa * v2 * v2;
        const double hab = ab.kappa * u1 * v1 + ba.kappa * u2 * v2;
        const int row = ia > ib ? ia : ib, col = ia > ib ? ib : ia;
        // the diagonal through an LDS tile like psi; the off-diagonal entry through the workgroup's pair cache (HessCache)
        unsafeAtomicAdd(&hc.diag[ia], haa);
        unsafeAtomicAdd(&hc.diag[ib], hbb);
        hc.add(col * a.ldh + row, hab, a.H);
    }
}

constexpr int SMOOTH_THREADS = 512;

// LDS: psi_s[n] | nu_s[n] | red[2 * 8] | ticket, tile-range table | HESS: diag[n] | pair-cache values | pair-cache keys
__host__ __device__ inline size_t smooth_lds_bytes(int n, bool hess)
{
    return (size_t)(2 * n + 32 + (hess ? n + (n & 1) + HessCache::SLOTS : 0)) * sizeof(double) + (hess ? HessCache::SLOTS * sizeof(int) : 0);
}

template <bool HESS>
__global__ void __launch_bounds__(SMOOTH_THREADS)
smooth_kernel(SmoothArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = a.n;
    double *psi_s = lds, *nu_s = lds + n, *red = lds + 2 * n;
    int *next_tile = reinterpret_cast<int *>(red + 16);          // [0] ticket, [2 + q] tiles of buckets <= q in this workgroup, [2 + N_KINDS2 + q] its first tile in bucket q
    HessCache hc = {};
    if (HESS) {
        hc.diag = lds + 2 * n + 32;
        hc.val = hc.diag + n + (n & 1);
        hc.key = reinterpret_cast<int *>(hc.val + HessCache::SLOTS);
        for (int j = threadIdx.x; j < n; j += blockDim.x) hc.diag[j] = 0.0;
        for (int j = threadIdx.x; j < HessCache::SLOTS; j += blockDim.x) { hc.val[j] = 0.0; hc.key[j] = -1; }
    }
    // workgroup b walks a contiguous share of every bucket, [b n_q / G, (b + 1) n_q / G) of its n_q tiles (as the exact
    // evaluation does): with the pools ordered by token blocks its pairs are few -- what the pair cache lives on
    if (threadIdx.x == 0) {
        int c = 0;
        for (int q = 0; q < N_KINDS2; ++q) {
            const int nq = a.tile_end[q] - (q ? a.tile_end[q - 1] : 0);
            const int s0 = (int)(((double)blockIdx.x * nq) / (double)gridDim.x), s1 = (int)(((double)(blockIdx.x + 1) * nq) / (double)gridDim.x);
            c += s1 - s0;
            next_tile[2 + q] = c; next_tile[2 + N_KINDS2 + q] = s0;
        }
        *next_tile = 0;
    }
    for (int j = threadIdx.x; j < n; j += blockDim.x) { nu_s[j] = a.nu[j]; psi_s[j] = 0.0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int nlocal = next_tile[2 + N_KINDS2 - 1];
    double vsum = 0.0, tsum = 0.0;
    int ticket = 0;
    if (lane == 0) ticket = atomicAdd(next_tile, 1);
    for (;;) {
        const int t = __builtin_amdgcn_readfirstlane(ticket);
        if (t >= nlocal) break;
        if (lane == 0) ticket = atomicAdd(next_tile, 1);
        int bk = 0;
#pragma unroll
        for (int q = 0; q < N_KINDS2 - 1; ++q) bk += (t >= next_tile[2 + q]) ? 1 : 0;
        const long long i0 = (long long)(next_tile[2 + N_KINDS2 + bk] + t - (bk ? next_tile[2 + bk - 1] : 0)) * 64;
        switch (bk) {
        case 0: smooth_tile<3, HESS>(a.b2[3], i0, lane, nu_s, psi_s, a, hc, vsum, tsum); break;
        case 1: smooth_tile<4, HESS>(a.b2[4], i0, lane, nu_s, psi_s, a, hc, vsum, tsum); break;
        case 2: smooth_tile<1, HESS>(a.b2[1], i0, lane, nu_s, psi_s, a, hc, vsum, tsum); break;
        case 3: smooth_tile<0, HESS>(a.b2[0], i0, lane, nu_s, psi_s, a, hc, vsum, tsum); break;
        default: smooth_tile<2, HESS>(a.b2[2], i0, lane, nu_s, psi_s, a, hc, vsum, tsum); break;
        }
    }
    vsum = wave_allsum(vsum); tsum = wave_allsum(tsum);
    if (lane == 0) { red[wib] = vsum; red[8 + wib] = tsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = 0.0, t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { v += red[w]; t += red[8 + w]; }
        unsafeAtomicAdd(&a.out[n], v);
        unsafeAtomicAdd(&a.out[n + 1], t);
    }
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const double v = psi_s[j];
        if (v != 0.0) unsafeAtomicAdd(&a.out[j], v);
        if (HESS) { const double d = hc.diag[j]; if (d != 0.0) unsafeAtomicAdd(&a.H[(size_t)j * a.ldh + j], d); }
    }
    if (HESS) {
        for (int j = threadIdx.x; j < HessCache::SLOTS; j += blockDim.x) {
            const int k = hc.key[j];
            if (k >= 0) unsafeAtomicAdd(&a.H[k], hc.val[j]);
        }
    }
}

// k-asset weighted geo-mean pools (arbitrage.py:65) inside the second-order iteration: strictly curved, so they are
// NOT smoothed -- exact solution (pool_math.hpp's KKT: x_j = R_j e^{f(t - a_j)}), exact (generalised) Hessian.  With
// A the legs that trade at the root t, the pool's value has, in log-prices and without its diag(nu * psi) term,
//     H_jk = e^t (w_j delta_jk - w_j w_k / sum_A w),   j, k in A
// (a leg's traded value is p_j c_j x_j = w_j e^t on either side of the fee; d t / d log p_k = w_k / sum_A w).
// One pool per thread, psi / value through global atomics: this kernel runs a few dozen times per solve, the
// leg-per-lane machinery of the exact evaluation kernel is not worth repeating here.
template <int K, bool HESS>
__global__ void __launch_bounds__(256)
gn_newton_kernel(BucketN b, const double *__restrict__ nu, const double *__restrict__ slo, double *__restrict__ out, int n,
                 double *__restrict__ H, int ldh)
{
    double vsum = 0.0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < b.m; i += (long long)gridDim.x * blockDim.x) {
        double R[K], w[K], p[K], a[K];
        int tok[K];
        const double g = b.fee[i], lg = b.lfee[i];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            R[j] = b.R[i * K + j]; w[j] = b.w[i * K + j]; tok[j] = b.idx[i * K + j];
            p[j] = nu[tok[j]];
            a[j] = log(R[j] * p[j] / w[j]);
        }
        // root of the piecewise-linear residual F(t) = sum_j w_j f(t - a_j): bracket among its 2K breakpoints
        double tL = -1.7976931348623157e308, fL = 0.0, tR = 1.7976931348623157e308, fR = 0.0;
#pragma unroll
        for (int q = 0; q < 2 * K; ++q) {
            const double t = (q < K) ? a[q % K] : a[q % K] - lg;
            double f = 0.0;
#pragma unroll
            for (int j = 0; j < K; ++j) { const double u = t - a[j]; f += w[j] * (u < 0.0 ? u : (u > -lg ? u + lg : 0.0)); }
            if (f <= 0.0 && t > tL) { tL = t; fL = f; }
            if (f >= 0.0 && t < tR) { tR = t; fR = f; }
        }
        const double t = fL == 0.0 ? tL : (fR == 0.0 ? tR : tL - fL * (tR - tL) / (fR - fL));
        const double et = exp(t);
        double wa = 0.0, val = 0.0, dt = 0.0;
        bool act[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const double u = t - a[j];
            act[j] = u < 0.0 || u > -lg;                           // withdrawn / deposited
            if (act[j]) { wa += w[j]; if (slo) dt += w[j] * slo[tok[j]]; }
        }
        if (wa > 0.0) dt /= wa;                                    // first-order move of the root under the low-order log-prices
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const double u = t - a[j];
            const bool wd = u < 0.0, dp = u > -lg;
            double y = 0.0;
            if (wd) y = -R[j] * expm1(u);                          // R - x,  x = R e^u
            if (dp) y = -R[j] * expm1(u + lg) / g;                 // (R - x) / gamma,  x = R e^{u + lg}
            if (slo && act[j]) y -= w[j] * et / p[j] * (dt - slo[tok[j]]);      // dy_j = -(c_j x_j)(dt - s_lo_j), c_j x_j = w_j e^t / p_j
            if (y != 0.0) { unsafeAtomicAdd(&out[tok[j]], y); val += p[j] * y; }
        }
        vsum += val;
        if (HESS && wa > 0.0) {
            const double s = et / wa;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (!act[j]) continue;
#pragma unroll
                for (int k = 0; k <= j; ++k) {
                    if (!act[k]) continue;
                    const double h = (j == k) ? et * w[j] - s * w[j] * w[j] : -s * w[j] * w[k];
                    const int row = tok[j] > tok[k] ? tok[j] : tok[k], col = tok[j] > tok[k] ? tok[k] : tok[j];
                    unsafeAtomicAdd(&H[(size_t)col * ldh + row], h);
                }
            }
        }
    }
    vsum = wave_allsum(vsum);
    if ((threadIdx.x & 63) == 0 && vsum != 0.0) { unsafeAtomicAdd(&out[n], vsum); unsafeAtomicAdd(&out[n + 1], vsum); }
}

// tenders of the smoothed solution, slot-major [2][m] like trades2_kernel: what cfmm_get_trades2 returns after a
// second-order solve -- the primal point the certificates were computed on.  Both directions are (slightly) open,
// so a pool both tenders and receives each token, as the reference's Delta_i, Lambda_i >= 0 allow (arbitrage.py:51-52).
template <int KIND>
__global__ void __launch_bounds__(256)
smooth_trades_kernel(Bucket2 b, const double *__restrict__ nu, const double *__restrict__ slo, double mu,
                     double *__restrict__ delta, double *__restrict__ lambda)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < b.m; i += (long long)gridDim.x * blockDim.x) {
        Branch ab, ba;
        ab.D = ab.L = ba.D = ba.L = 0.0;
        if (!(KIND == 2 && b.flags && b.flags[i])) {
            const int ia = b.ia[i], ib = b.ib[i];
            const double pa = nu[ia], pb = nu[ib];
            smooth_pool<KIND>(b, i, pa, pb, mu, ab, ba);
            if (slo) { apply_slo(ab, pa, -ab.L1 * pb, slo[ia], slo[ib]); apply_slo(ba, pb, -ba.L1 * pa, slo[ib], slo[ia]); }
        }
        const long long o = b.perm ? b.perm[i] : i;
        delta[o] = ab.D;   delta[b.m + o] = ba.D;
        lambda[o] = ba.L;  lambda[b.m + o] = ab.L;
    }
}

// Finish the linear system in place: H <- H + diag(hd); rows / columns of the pinned tokens (mask != 0) and of
// the padding (n .. nr) replaced by the identity; the right-hand side written as row nr (chol.hpp's layout).
// Only the lower triangle is referenced by the factorisation.
__global__ void __launch_bounds__(256)
hess_finish_kernel(double *__restrict__ H, int n, int nr, int ldh, const double *__restrict__ hd, const int *__restrict__ mask,
                   const double *__restrict__ rhs)
{
    const long long total = (long long)(nr + 1) * nr;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(e % (nr + 1)), col = (int)(e / (nr + 1));
        if (row < col) continue;
        double v;
        if (row == nr) v = (col < n && !mask[col]) ? rhs[col] : 0.0;
        else if (row >= n || col >= n || mask[row] || mask[col]) v = row == col ? 1.0 : 0.0;
        else v = H[(size_t)col * ldh + row] + (row == col ? hd[row] : 0.0);
        H[(size_t)col * ldh + row] = v;
    }
}

}  // namespace cfmm
