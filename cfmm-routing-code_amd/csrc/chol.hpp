// Dense Cholesky factorisation and solve for the n x n dual Hessian of the second-order iteration
// (n = tokens, 10^2 .. a few 10^3), fp64, gfx950 only.  Written here rather than taken from rocSOLVER:
// at this size the vendor path costs ~3.5 ms per factorisation (and minutes of one-time library
// initialisation on a fresh box) where the problem is bound by the length of the elimination chain
// (n dependent steps) and by launch latency.
//
// Layout: column-major, lower triangle.  nr = n rounded up to the block size NB = 32 (padding rows / columns
// carry an identity diagonal); the right-hand side rides along as ONE EXTRA ROW at index nr, so the
// factorisation forward-substitutes it for free (row nr of L is y = L^-1 b); ld = nr + NB.
// Right-looking with a look-ahead of one block column, ONE launch per block column (chol_step_kernel below): the panel
// workgroups of column k + 1 apply their own share of update k, factor the diagonal block (one wave, in registers, factor and
// inverse together: WaveFactor) and form their rows as a product with that inverse, while the other workgroups of the same
// launch give update k to the columns from k + 2 on.  1000 x 1000: 0.75 ms (round 2: two launches per column, barrier-stepped
// block routines) -> 0.35 ms.
// chol_back_kernel: L' x = y, one workgroup, dot form (every pass over L reads contiguous columns: wave w owns NB/16 columns
// of the block), the diagonal block applied as an NB x NB mat-vec with its inverse.
// (Tried and dropped: NB = 64 -- half the launches, but the diagonal block's elimination chain grows fourfold: 1.06 ms;
//  round 2's one-launch scheme with the panel rows formed through an inverse that took a 16-18 us barrier-stepped chain of
//  its own; a barrier-stepped four-wave block factorisation, two columns per barrier: 16 300 cycles a block against 11 700
//  for the single wave, and it does not yield the inverse.)
#pragma once
#include "kernels.hpp"

namespace cfmm {

constexpr int CH_NB = 32;

// v of lane `src` (a compile-time lane) to every lane: two v_readlane_b32
template <int SRC> __device__ __forceinline__ double lane_bcast(double v)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), SRC), hi = __builtin_amdgcn_readlane(__double2hiint(v), SRC);
    return __hiloint2double(hi, lo);
}

// Cholesky factor AND its inverse of a 32 x 32 block by ONE wave, in registers, no barrier.  Lane c < 32 holds column c
// (= row c) of the symmetric block in a[0..32), lane 32 + c column c of the identity.  Elimination step j is the same
// instruction stream on both halves: d = 1 / sqrt(pivot); row j scaled by d (lane c now holds L[c][j], lane 32 + c holds
// Linv[j][c]); row r > j minus L[r][j] times row j.  The left half is the right-looking Cholesky on the full symmetric
// square, the right half the same row operations on I, i.e. L^-1.
// What bounds it is the DEPENDENT chain from one pivot to the next (a lone wave waits out every fp64 result: ~16 cycles an
// operation), so that chain is kept to [rsq, 3 operations of a third-order correction, 2 operations for the next pivot]:
//   * the next pivot is formed from uniform values, p' = c - (b y)^2 with b = a[j] and c = a[j + 1] of lane j + 1, both
//     broadcast (v_readlane) BEFORE y = 1 / sqrt(p) is known -- no lane traffic and no scalar-unit round trip inside the chain;
//   * the pivot's sign test is off the chain (a non-positive pivot yields NaNs and the `false` the caller reports);
//   * rows j + 1 and j + 2 take step j's update at once, through uniform multipliers (a[j] of lanes j + 1, j + 2 times y);
//     the other 29 - j rows have two steps of slack: they are updated one step LATE, in the shadow of step j + 1's chain,
//     with L[r][j] broadcast out of a 2 x 64 LDS buffer (ds_read_b128: one LDS instruction and two fmas per pair of terms,
//     against two v_readlane, a hazard nop and an fma per term), so no LDS latency sits inside the chain either.
// `Lb`: 128 doubles of LDS.  On return lane c < 32: a[j] = L[c][j] (j <= c), lane 32 + c: a[j] = Linv[j][c] (exactly 0 for j < c).
template <int J> struct WaveFactor {
    // late rows J + 2 + K, J + 2 + K + NS, ...: a[r] -= L[r][J - 1] lp, the multipliers in lv (read from LDS a step ago)
    template <int K, int NS> static __device__ __forceinline__ void slot(double (&a)[CH_NB], const double (&lv)[CH_NB], double lp)
    {
        if constexpr (J > 0) {
#pragma unroll
            for (int r = J + 2 + K; r < CH_NB; r += NS) a[r] = fma(-lv[r], lp, a[r]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // p: the pivot a[J][J] with every earlier column's update applied, the same value in every lane.  lp: column J - 1 (this lane's
    // entry), lv[r] = L[r][J - 1] for the rows r >= J + 2 that still lack column J - 1's update.
    // The issue order is pinned (sched_barrier): one operation of the chain, then a share of the late rows -- a lone wave issues in
    // order, and left alone the scheduler puts the chain first (seven stalls on fp64 latency) and the independent fmas behind it.
    static __device__ __forceinline__ bool run(double (&a)[CH_NB], double p, double lp, const double (&lv)[CH_NB], double *Lb, int lane)
    {
        if constexpr (J < CH_NB) {
            constexpr int NS = 6;
            const double y0 = __builtin_amdgcn_rsq(p);
            double b1 = 0.0, b2 = 0.0, c = 0.0;
            if constexpr (J + 1 < CH_NB) { b1 = lane_bcast<J + 1>(a[J]); c = lane_bcast<J + 1>(a[J + 1]); }
            if constexpr (J + 2 < CH_NB) b2 = lane_bcast<J + 2>(a[J]);
            __builtin_amdgcn_sched_barrier(0);
            const double t = -p * y0;
            slot<0, NS>(a, lv, lp);
            const double e = fma(t, y0, 1.0);                             // y = y0 (1 + e / 2 + 3 e^2 / 8): relative error O(e^3)
            slot<1, NS>(a, lv, lp);
            const double q = y0 * e, h = fma(0.375, e, 0.5);
            slot<2, NS>(a, lv, lp);
            const double y = fma(q, h, y0);
            slot<3, NS>(a, lv, lp);
            const double m1 = b1 * y, l = a[J] * y, m2 = b2 * y;          // m1 = L[J + 1][J], m2 = L[J + 2][J]
            Lb[(J & 1) * 64 + lane] = l;                                  // (no branch: the identity half's lanes write slots nobody reads)
            double lvn[CH_NB];
#pragma unroll
            for (int r = J + 3; r < CH_NB; ++r) lvn[r] = Lb[(J & 1) * 64 + r];     // (the next step's late rows: issued now, used a chain later)
            slot<4, NS>(a, lv, lp);
            const double pn = fma(-m1, m1, c);
            a[J] = l;
            slot<5, NS>(a, lv, lp);
            if constexpr (J + 1 < CH_NB) a[J + 1] = fma(-m1, l, a[J + 1]);
            if constexpr (J + 2 < CH_NB) a[J + 2] = fma(-m2, l, a[J + 2]);
            const bool pos = p > 0.0 && p < 1.7976931348623157e308;
            return WaveFactor<J + 1>::run(a, pn, l, lvn, Lb, lane) && pos;
        } else return true;
    }
};

// ------------------------------------------------------------------------------------------------------------------------
// One launch per block column (round 3): panel k + 1 and the trailing update of panel k run SIDE BY SIDE.
//
// The two-launch scheme of round 2 serialised [panel k: a 32-column elimination chain, ~13 us] -> [update k: ~10 us of mostly
// latency] 32 times.  But block column k + 1 is the only part of the trailing matrix the next panel needs, and it needs
// only ITS OWN share of update k: so the workgroups that will factor / solve column k + 1 apply that share themselves --
// D = A_dd - P_d P_d' for the diagonal block (redundantly in every panel workgroup, like its factorisation), B_i = A_i -
// P_i P_d' for their 64 rows -- and go straight on to the panel, while other workgroups of the SAME launch give panel k's
// update to the columns from k + 2 on.  No workgroup waits for another: what a launch writes (column k + 1 by the panel
// workgroups, columns >= k + 2 by the update workgroups) is disjoint, and what it reads of panel k was finished by the
// previous launch.  32 launches instead of 63.
//
// The panel role, per workgroup of four waves (cycles at 2.4 GHz measured in the first version -> this one):
//   stage    every global operand (own 64 rows of the column, the diagonal block, the previous panel's rows for both) into
//            registers, then LDS                                                                    3700
//   D        A_dd - P_d P_d', full symmetric square, 2 x 2 register tiles                            2700 -> ~800
//   factor   WaveFactor in wave 0 (factor + inverse)  ||  waves 2-3: B = A_i - P_i P_d' (4 x 4 tiles)  16 300 + 5700 -> ~9000
//   solve    X = B Linv' as a 64 x 32 x 32 product out of LDS (the triangular solve it replaces was a 32-step chain
//            with a barrier every other step)                                                        9400 -> ~1500
// blockIdx < npanel: panel role for block column k1 (workgroup 0 has no rows: it writes the factor and its inverse); the
// others: tile (ti, tj) of the trailing update with panel k0 = k1 - NB over the rows / columns from k1 + NB on.
// ------------------------------------------------------------------------------------------------------------------------
// Round 4, a third role: the INVERSE FACTOR W = L^-1 rides along (workgroups >= nfac), so that the back substitution
// L' x = y -- 118 us in ONE workgroup, a chain of 32 dependent block steps -- becomes ONE matrix-vector product x = W' y over
// the whole chip (chol_wt_kernel).  Right-looking, one block row of W per launch, one launch behind the factorisation:
//   R starts as the identity (implied: a block's first update, from row j, starts from zero);  launch t finishes block row q = t - 1,
//       W_qj = Linv_qq R_qj   (j <= q;  Linv_qq, the inverse diagonal block, came out of launch q's WaveFactor),
//   and gives its update to every row below,  R_ij -= L_iq W_qj  (i > q;  L_iq is panel q, finished by launch q).
// Workgroup (i, j) forms W_qj ITSELF (one 32^3 product, redundantly in the nb - 1 - q workgroups of column j) and applies it to
// its R_ij (a second one): no workgroup waits for another, what a launch reads was finished by earlier launches, what it
// writes is disjoint -- the scheme of the factorisation's own look-ahead.  The workgroup with i = q + 1 also stores W_qj.
// Balanced (every workgroup: two block products) where the row-sum form  W_qj = -Linv_qq sum_k L_qk W_kj  would put q products
// on one workgroup -- as long as the panel's chain.  The last block row is never formed: chol_wt_kernel applies it as
// R_last' (Linv_last' y_last).  Fixed summation orders throughout: bitwise the same on every rank of a pool-sharded solve.
__global__ void __launch_bounds__(256)
chol_step_kernel(double *__restrict__ A, int ld, int nrows, int ncols, int k1, int npanel, double *__restrict__ Dinv, int *__restrict__ info,
                 int nfac, double *__restrict__ Wm, double *__restrict__ Rm, int ldw, int arrive_target)
{
    constexpr int NB = CH_NB;
    __shared__ __attribute__((aligned(16))) double lds[NB * NB + NB * 64 + NB * 64 + NB * (NB + 1) + NB * NB + 2 * 64];      // 57.25 KB (the tile role uses 33 KB of it)
    const int tid = threadIdx.x;
    const bool have_prev = k1 > 0;
    const int k0 = k1 - NB;
    // (workgroup order: panel | inverse-factor tiles | trailing-update tiles -- the dispatcher hands workgroups out in index order, and
    //  the inverse-factor tiles are the longer of the two side roles: `nfac` = where the trailing tiles begin)
    if ((int)blockIdx.x >= npanel && (int)blockIdx.x < nfac) {
        // ---- inverse factor: block row q = k1 / NB - 1, tile (i, j), j <= q < i < ncols / NB ------------------------------------------
        const int q = k1 / NB - 1, nbk = ncols / NB;
        const int t = (int)blockIdx.x - npanel, j = t % (q + 1), i = q + 1 + t / (q + 1);
        double *Li = lds, *Rq = Li + NB * NB, *Wq = Rq + NB * NB, *Lq = Wq + NB * NB;       // 4 x 8 KB: Linv_qq | R_qj | W_qj | L_iq
        (void)nbk;
        const int r2 = 2 * (tid & 15), c2 = 2 * (tid >> 4);      // this thread's 2 x 2 sub-tile: rows r2, r2 + 1, columns c2, c2 + 1
        double lv[4], rv[4], av[4], xv[2][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                             // (all loads first)
            const int e = tid + 256 * u, rr = e & 31, cc = e >> 5;
            lv[u] = Dinv[(size_t)q * NB * NB + e];                // Dinv[q][row * NB + col] = Linv[row][col]
            rv[u] = j == q ? (rr == cc ? 1.0 : 0.0) : Rm[(size_t)(j * NB + cc) * ldw + q * NB + rr];
            av[u] = A[(size_t)(q * NB + cc) * ld + i * NB + rr];  // L_iq: rows i NB + rr, column q NB + cc
        }
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int u = 0; u < 2; ++u) xv[u][v] = j == q ? 0.0 : Rm[(size_t)(j * NB + c2 + v) * ldw + i * NB + r2 + u];      // (R_ij's first update comes from row j: it starts from zero -- no memset of R per factorisation)
        // (every LDS operand k-major, so that a step of a product is two 16-byte reads: [k][row pair] and [k][column pair] --
        //  the first layout read Linv row-major, sixteen lanes on one bank: the tiles took 12 us, longer than the panel's chain)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u, rr = e & 31, cc = e >> 5;
            Li[rr * NB + cc] = lv[u];                             // Linv[row = cc][col = rr] -> Li[k = rr][r = cc]
            Rq[rr * NB + cc] = rv[u];                             // R_qj[rr][cc] -> Rq[k = rr][c = cc]
            Lq[cc * NB + rr] = av[u];                             // L_iq[rr][cc] -> Lq[k = cc][r = rr]
        }
        __syncthreads();
        double w00 = 0.0, w01 = 0.0, w10 = 0.0, w11 = 0.0;        // W_qj = Linv_qq R_qj
#pragma unroll 8
        for (int k = 0; k < NB; ++k) {
            const double2 a = *reinterpret_cast<const double2 *>(Li + k * NB + r2), b = *reinterpret_cast<const double2 *>(Rq + k * NB + c2);
            w00 = fma(a.x, b.x, w00); w01 = fma(a.x, b.y, w01); w10 = fma(a.y, b.x, w10); w11 = fma(a.y, b.y, w11);
        }
        *reinterpret_cast<double2 *>(Wq + r2 * NB + c2) = make_double2(w00, w01);            // Wq[k = row][c]
        *reinterpret_cast<double2 *>(Wq + (r2 + 1) * NB + c2) = make_double2(w10, w11);
        if (i == q + 1) {                                        // the finished block row: W_qj (strictly lower blocks; the diagonal ones stay in Dinv)
            if (j < q) {
                Wm[(size_t)(j * NB + c2) * ldw + q * NB + r2] = w00; Wm[(size_t)(j * NB + c2 + 1) * ldw + q * NB + r2] = w01;
                Wm[(size_t)(j * NB + c2) * ldw + q * NB + r2 + 1] = w10; Wm[(size_t)(j * NB + c2 + 1) * ldw + q * NB + r2 + 1] = w11;
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < NB; ++k) {                            // R_ij -= L_iq W_qj
            const double2 a = *reinterpret_cast<const double2 *>(Lq + k * NB + r2), b = *reinterpret_cast<const double2 *>(Wq + k * NB + c2);
            xv[0][0] = fma(-a.x, b.x, xv[0][0]); xv[0][1] = fma(-a.x, b.y, xv[0][1]); xv[1][0] = fma(-a.y, b.x, xv[1][0]); xv[1][1] = fma(-a.y, b.y, xv[1][1]);
        }
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int u = 0; u < 2; ++u) Rm[(size_t)(j * NB + c2 + v) * ldw + i * NB + r2 + u] = xv[u][v];
        return;
    }
    if ((int)blockIdx.x >= nfac) {
        // ---- trailing update with panel k0 over rows / columns >= k1 + NB: 64 x 64 tile (ti, tj) of the lower triangle ---------
        double (*Pi)[64 + 1] = reinterpret_cast<double (*)[64 + 1]>(lds), (*Pj)[64 + 1] = Pi + 32;
        int t = (int)blockIdx.x - nfac, ti = 0;
        while (t > ti) { t -= ti + 1; ++ti; }
        const int tj = t;
        const int base = k1 + NB;
        const int i0 = base + 64 * ti, j0 = base + 64 * tj;
        const int tx = tid & 15, ty = tid >> 4;      // rows tx + 16 u, columns ty + 16 v
        double cv[4][4];
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = i0 + tx + 16 * u, col = j0 + ty + 16 * v;
                cv[u][v] = (row < nrows && col < ncols && row >= col) ? A[(size_t)col * ld + row] : 0.0;
            }
        for (int e = tid; e < 32 * 64; e += 256) {
            const int c = e >> 6, rr = e & 63;
            Pi[c][rr] = (i0 + rr < nrows) ? A[(size_t)(k0 + c) * ld + i0 + rr] : 0.0;
            Pj[c][rr] = (j0 + rr < ncols) ? A[(size_t)(k0 + c) * ld + j0 + rr] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int c = 0; c < 32; ++c) {
            double pi[4], pj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { pi[u] = Pi[c][tx + 16 * u]; pj[u] = Pj[c][ty + 16 * u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) cv[u][v] = fma(-pi[u], pj[v], cv[u][v]);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int col = j0 + ty + 16 * v;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = i0 + tx + 16 * u;
                if (row < nrows && col < ncols && row >= col) A[(size_t)col * ld + row] = cv[u][v];
            }
        }
        return;
    }
    // ---- panel role for block column k1 ----------------------------------------------------------------------------------
    double *Pd = lds;                        // Pd[m * NB + r]   = panel k0, column m, row k1 + r (the diagonal block's rows)
    double *Pr = Pd + NB * NB;               // Pr[m * 64 + rr]  = panel k0, column m, this workgroup's row rr
    double *Xs = Pr + NB * 64;               // Xs[c * 64 + rr]  = block column k1, column c, this workgroup's row rr
    double *Dd = Xs + NB * 64;               // Dd[r * (NB + 1) + c]: the diagonal block, full symmetric square
    double *Li = Dd + NB * (NB + 1);         // Li[k * NB + c]   = Linv[c][k]
    double *Lb = Li + NB * NB;               // WaveFactor's exchange
    const bool diag_wg = blockIdx.x == 0;
    const int row0 = k1 + NB + 64 * ((int)blockIdx.x - 1);           // (row-panel workgroups) first of the 64 rows
    {
        // every load first, then the LDS stores (written as a load-store loop the round trips ran one after the other)
        double xv[NB * 64 / 256], prv[NB * 64 / 256], pdv[NB * NB / 256], dv[NB * NB / 256];
        if (!diag_wg) {
#pragma unroll
            for (int i = 0; i < NB * 64 / 256; ++i) {
                const int e = tid + 256 * i, rw = row0 + (e & 63);
                xv[i] = rw < nrows ? A[(size_t)(k1 + (e >> 6)) * ld + rw] : 0.0;
                if (have_prev) prv[i] = rw < nrows ? A[(size_t)(k0 + (e >> 6)) * ld + rw] : 0.0;
            }
        }
#pragma unroll
        for (int i = 0; i < NB * NB / 256; ++i) {
            const int e = tid + 256 * i, c = e / NB, r = e % NB;
            dv[i] = A[(size_t)(k1 + c) * ld + k1 + r];                                    // (the lower triangle is what counts: c <= r)
            if (have_prev) pdv[i] = A[(size_t)(k0 + c) * ld + k1 + r];
        }
        if (!diag_wg) {
#pragma unroll
            for (int i = 0; i < NB * 64 / 256; ++i) { const int e = tid + 256 * i; Xs[e] = xv[i]; if (have_prev) Pr[e] = prv[i]; }
        }
#pragma unroll
        for (int i = 0; i < NB * NB / 256; ++i) {
            const int e = tid + 256 * i, c = e / NB, r = e % NB;
            if (c <= r) { Dd[r * (NB + 1) + c] = dv[i]; Dd[c * (NB + 1) + r] = dv[i]; }
            if (have_prev) Pd[e] = pdv[i];
        }
    }
    __syncthreads();
    // (in place: workgroup 0 overwrites the diagonal block every row workgroup has just loaded -- it waits for their count, chol2.hpp)
    if (!diag_wg && tid == 0) __hip_atomic_fetch_add(info + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (have_prev) {
        // D -= P_d P_d' on the whole square (both triangles come out bitwise equal: the same products in the same order)
        const int r2 = 2 * (tid & 15), c2 = 2 * (tid >> 4);
        double d00 = Dd[r2 * (NB + 1) + c2], d01 = Dd[r2 * (NB + 1) + c2 + 1], d10 = Dd[(r2 + 1) * (NB + 1) + c2], d11 = Dd[(r2 + 1) * (NB + 1) + c2 + 1];
#pragma unroll 8
        for (int m = 0; m < NB; ++m) {
            const double2 pr = *reinterpret_cast<const double2 *>(Pd + m * NB + r2), pc = *reinterpret_cast<const double2 *>(Pd + m * NB + c2);
            d00 = fma(-pr.x, pc.x, d00); d01 = fma(-pr.x, pc.y, d01); d10 = fma(-pr.y, pc.x, d10); d11 = fma(-pr.y, pc.y, d11);
        }
        Dd[r2 * (NB + 1) + c2] = d00; Dd[r2 * (NB + 1) + c2 + 1] = d01; Dd[(r2 + 1) * (NB + 1) + c2] = d10; Dd[(r2 + 1) * (NB + 1) + c2 + 1] = d11;
        __syncthreads();
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (wave == 0) {
        // (the launch's critical path is THIS wave's dependent chain -- a lone fp64 wave, issue-bound: it must not queue behind the
        //  inverse-factor / trailing-update waves that share its SIMD)
        __builtin_amdgcn_s_setprio(3);
        double a[NB];
        const int c = lane & (NB - 1);
#pragma unroll
        for (int r = 0; r < NB; ++r) a[r] = lane < NB ? Dd[c * (NB + 1) + r] : (r == c ? 1.0 : 0.0);
        double lv0[NB];
        const bool ok = WaveFactor<0>::run(a, lane_bcast<0>(a[0]), 0.0, lv0, Lb, lane);
        if (lane >= NB) {
#pragma unroll
            for (int j = 0; j < NB; j += 2) *reinterpret_cast<double2 *>(Li + c * NB + j) = make_double2(a[j], a[j + 1]);      // Li[k = c][j] = Linv[j][c]
        }
        if (diag_wg) {
            if (!ok && lane == 0) atomicMax(info, k1 + 1);
            while (__hip_atomic_load(info + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < arrive_target) __builtin_amdgcn_s_sleep(2);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (lane < NB) A[(size_t)(k1 + j) * ld + k1 + c] = (j <= c) ? a[j] : 0.0;       // L[c][j]
                else Dinv[(size_t)(k1 / NB) * NB * NB + j * NB + c] = a[j];                      // Linv[j][c]
            }
        }
    } else if (wave >= 2 && have_prev && !diag_wg) {
        // B = A_i - P_i P_d' for the 64 rows, in place in Xs: 4 x 4 tiles over the 128 threads of waves 2-3
        const int t2 = tid - 128, rq = 4 * (t2 & 15), cq = 4 * (t2 >> 4);
        double acc[4][4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const double2 lo = *reinterpret_cast<const double2 *>(Xs + (cq + v) * 64 + rq), hi = *reinterpret_cast<const double2 *>(Xs + (cq + v) * 64 + rq + 2);
            acc[0][v] = lo.x; acc[1][v] = lo.y; acc[2][v] = hi.x; acc[3][v] = hi.y;
        }
#pragma unroll 4
        for (int m = 0; m < NB; ++m) {
            const double2 p0 = *reinterpret_cast<const double2 *>(Pr + m * 64 + rq), p1 = *reinterpret_cast<const double2 *>(Pr + m * 64 + rq + 2);
            const double2 q0 = *reinterpret_cast<const double2 *>(Pd + m * NB + cq), q1 = *reinterpret_cast<const double2 *>(Pd + m * NB + cq + 2);
            const double pr[4] = {p0.x, p0.y, p1.x, p1.y}, pc[4] = {q0.x, q0.y, q1.x, q1.y};
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = fma(-pr[u], pc[v], acc[u][v]);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            *reinterpret_cast<double2 *>(Xs + (cq + v) * 64 + rq) = make_double2(acc[0][v], acc[1][v]);
            *reinterpret_cast<double2 *>(Xs + (cq + v) * 64 + rq + 2) = make_double2(acc[2][v], acc[3][v]);
        }
    }
    if (diag_wg) return;
    __syncthreads();
    // ---- X = B Linv':  X[rr][c] = sum_{k <= c} B[rr][k] Linv[c][k]; thread: rows rq..rq+3, columns c2, c2 + 1 ------------------
    {
        const int rq = 4 * (tid & 15), c2 = 2 * (tid >> 4);
        const int kmax = 2 * (4 * wave + 3) + 1;                 // the wave's largest column (Linv is lower triangular: exact zeros beyond)
        double acc[4][2] = {};
#pragma unroll 4
        for (int k = 0; k <= kmax; ++k) {
            const double2 b0 = *reinterpret_cast<const double2 *>(Xs + k * 64 + rq), b1 = *reinterpret_cast<const double2 *>(Xs + k * 64 + rq + 2);
            const double2 li = *reinterpret_cast<const double2 *>(Li + k * NB + c2);
            acc[0][0] = fma(b0.x, li.x, acc[0][0]); acc[0][1] = fma(b0.x, li.y, acc[0][1]);
            acc[1][0] = fma(b0.y, li.x, acc[1][0]); acc[1][1] = fma(b0.y, li.y, acc[1][1]);
            acc[2][0] = fma(b1.x, li.x, acc[2][0]); acc[2][1] = fma(b1.x, li.y, acc[2][1]);
            acc[3][0] = fma(b1.y, li.x, acc[3][0]); acc[3][1] = fma(b1.y, li.y, acc[3][1]);
        }
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rw = row0 + rq + u;
                if (rw < nrows) A[(size_t)(k1 + c2 + v) * ld + rw] = acc[u][v];
            }
    }
}

// L' x = y with y = row nr of the factored array; x -> out[0..n).  One workgroup of 1024.  LDS: x[nr] | d[NB]
constexpr int CH_SOLVE_THREADS = 1024;
__global__ void __launch_bounds__(CH_SOLVE_THREADS)
chol_back_kernel(const double *__restrict__ L, int ld, int nr, int n, const double *__restrict__ Dinv, double *__restrict__ out)
{
    constexpr int NB = CH_NB, CW = NB / 16;       // columns per wave
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *x = lds, *dsum = lds + nr, *Dblk = dsum + NB;          // Dblk[2][NB * NB]: the inverse diagonal blocks, staged a step ahead
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < nr; i += blockDim.x) x[i] = L[(size_t)i * ld + nr];
    Dblk[tid] = Dinv[(size_t)(nr / NB - 1) * NB * NB + tid];       // (NB * NB == CH_SOLVE_THREADS)
    __syncthreads();
    int buf = 0;
    for (int k0 = nr - NB; k0 >= 0; k0 -= NB) {
        // (the next step's inverse block does not depend on x: it travels while this step's dots are formed, instead of 32
        //  dependent global loads inside the block solve)
        if (k0 >= NB) Dblk[(buf ^ 1) * NB * NB + tid] = Dinv[(size_t)(k0 / NB - 1) * NB * NB + tid];
        // wave w: columns k0 + 2 w, k0 + 2 w + 1 of L dotted with the part of x already solved
        {
            static_assert(CW == 2 && NB * NB == CH_SOLVE_THREADS, "two columns per wave; one inverse-block entry per thread");
            const int c0 = k0 + 2 * wave;
            double p0 = 0.0, p1 = 0.0;
            const double *L0 = L + (size_t)c0 * ld, *L1 = L0 + ld;
            int j = k0 + NB + lane;
            for (; j + 448 < nr; j += 512) {              // eight independent row strips of both columns in flight per pass
                double a[8], b[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) { a[s] = L0[j + 64 * s]; b[s] = L1[j + 64 * s]; }
#pragma unroll
                for (int s = 0; s < 8; ++s) { const double xs = x[j + 64 * s]; p0 = fma(a[s], xs, p0); p1 = fma(b[s], xs, p1); }
            }
            for (; j + 192 < nr; j += 256) {              // four
                const double a0 = L0[j], a1 = L0[j + 64], a2 = L0[j + 128], a3 = L0[j + 192];
                const double b0 = L1[j], b1 = L1[j + 64], b2 = L1[j + 128], b3 = L1[j + 192];
                p0 = fma(a0, x[j], p0); p0 = fma(a1, x[j + 64], p0); p0 = fma(a2, x[j + 128], p0); p0 = fma(a3, x[j + 192], p0);
                p1 = fma(b0, x[j], p1); p1 = fma(b1, x[j + 64], p1); p1 = fma(b2, x[j + 128], p1); p1 = fma(b3, x[j + 192], p1);
            }
            for (; j < nr; j += 64) {
                const double xj = x[j];
                p0 = fma(L0[j], xj, p0);
                p1 = fma(L1[j], xj, p1);
            }
            p0 = wave_allsum(p0); p1 = wave_allsum(p1);
            if (lane == 0) { dsum[2 * wave] = p0; dsum[2 * wave + 1] = p1; }
        }
        __syncthreads();
        if (wave == 0 && lane < NB) {
            // x_r = sum_{c >= r} Linv[c][r] (y_c - d_c)
            const double *D = Dblk + buf * NB * NB;
            double acc = 0.0;
#pragma unroll 8
            for (int c = 0; c < NB; ++c) acc = fma(D[c * NB + lane], x[k0 + c] - dsum[c], acc);
            x[k0 + lane] = acc;
        }
        __syncthreads();
        buf ^= 1;
    }
    for (int i = tid; i < n; i += blockDim.x) out[i] = x[i];
}

// x = L^-T y through the inverse factor (above): y = row nr of the factored array (the forward substitution the factorisation
// carried along), x -> out[0..n).  One wave per column c of W (block jb = c / NB):
//     x_c = sum_{j >= c in the diagonal block} Linv_jb[j][c] y_j  +  sum_{k in blocks jb < i <= nb - 2} W[k][c] y_k
//           +  sum_r R[(nb - 1) NB + r][c] u_r,      u = Linv_last' y_last   (the last block row, never formed: W_last = Linv_last R_last)
// contiguous reads down column c, a fixed summation order (lane-strided partial sums, one butterfly): ~5 us for 1024 columns.
constexpr int CH_WT_THREADS = 256;
__global__ void __launch_bounds__(CH_WT_THREADS)
chol_wt_kernel(const double *__restrict__ yv, int ys, int nr, int n, const double *__restrict__ Dinv, const double *__restrict__ Wm,
               const double *__restrict__ Rm, int ldw, double *__restrict__ out)
{
    // y_k = yv[k ys]: row nr of the factored array (yv = A + nr, ys = ld) or a plain vector (ys = 1: chol_w_kernel's output)
    constexpr int NB = CH_NB;
    __shared__ double u_s[NB];
    extern __shared__ __attribute__((aligned(16))) double y_s[];      // [nr]: y, staged ONCE per workgroup
    const int nb = nr / NB, last = nb - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Round 6: y is a ROW of the column-major factored array (stride ld: one cache line per element).  Every wave used to gather
    // it again for its column -- 1024 line requests per wave, 64 MB of L2 traffic per launch for an 8 MB matrix: 13.9 us.  Now the
    // workgroup stages it in LDS (one round trip, every load issued at once), and the columns read it from there.
    for (int k = threadIdx.x; k < nr; k += CH_WT_THREADS) y_s[k] = yv[(size_t)k * ys];
    __syncthreads();
    if (threadIdx.x < NB) {                            // u_c = sum_j Linv_last[j][c] y_(last NB + j)
        double dv[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) dv[j] = Dinv[(size_t)last * NB * NB + j * NB + threadIdx.x];      // (all loads first; rows j < c are exact zeros)
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < NB; ++j) if (j >= (int)threadIdx.x) acc = fma(dv[j], y_s[last * NB + j], acc);
        u_s[threadIdx.x] = acc;
    }
    __syncthreads();
    const int c = (int)blockIdx.x * (CH_WT_THREADS / 64) + wave;
    if (c >= nr) return;
    const int jb = c / NB, cc = c - jb * NB;
    double acc = 0.0;
    if (jb == last) acc = lane == 0 ? u_s[cc] : 0.0;
    else {
        if (lane < NB && lane >= cc) acc = Dinv[(size_t)jb * NB * NB + lane * NB + cc] * y_s[jb * NB + lane];
        const double *Wc = Wm + (size_t)c * ldw;
        // (eight rows of the column requested before the first is used: the plain loop waited out one memory round trip per 64 rows;
        //  the order of the sum is unchanged)
        const int kend = last * NB;
        for (int k0 = (jb + 1) * NB + lane; k0 < kend; k0 += 64 * 8) {
            double w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = k0 + 64 * u < kend ? Wc[k0 + 64 * u] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) if (k0 + 64 * u < kend) acc = fma(w[u], y_s[k0 + 64 * u], acc);
        }
        if (lane < NB) acc = fma(Rm[(size_t)c * ldw + last * NB + lane], u_s[lane], acc);
    }
    acc = wave_allsum(acc);
    if (lane == 0 && c < n) out[c] = acc;
}

// y = W g = L^-1 g for a NEW right-hand side against the factor (and inverse factor) of an earlier factorisation: with
// chol_wt_kernel behind it,  x = W' (W g) = H_old^-1 g  costs two matrix-vector products (~20 us) where a fresh factorisation
// costs ~400 -- the "chord" steps of the second-order iteration (cfmm_hip.hip: solve_newton).  One thread per row k (adjacent
// rows are adjacent in the column-major W: coalesced), a fixed summation order:
//     k in block i <= nb - 2:   y_k = sum_{c < i NB} W[k][c] g_c + sum_{c <= k in the block} Linv_i[k][c] g_c
//     last block:               y = Linv_last (g_last + sum_{c < last NB} R[last rows][c] g_c)
// g is masked (pinned tokens carry identity rows in the factored matrix: their g is what comes out).
constexpr int CH_W_ROWS = 64, CH_W_SLICES = 16, CH_W_THREADS = CH_W_ROWS * CH_W_SLICES;
__global__ void __launch_bounds__(CH_W_THREADS)
chol_w_kernel(const double *__restrict__ g, int nr, int n, const double *__restrict__ Dinv, const double *__restrict__ Wm,
              const double *__restrict__ Rm, int ldw, double *__restrict__ y)
{
    // 64 rows per workgroup, the columns dealt to 16 waves in groups of four (wave s: columns 4 s + 64 t ...): a row's sum is
    // 1000 terms long, one thread per row with four loads in flight took 85 us for the 16 workgroups; partial sums meet in LDS
    // and are added in wave order
    constexpr int NB = CH_NB;
    __shared__ double t_s[CH_W_ROWS];
    __shared__ double part[CH_W_SLICES][CH_W_ROWS];
    const int nb = nr / NB, last = nb - 1;
    const int r = threadIdx.x & (CH_W_ROWS - 1), sl = threadIdx.x / CH_W_ROWS;
    const int k = (int)blockIdx.x * CH_W_ROWS + r;
    const int ib = k / NB, kk = k - ib * NB;
    const bool islast = ib == last;
    const double *M = islast ? Rm : Wm;                       // strictly-lower part: finished W rows, or the last row's residual
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const int cend = ib * NB;                                 // (a multiple of 4)
    if (k < nr) {
        for (int c = 4 * sl; c + 4 <= cend; c += 4 * CH_W_SLICES) {
            a0 = fma(M[(size_t)c * ldw + k], c < n ? g[c] : 0.0, a0);
            a1 = fma(M[(size_t)(c + 1) * ldw + k], c + 1 < n ? g[c + 1] : 0.0, a1);
            a2 = fma(M[(size_t)(c + 2) * ldw + k], c + 2 < n ? g[c + 2] : 0.0, a2);
            a3 = fma(M[(size_t)(c + 3) * ldw + k], c + 3 < n ? g[c + 3] : 0.0, a3);
        }
    }
    part[sl][r] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    double acc = 0.0;
    if (sl == 0) {
#pragma unroll
        for (int q = 0; q < CH_W_SLICES; ++q) acc += part[q][r];
        if (islast && k < nr) acc += k < n ? g[k] : 0.0;      // t = g_last + R_last g
        t_s[r] = islast ? acc : (k < n ? g[k] : 0.0);         // the diagonal block's input: t (last block) or g (the others)
    }
    __syncthreads();
    if (sl != 0 || k >= nr) return;
    const int base = (r / NB) * NB;                           // this row's block inside the workgroup's 64 rows
    double d = 0.0;
    for (int cc = 0; cc <= kk; ++cc) d = fma(Dinv[(size_t)ib * NB * NB + kk * NB + cc], t_s[base + cc], d);
    y[k] = islast ? d : acc + d;
}

}  // namespace cfmm
