// Dense Cholesky, TWO block columns per launch (round 4): chol_step2_kernel.
//
// chol.hpp's scheme costs one launch per 32-wide block column, and at n = 1000 every one of its 32 launches is ~4.4 us of
// dependent-launch overhead + ~7 us of chain (stage the operands, update and factor the diagonal block in one wave, form the
// panel rows as a product with its inverse).  The overhead is per LAUNCH, the chain per BLOCK COLUMN: this kernel takes the
// block columns in pairs (a, b = a + 32) -- 16 launches -- with the same side-by-side roles:
//
//   panel     workgroup w > 0 owns 64 rows under the pair's 64 x 64 diagonal region, workgroup 0 only the region itself.
//             Everything the pair needs of the previous pair's update is applied by the panel workgroups themselves (the
//             diagonal region redundantly in each, like its factorisation), so no workgroup waits for another:
//               1  D_aa -= P_a P_a'                                         all four waves                      (chain)
//               2  wave 0: factor + invert D_aa  ||  waves 1-3: D_ba, D_bb, X_a -= (previous pair)
//               3  L_ba = D_ba Linv_a',  D_bb -= L_ba L_ba'                 all four waves                      (chain)
//               4  wave 0: factor + invert D_bb  ||  waves 1-3: X_b -= (previous pair), X_a = X_a Linv_a', X_b -= X_a L_ba'
//               5  X_b = X_b Linv_b'                                        all four waves                      (chain)
//             (X_a, X_b: the workgroup's 64 rows of block columns a and b; P: its rows of the previous pair's 64 columns.)
//   trailing  64 x 64 tiles of the columns from a + 64 on take the PREVIOUS pair's rank-64 update (one launch behind, as before)
//   inverse   the inverse factor W = L^-1 (chol.hpp) advances by the previous pair's TWO block rows per launch: workgroup (i, j)
//             forms W_aj = Linv_a R_aj, carries it into row b (R_bj -= L_ba W_aj), forms W_bj = Linv_b R_bj and gives both to
//             its own block, R_ij -= L_ia W_aj + L_ib W_bj -- five 32^3 products where two launches of the old kernel did four.
//             One more launch with this role alone finishes the last-but-one block row behind the factorisation.
//
// Fixed summation orders throughout (bitwise the same on every rank of a pool-sharded solve).  nr is a multiple of 64.
#pragma once
#include "chol.hpp"

namespace cfmm {

constexpr int CH2_ROWS = 32;             // rows of a panel workgroup (64: the side waves' products outlast wave 0's factorisations)
constexpr int CH2_PANEL_LDS = 2 * 2048 + 64 * CH2_ROWS + 2 * CH_NB * CH2_ROWS + 2 * (CH_NB * (CH_NB + 1)) + 6 * CH_NB * CH_NB + 128 + 8;      // doubles
constexpr int CH2_TILE_LDS = 2 * 64 * 65;
constexpr int CH2_W_LDS = 9 * CH_NB * CH_NB;
constexpr int CH2_LDS_DOUBLES = CH2_PANEL_LDS > CH2_TILE_LDS ? (CH2_PANEL_LDS > CH2_W_LDS ? CH2_PANEL_LDS : CH2_W_LDS) : (CH2_TILE_LDS > CH2_W_LDS ? CH2_TILE_LDS : CH2_W_LDS);

// acc[u][v] (TM x TN, rows m0.., columns n0..) += sign * sum_{k < kend} Ak[k lda + m0 + u] Bk[k ldb + n0 + v]: both operands k-major in
// LDS, so a step is TM / 2 + TN / 2 16-byte reads.  m0, n0, lda, ldb even.
template <int TM, int TN, bool NEG>
__device__ __forceinline__ void mm_acc(double (&acc)[TM][TN], const double *Ak, int lda, const double *Bk, int ldb, int m0, int n0, int kend)
{
#pragma unroll 4
    for (int k = 0; k < kend; ++k) {
        double av[TM], bv[TN];
#pragma unroll
        for (int u = 0; u < TM; u += 2) { const double2 t = *reinterpret_cast<const double2 *>(Ak + k * lda + m0 + u); av[u] = t.x; av[u + 1] = t.y; }
#pragma unroll
        for (int v = 0; v < TN; v += 2) { const double2 t = *reinterpret_cast<const double2 *>(Bk + k * ldb + n0 + v); bv[v] = t.x; bv[v + 1] = t.y; }
#pragma unroll
        for (int u = 0; u < TM; ++u)
#pragma unroll
            for (int v = 0; v < TN; ++v) acc[u][v] = fma(NEG ? -av[u] : av[u], bv[v], acc[u][v]);
    }
}

// One 16 x 16 tile by ONE wave on the matrix pipe (v_mfma_f64_16x16x4_f64: 1024 fmas per instruction at the vector rate, operands
// ONE f64 per lane):  C[(n0 + i) ldc + m0 + j] += sign * sum_{k < K} Ak[k lda + m0 + j] Bk[k ldb + n0 + i],  i, j < 16, K a multiple of 4.
// The instruction's A operand carries OUR column operand and its B operand our row operand, so that the accumulator's lane index
// (col = lane & 15) runs along the contiguous m of C: loads and stores of C are 128-byte runs, not a 16-way bank conflict.
// Why the matrix pipe here: the side waves work while wave 0 runs the factorisation's dependent chain, whose multipliers travel
// through LDS -- as 4 x 4 register tiles on the vector pipe these products kept the LDS port saturated (4 KB of operands per 1024
// fmas; 1 KB here) and the chain took 2.5x as long.
typedef double ch2_f64x4 __attribute__((ext_vector_type(4)));
#ifdef CFMM_CH2_STAMPS
__device__ unsigned long long g_ch2_stamps[64];
#define CH2_STAMP(i) do { if (c0 == 512 && blockIdx.x == CFMM_CH2_STAMP_WG && (tid & 63) == 0) g_ch2_stamps[(i) + 16 * (tid >> 6)] = wall_clock64(); } while (0)
#else
#define CH2_STAMP(i) do { } while (0)
#endif
// K4 = K / 4 steps, ALL operands of the tile requested before the first product: a step-by-step loop waited out an LDS round
// trip (~100 cycles) in front of every 64-cycle instruction.  `ksteps` <= K4 (a runtime bound for the triangular products: the
// steps beyond it are skipped, wave-uniformly).
// NACC independent accumulators (step q goes to accumulator q % NACC; summed at the end, in order): a chain of DEPENDENT products
// runs at the instruction's latency, not its issue rate -- what the phases on the panel's critical path (one tile per wave) pay.
template <int K4, bool NEG, bool ZERO, int NACC = 2>
__device__ __forceinline__ ch2_f64x4 mfma_tile(const double *C, int ldc, const double *Ak, int lda, const double *Bk, int ldb, int m0, int n0, int ksteps, int lane)
{
    const int lj = lane & 15, lk = lane >> 4;
    const double *pa = Bk + lk * ldb + n0 + lj, *pb = Ak + lk * lda + m0 + lj;
    double av[K4], bv[K4];
#pragma unroll
    for (int q = 0; q < K4; ++q) { av[q] = q < ksteps ? pa[4 * q * ldb] : 0.0; bv[q] = q < ksteps ? pb[4 * q * lda] : 0.0; }
    ch2_f64x4 acc[NACC];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[0][r] = ZERO ? 0.0 : C[(n0 + lk + 4 * r) * ldc + m0 + lj];
#pragma unroll
    for (int u = 1; u < NACC; ++u) acc[u] = ch2_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < K4; ++q)
        if (q < ksteps) acc[q % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -av[q] : av[q], bv[q], acc[q % NACC], 0, 0, 0);
#pragma unroll
    for (int u = 1; u < NACC; ++u) acc[0] += acc[u];
    return acc[0];
}
__device__ __forceinline__ void mfma_store(double *C, int ldc, int m0, int n0, int lane, const ch2_f64x4 &acc)
{
    const int lj = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(n0 + lk + 4 * r) * ldc + m0 + lj] = acc[r];
}
// acc += sign * (the same 16 x 16 x 4 K4 product), for products that continue an accumulator held in registers (the side roles below)
template <int K4, bool NEG>
__device__ __forceinline__ void mfma_into(ch2_f64x4 &acc, const double *Ak, int lda, const double *Bk, int ldb, int m0, int n0, int lane)
{
    const int lj = lane & 15, lk = lane >> 4;
    const double *pa = Bk + lk * ldb + n0 + lj, *pb = Ak + lk * lda + m0 + lj;
    double av[K4], bv[K4];
#pragma unroll
    for (int q = 0; q < K4; ++q) { av[q] = pa[4 * q * ldb]; bv[q] = pb[4 * q * lda]; }
#pragma unroll
    for (int q = 0; q < K4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -av[q] : av[q], bv[q], acc, 0, 0, 0);
}

__global__ void __launch_bounds__(256)
chol_step2_kernel(double *__restrict__ A, int ld, int nrows, int ncols, int c0, int npanel, double *__restrict__ Dinv, int *__restrict__ info,
                  int nfac, double *__restrict__ Wm, double *__restrict__ Rm, int ldw, int w_single, int ntask, int arrive_target)
{
    constexpr int NB = CH_NB, NB1 = CH_NB + 1;
    extern __shared__ __attribute__((aligned(16))) double lds2[];
    const int tid = threadIdx.x;
    const bool have_prev = c0 > 0;
    const int pa = c0 - 2 * NB;                                   // the previous pair's first column
    // side roles: `ntask` tasks (inverse-factor tiles first: the longer ones) dealt over the gridDim.x - npanel workgroups behind the
    // panel's -- a workgroup takes 149 KB of LDS, ONE per CU: the host never asks for more workgroups than CUs, a task beyond
    // that rides with an earlier one (the chain of the panel workgroups is longer than two tasks)
    const int ntw = nfac - npanel;
    CH2_STAMP(9);
    if ((int)blockIdx.x >= npanel)
    for (int task = (int)blockIdx.x - npanel, nth = 0; task < ntask; task += (int)gridDim.x - npanel, ++nth) {
    CH2_STAMP(10 + 2 * nth);
    if (task < ntw) {
        // ---- inverse factor: block rows qa (and qb = qa + 1), tile (i, j) ---------------------------------------------------------
        const int qa = w_single ? ncols / NB - 2 : c0 / NB - 2, qb = qa + 1;
        const int ncol_j = w_single ? qa + 1 : qb + 1;
        const int t = task, j = t % ncol_j, i = (w_single ? qa : qb) + 1 + t / ncol_j;
        double *Lia = lds2, *Lib = Lia + NB * NB, *Ra = Lib + NB * NB, *Rb = Ra + NB * NB, *Lba = Rb + NB * NB, *Lxa = Lba + NB * NB,
               *Lxb = Lxa + NB * NB, *Wa = Lxb + NB * NB, *Wb = Wa + NB * NB;
        const bool two = !w_single;
        const bool wa_zero = j > qa;                              // (j == qb: row a has no block in column b)
        {
            double va[4], vb[4], ra[4], rb[4], lba[4], lxa[4], lxb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                         // (all loads first)
                const int e = tid + 256 * u, rr = e & 31, cc = e >> 5;
                va[u] = Dinv[(size_t)qa * NB * NB + e];
                vb[u] = two ? Dinv[(size_t)qb * NB * NB + e] : 0.0;
                ra[u] = j == qa ? (rr == cc ? 1.0 : 0.0) : (j < qa ? Rm[(size_t)(j * NB + cc) * ldw + qa * NB + rr] : 0.0);
                // (a block's first update comes from row j and starts from zero: R_b,a has had none yet -- never read what an
                //  earlier factorisation left there)
                rb[u] = !two ? 0.0 : (j == qb ? (rr == cc ? 1.0 : 0.0) : (j == qa ? 0.0 : Rm[(size_t)(j * NB + cc) * ldw + qb * NB + rr]));
                lba[u] = two ? A[(size_t)(qa * NB + cc) * ld + qb * NB + rr] : 0.0;
                lxa[u] = A[(size_t)(qa * NB + cc) * ld + i * NB + rr];
                lxb[u] = two ? A[(size_t)(qb * NB + cc) * ld + i * NB + rr] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {                         // every operand k-major (chol.hpp)
                const int e = tid + 256 * u, rr = e & 31, cc = e >> 5;
                Lia[rr * NB + cc] = va[u]; Lib[rr * NB + cc] = vb[u];          // Dinv[q][e]: e = row NB + col -> [k = col = rr][r = row = cc]
                Ra[rr * NB + cc] = ra[u]; Rb[rr * NB + cc] = rb[u];            // R[rr][cc] -> [k = rr][c = cc]
                Lba[cc * NB + rr] = lba[u]; Lxa[cc * NB + rr] = lxa[u]; Lxb[cc * NB + rr] = lxb[u];      // L[rr][cc] -> [k = cc][r = rr]
            }
        }
        // Round 6: the five 32^3 products on the matrix pipe, one 16 x 16 tile per wave (rows rt .., columns ct ..).  As 2 x 2 register
        // tiles on the vector pipe a product read 64 KB of LDS operands per workgroup (0.9 us at the port's 128 B / clk: the task was
        // LDS-bound at 7.5 us, and the workgroups that take TWO side tasks ended 2.7 us behind the panel's chain in the eight launches
        // with a full grid); here it reads 8 KB per wave.  Orientation per product: the accumulator's lanes run along the contiguous
        // index of whatever the product is stored into (mfma_tile's convention).
        {
            const int wv = tid >> 6, ln = tid & 63, lj = ln & 15, lk = ln >> 4;
            const int rt = 16 * (wv & 1), ct = 16 * (wv >> 1);
            const bool store_rows = i == (w_single ? qa : qb) + 1;
            // R_ij's own tile, requested now (rows contiguous in memory: lanes along r)
            ch2_f64x4 x;
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = j >= qa ? 0.0 : Rm[(size_t)(j * NB + ct + lk + 4 * r) * ldw + i * NB + rt + lj];
            __syncthreads();
            ch2_f64x4 w = {0.0, 0.0, 0.0, 0.0};
            if (!wa_zero) mfma_into<8, false>(w, Ra, NB, Lia, NB, ct, rt, ln);                 // W_aj = Linv_a R_aj, as Wa[r][c]
            mfma_store(Wa, NB, ct, rt, ln, w);
            if (store_rows && j < qa) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Wm[(size_t)(j * NB + ct + lj) * ldw + qa * NB + rt + lk + 4 * r] = w[r];
            }
            __syncthreads();
            if (two) {
                // R_bj -= L_ba W_aj in place: a wave reads and writes its own tile of Rb only
                ch2_f64x4 rbv;
#pragma unroll
                for (int r = 0; r < 4; ++r) rbv[r] = Rb[(rt + lk + 4 * r) * NB + ct + lj];
                if (!wa_zero) mfma_into<8, true>(rbv, Wa, NB, Lba, NB, ct, rt, ln);
                mfma_store(Rb, NB, ct, rt, ln, rbv);
                __syncthreads();
                ch2_f64x4 wb = {0.0, 0.0, 0.0, 0.0};
                mfma_into<8, false>(wb, Rb, NB, Lib, NB, ct, rt, ln);                          // W_bj = Linv_b R_bj
                mfma_store(Wb, NB, ct, rt, ln, wb);
                if (store_rows && j < qb) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Wm[(size_t)(j * NB + ct + lj) * ldw + qb * NB + rt + lk + 4 * r] = wb[r];
                }
                __syncthreads();
            }
            if (!wa_zero) mfma_into<8, true>(x, Lxa, NB, Wa, NB, rt, ct, ln);                  // R_ij -= L_ia W_aj + L_ib W_bj
            if (two) mfma_into<8, true>(x, Lxb, NB, Wb, NB, rt, ct, ln);
#pragma unroll
            for (int r = 0; r < 4; ++r) Rm[(size_t)(j * NB + ct + lk + 4 * r) * ldw + i * NB + rt + lj] = x[r];
        }
    } else {
        // ---- trailing update with the previous pair over rows / columns >= c0 + 64: 64 x 64 tile (ti, tj) of the lower triangle ----
        double *Pi = lds2, *Pj = Pi + 64 * 65;                    // [c][rr], 65-double rows
        int t = task - ntw, ti = 0;
        while (t > ti) { t -= ti + 1; ++ti; }
        const int tj = t;
        const int base = c0 + 2 * NB;
        const int i0 = base + 64 * ti, j0 = base + 64 * tj;
        // (round 6: on the matrix pipe too -- sixteen 16 x 16 tiles, four per wave, k = 64; as 4 x 4 register tiles the 64 steps read
        //  512 LDS words per thread: 3.5 us of LDS port for 1.7 us of arithmetic)
        {
            double pv[16], qv[16];                                // (every load first: chol.hpp)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = tid + 256 * q, c = e >> 6, rr = e & 63;
                pv[q] = (i0 + rr < nrows) ? A[(size_t)(pa + c) * ld + i0 + rr] : 0.0;
                qv[q] = (j0 + rr < ncols) ? A[(size_t)(pa + c) * ld + j0 + rr] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = tid + 256 * q, c = e >> 6, rr = e & 63;
                Pi[c * 65 + rr] = pv[q]; Pj[c * 65 + rr] = qv[q];
            }
        }
        __syncthreads();
        {
            const int wv = tid >> 6, ln = tid & 63, lj = ln & 15, lk = ln >> 4;
#pragma nounroll
            for (int tt = wv; tt < 16; tt += 4) {
                const int mt = 16 * (tt & 3), nt = 16 * (tt >> 2);               // rows i0 + mt .., columns j0 + nt ..
                if (ti == tj && mt < nt) continue;                                // (wholly above the diagonal)
                ch2_f64x4 acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i0 + mt + lj, col = j0 + nt + lk + 4 * r;
                    acc[r] = (row < nrows && col < ncols && row >= col) ? A[(size_t)col * ld + row] : 0.0;
                }
                mfma_into<16, true>(acc, Pi, 65, Pj, 65, mt, nt, ln);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i0 + mt + lj, col = j0 + nt + lk + 4 * r;
                    if (row < nrows && col < ncols && row >= col) A[(size_t)col * ld + row] = acc[r];
                }
            }
        }
    }
    __syncthreads();                                            // (the next task reuses the LDS)
    CH2_STAMP(11 + 2 * nth);
    }
    if ((int)blockIdx.x >= npanel) return;
    // ---- panel role for the pair (a, b) = (c0, c0 + NB) -------------------------------------------------------------------------
    const int ca = c0, cb = c0 + NB;
    double *Pda = lds2;                      // Pda[k * 32 + r]: previous pair, column k, row ca + r.  After phase 2: X_a's result, Xo[c * RW + rr]
    double *Pdb = Pda + 2048;                // Pdb[k * 32 + r]: ... row cb + r
    constexpr int RW = CH2_ROWS;             // this workgroup's rows
    double *Pr = Pdb + 2048;                 // Pr[k * RW + rr]: previous pair, column k, this workgroup's row rr
    double *Xa = Pr + 64 * RW;               // Xa[c * RW + rr]: block column a, this workgroup's rows
    double *Xb = Xa + NB * RW;
    double *Daa = Xb + NB * RW;              // Daa[c * 33 + r]: full symmetric square
    double *Dbb = Daa + NB * NB1;
    double *Dba = Dbb + NB * NB1;            // Dba[c * 32 + r]: rows of b, columns of a (k-major for L_ba = D_ba Linv_a')
    double *Lia = Dba + NB * NB;             // Lia[k * 32 + c] = Linv_a[c][k]
    double *Lib = Lia + NB * NB;
    double *Lba = Lib + NB * NB;             // Lba[c * 32 + r] = L_ba[r][c]
    double *Lbx = Lba + NB * NB;             // WaveFactor's exchange (128)
    int *sub = reinterpret_cast<int *>(Lbx + 128);                // the side waves' own rendezvous (phase 4)
    double *Lfac = Lbx + 128 + 8;            // workgroup 0: the two factored diagonal blocks on their way out, Lfac[h][j * 32 + c] = L[c][j]
    double *Xo = Pda;
    const bool diag_wg = blockIdx.x == 0;
    const int row0 = c0 + 2 * NB + RW * ((int)blockIdx.x - 1);
    const int wave = tid >> 6, lane = tid & 63;
    if (tid == 0) *sub = 0;
    CH2_STAMP(0);
    {
        // every global load first, the LDS stores behind them (written as load-store loops the round trips ran one after the
        // other: 7.5 us in front of the chain).  Per thread: 12 of the diagonal region, 16 of the previous pair's rows of it,
        // 8 + 8 of the workgroup's own rows.
        constexpr int NX = NB * RW / 256, NP = 64 * RW / 256;
        double vaa[4], vbb[4], vba[4], pda[8], pdb[8], xa[NX], xb[NX], pr[NP];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q, c = e >> 5, r = e & 31;
            vaa[q] = A[(size_t)(ca + c) * ld + ca + r]; vbb[q] = A[(size_t)(cb + c) * ld + cb + r]; vba[q] = A[(size_t)(ca + c) * ld + cb + r];
        }
        if (have_prev) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = tid + 256 * q, k = e >> 5, r = e & 31;
                pda[q] = A[(size_t)(pa + k) * ld + ca + r]; pdb[q] = A[(size_t)(pa + k) * ld + cb + r];
            }
        }
        if (!diag_wg) {
#pragma unroll
            for (int q = 0; q < NX; ++q) {
                const int e = tid + 256 * q, c = e / RW, rw = row0 + (e % RW);
                xa[q] = rw < nrows ? A[(size_t)(ca + c) * ld + rw] : 0.0; xb[q] = rw < nrows ? A[(size_t)(cb + c) * ld + rw] : 0.0;
            }
            if (have_prev) {
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const int e = tid + 256 * q, k = e / RW, rw = row0 + (e % RW);
                    pr[q] = rw < nrows ? A[(size_t)(pa + k) * ld + rw] : 0.0;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q, c = e >> 5, r = e & 31;
            if (c <= r) { Daa[c * NB1 + r] = vaa[q]; Daa[r * NB1 + c] = vaa[q]; Dbb[c * NB1 + r] = vbb[q]; Dbb[r * NB1 + c] = vbb[q]; }
            Dba[c * NB + r] = vba[q];
        }
        if (have_prev) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { Pda[tid + 256 * q] = pda[q]; Pdb[tid + 256 * q] = pdb[q]; }
        }
        if (!diag_wg) {
#pragma unroll
            for (int q = 0; q < NX; ++q) { Xa[tid + 256 * q] = xa[q]; Xb[tid + 256 * q] = xb[q]; }
            if (have_prev) {
#pragma unroll
                for (int q = 0; q < NP; ++q) Pr[tid + 256 * q] = pr[q];
            }
        }
    }
    __syncthreads();
    CH2_STAMP(1);
    // The factorisation is IN PLACE: workgroup 0 writes L_aa, L_ba, L_bb over the very diagonal region every row workgroup has just loaded
    // as input.  That used to rest on timing alone -- all workgroups of a launch start within a microsecond, workgroup 0 stores ~5 us
    // later -- and broke when the device was shared (round 6: six fuzzers on one GPU; a row workgroup dispatched late loaded the FACTOR,
    // factored garbage, and the forward-substituted right-hand side came back NaN with a perfectly good factor beside it).  Now every row
    // workgroup counts itself in once its loads have landed (the barrier above), and workgroup 0 waits for the launch's full count before
    // its first in-place store.  info[2]: zeroed with the flag per factorisation; arrive_target: the running total the host passes.
    if (!diag_wg && tid == 0) __hip_atomic_fetch_add(info + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (relaxed: nothing this workgroup WROTE is handed over, and its loads completed in front of the barrier)
    // one wave factors and inverts a 32 x 32 block (chol.hpp: WaveFactor); workgroup 0 writes the factor and its inverse.
    // The two halves of the pair run through ONE copy of this code (the loop below is kept a loop): WaveFactor is ~1700
    // straight-line instructions, and a second inlined copy made the kernel outgrow the instruction cache -- every launch took
    // 33 us whatever its size, the first of a factorisation 52.
    int nhalf = 2;
    asm volatile("" : "+s"(nhalf));
#pragma nounroll
    for (int h = 0; h < nhalf; ++h) {
        double *Dd = h ? Dbb : Daa, *Li = h ? Lib : Lia;
        const int kcol = h ? cb : ca;
        if (h == 0) {
            // ---- 1: D_aa -= P_a P_a' on the whole square, one 16 x 16 tile per wave (both triangles bitwise equal: the same
            //         products in the same order) ----
            if (have_prev) {
                const int m0 = 16 * (wave & 1), n0 = 16 * (wave >> 1);
                const ch2_f64x4 acc = mfma_tile<16, true, false, 4>(Daa, NB1, Pda, NB, Pda, NB, m0, n0, 16, lane);
                mfma_store(Daa, NB1, m0, n0, lane, acc);
            }
        } else {
            // ---- 3: L_ba = D_ba Linv_a' (Linv lower triangular: columns n0 .. n0 + 15 need k < n0 + 16), then D_bb -= L_ba L_ba':
            //         one tile per wave each ----
            const int m0 = 16 * (wave & 1), n0 = 16 * (wave >> 1);
            const ch2_f64x4 l = mfma_tile<8, false, true, 4>(nullptr, 0, Dba, NB, Lia, NB, m0, n0, (n0 + 16) / 4, lane);
            mfma_store(Lba, NB, m0, n0, lane, l);
            if (diag_wg) {
                // (the first in-place store of the launch: not before every row workgroup has loaded the region -- see the prologue)
                while (__hip_atomic_load(info + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < arrive_target) __builtin_amdgcn_s_sleep(2);
                const int lj = lane & 15, lk = lane >> 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) A[(size_t)(ca + n0 + lk + 4 * r) * ld + cb + m0 + lj] = l[r];
            }
            __syncthreads();
            const ch2_f64x4 d = mfma_tile<8, true, false, 4>(Dbb, NB1, Lba, NB, Lba, NB, m0, n0, 8, lane);
            mfma_store(Dbb, NB1, m0, n0, lane, d);
        }
        __syncthreads();
        CH2_STAMP(2 + 3 * h);
        if (wave == 0) {
            // ---- 2 / 4 (wave 0): factor + invert the half's diagonal block ----
            double a[NB];
            const int c = lane & (NB - 1);
#pragma unroll
            for (int r = 0; r < NB; ++r) a[r] = lane < NB ? Dd[c * NB1 + r] : (r == c ? 1.0 : 0.0);
            double lv0[NB];
            const bool ok = WaveFactor<0>::run(a, lane_bcast<0>(a[0]), 0.0, lv0, Lbx, lane);
            if (lane >= NB) {
#pragma unroll
                for (int j = 0; j < NB; j += 2) *reinterpret_cast<double2 *>(Li + c * NB + j) = make_double2(a[j], a[j + 1]);      // Li[k = c][j] = Linv[j][c]
            }
            if (diag_wg) {
                // (the factor leaves through LDS: the other waves write it out -- 64 scattered global stores per lane at the end of
                //  BOTH factorisations made workgroup 0's chain 17 us where the row workgroups' is 13.8, and the launch lasts as long)
                if (!ok && lane == 0) atomicMax(info, kcol + 1);
                if (lane < NB) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) Lfac[h * NB * NB + j * NB + c] = (j <= c) ? a[j] : 0.0;       // L[c][j]
                }
            }
        } else if (h == 0) {
            // ---- 2 (waves 1-3): the previous pair's update of D_ba, D_bb and X_a: 16 tiles of 16 x 16, one wave each ----
            if (have_prev) {
                constexpr int MT = RW / 16;                             // tiles down the workgroup's rows
                const int ntile = diag_wg ? 8 : 8 + 2 * MT;
                for (int w = wave - 1; w < ntile; w += 3) {
                    if (w < 8) {                                        // D_ba (w < 4) or D_bb: 2 x 2 tiles
                        const bool bb = w >= 4;
                        const int m0 = 16 * (w & 1), n0 = 16 * ((w >> 1) & 1);
                        double *D = bb ? Dbb : Dba;
                        const int ldd = bb ? NB1 : NB;
                        const ch2_f64x4 acc = mfma_tile<16, true, false>(D, ldd, Pdb, NB, bb ? Pdb : Pda, NB, m0, n0, 16, lane);
                        mfma_store(D, ldd, m0, n0, lane, acc);
                    } else {                                            // X_a -= P_r P_a': MT x 2 tiles
                        const int q = w - 8, m0 = 16 * (q % MT), n0 = 16 * (q / MT);
                        const ch2_f64x4 acc = mfma_tile<16, true, false>(Xa, RW, Pr, RW, Pda, NB, m0, n0, 16, lane);
                        mfma_store(Xa, RW, m0, n0, lane, acc);
                    }
                }
            }
        } else if (diag_wg) {
            // ---- 4 (waves 1-3 of workgroup 0): half a's factor and inverse out to global memory ----
            for (int e = tid - 64; e < NB * NB; e += 192) {
                const int j = e >> 5, c = e & 31;
                A[(size_t)(ca + j) * ld + ca + c] = Lfac[e];
                Dinv[(size_t)(ca / NB) * NB * NB + e] = Lia[c * NB + j];                                       // Linv[j][c]
            }
        } else {
            // ---- 4 (waves 1-3): X_b -= (previous pair), X_a = X_a Linv_a', X_b -= X_a L_ba': MT x 2 tiles each ----
            constexpr int MT = RW / 16;
            for (int w = wave - 1; w < 2 * MT; w += 3) {
                const int m0 = 16 * (w % MT), n0 = 16 * (w / MT);
                if (have_prev) {
                    const ch2_f64x4 acc = mfma_tile<16, true, false>(Xb, RW, Pr, RW, Pdb, NB, m0, n0, 16, lane);
                    mfma_store(Xb, RW, m0, n0, lane, acc);
                }
                // X_a Linv_a': Linv is lower triangular, the tile's columns n0 .. n0 + 15 need k < n0 + 16
                const ch2_f64x4 xo = mfma_tile<8, false, true>(nullptr, 0, Xa, RW, Lia, NB, m0, n0, (n0 + 16) / 4, lane);
                mfma_store(Xo, RW, m0, n0, lane, xo);
                const int lj = lane & 15, lk = lane >> 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rw = row0 + m0 + lj;
                    if (rw < nrows) A[(size_t)(ca + n0 + lk + 4 * r) * ld + rw] = xo[r];
                }
            }
            // the three side waves meet (wave 0 is inside its factorisation: no workgroup barrier here)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) atomicAdd(sub, 1);
            while (__hip_atomic_load(sub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 3) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            for (int w = wave - 1; w < 2 * MT; w += 3) {
                const int m0 = 16 * (w % MT), n0 = 16 * (w / MT);
                const ch2_f64x4 acc = mfma_tile<8, true, false>(Xb, RW, Xo, RW, Lba, NB, m0, n0, 8, lane);
                mfma_store(Xb, RW, m0, n0, lane, acc);
            }
        }
        CH2_STAMP(3 + 3 * h);
        __syncthreads();
        CH2_STAMP(4 + 3 * h);
    }
    if (diag_wg) {                                                // half b's factor and inverse out
        for (int e = tid; e < NB * NB; e += 256) {
            const int j = e >> 5, c = e & 31;
            A[(size_t)(cb + j) * ld + cb + c] = Lfac[NB * NB + e];
            Dinv[(size_t)(cb / NB) * NB * NB + e] = Lib[c * NB + j];
        }
        return;
    }
    // ---- 5: X_b = X_b Linv_b': one 16 x 16 tile per wave (MT x 2 of them) ----
    {
        constexpr int MT = RW / 16;
        for (int w = wave; w < 2 * MT; w += 4) {
            const int m0 = 16 * (w % MT), n0 = 16 * (w / MT);
            const ch2_f64x4 x = mfma_tile<8, false, true, 4>(nullptr, 0, Xb, RW, Lib, NB, m0, n0, (n0 + 16) / 4, lane);
            const int lj = lane & 15, lk = lane >> 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rw = row0 + m0 + lj;
                if (rw < nrows) A[(size_t)(cb + n0 + lk + 4 * r) * ld + rw] = x[r];
            }
        }
    }
    CH2_STAMP(8);
}

}  // namespace cfmm
