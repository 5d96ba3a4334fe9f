// Dense Cholesky factorisation and solve for the n x n dual Hessian of the second-order iteration
// (n = tokens, 10^2 .. a few 10^3), fp64, gfx950 only.  Written here rather than taken from rocSOLVER:
// at this size the vendor path costs ~3.5 ms per factorisation (and minutes of one-time library
// initialisation on a fresh box) where the problem is bound by the length of the elimination chain
// (n dependent steps) and by launch latency.
//
// Layout: column-major, lower triangle.  nr = n rounded up to the block size NB = 32 (padding rows / columns
// carry an identity diagonal); the right-hand side rides along as ONE EXTRA ROW at index nr, so the
// factorisation forward-substitutes it for free (row nr of L is y = L^-1 b); ld = nr + NB.
// Right-looking, two launches per block column:
//   chol_panel_kernel   every workgroup (256 threads) factors the NB x NB diagonal block itself, operands in
//                       registers, one column exchanged through LDS per elimination step (one barrier per
//                       step), then solves its own 64 rows of the panel against it column by column.
//                       Workgroup 0 writes the factor back and leaves the inverse of the diagonal block in
//                       `Dinv` for the back substitution.  (A one-wave variant with row-per-lane registers and
//                       v_readlane broadcasts needed no barrier but issued ~3300 instructions from a single
//                       wave: 24 us per launch against 16.6 us.  NB = 64 halves the launches but the per-step
//                       rank-one work quadruples: 50 us per panel launch, 1.06 ms per factorisation against
//                       0.85 ms at NB = 32.)
//   chol_update_kernel  trailing update C_ij -= P_i P_j' on the lower 64 x 64 tiles (C prefetched into
//                       registers before the panel is staged in LDS, 32 columns at a time).
// chol_back_kernel: L' x = y, one workgroup, dot form (every pass over L reads contiguous columns: wave w
// owns NB/16 columns of the block), the diagonal block applied as an NB x NB mat-vec with its inverse.
#pragma once
#include "kernels.hpp"

namespace cfmm {

constexpr int CH_NB = 32;

// shared scratch of the 32 x 32 block routines below
struct BlockLds {
    double Lc[CH_NB][CH_NB + 1];         // Lc[j][r]: FINAL column j of the block, rows r >= j (unscaled while factoring)
    double E0[2][CH_NB], E1[2][CH_NB];   // exchange: column j (final) and column j + 1 (still missing column j's update)
    double X0[2][64], X1[2][64];         // the same for the panel solve
    double piv[CH_NB];                   // 1 / L[j][j]
};

// In-place Cholesky of a 32 x 32 block by 256 threads.  Thread (r = tid % 32, g = tid / 32) holds the block's elements
// (r, c = g + 8 q), q < 4, in a[] (entries above the diagonal must be 0).  TWO columns per barrier: with column j final
// and column j + 1 published as it stands, every thread finishes column j + 1 for its own rows itself
// (f = e1 - e0 l, l = D[j+1][j] / p_j) and applies the rank-two update; the dependent chain per pair is one LDS round
// trip, two reciprocals and one barrier.  On return S.Lc[c][r] = L[r][c] (r >= c), S.piv[j] = 1 / L[j][j].
__device__ __forceinline__ bool block_factor(double (&a)[CH_NB / 8], BlockLds &S)
{
    constexpr int NB = CH_NB, G = 256 / NB, Q = NB / G;
    static_assert(NB % 2 == 0 && Q == CH_NB / 8, "columns are eliminated in pairs; 256 threads");
    const int tid = threadIdx.x, r = tid % NB, g = tid / NB;
    if (g == 0) { S.E0[0][r] = a[0]; S.Lc[0][r] = a[0]; }
    if (g == 1 % G) S.E1[0][r] = a[1 / G];
    __syncthreads();
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NB; j += 2) {
        const int pb = (j >> 1) & 1;
        const double p0 = S.E0[pb][j], d10 = S.E0[pb][j + 1];
        const bool pos0 = p0 > 0.0 && p0 < 1.7976931348623157e308;
        const double ip0 = rcp_nr(pos0 ? p0 : 1.0);
        const double m = d10 * ip0;
        const double p1 = fma(-d10, m, S.E1[pb][j + 1]);
        const bool pos1 = p1 > 0.0 && p1 < 1.7976931348623157e308;
        ok = ok && pos0 && pos1;
        const double ip1 = rcp_nr(pos1 ? p1 : 1.0);
        const double e0r = S.E0[pb][r];
        const double fr = fma(-e0r, m, S.E1[pb][r]);           // final column j + 1, own row
        const double s0 = e0r * ip0, s1 = fr * ip1;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int c = g + G * q;
            if (c > j + 1 && c <= r) {
                const double e0c = S.E0[pb][c];
                const double fc = fma(-e0c, m, S.E1[pb][c]);
                a[q] = fma(-s0, e0c, fma(-s1, fc, a[q]));
            }
        }
        if (g == ((j + 1) % G)) { a[(j + 1) / G] = (r >= j + 1) ? fr : 0.0; S.Lc[j + 1][r] = a[(j + 1) / G]; }
        if (j + 2 < NB) {
            if (g == ((j + 2) % G)) { S.E0[pb ^ 1][r] = a[(j + 2) / G]; S.Lc[j + 2][r] = a[(j + 2) / G]; }
            if (g == ((j + 3) % G)) S.E1[pb ^ 1][r] = a[(j + 3) / G];
        }
        __syncthreads();
    }
    if (tid < NB) { const double p = S.Lc[tid][tid]; S.piv[tid] = rsqrt_nr(p > 0.0 && p < 1.7976931348623157e308 ? p : 1.0); }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q) { const int c = g + G * q; if (c < r) S.Lc[c][r] *= S.piv[c]; }      // L[r][c]
    if (g == 0) S.Lc[r][r] *= S.piv[r];                                                              // sqrt(p)
    __syncthreads();
    return ok;
}

// X L' = B for 64 rows against the factor left in S by block_factor.  Thread (rr = tid % 64, h = tid / 64) holds
// x[rr][c = h + 4 q], q < 8.  Two columns per barrier:  x_j = raw_j / L_jj,  x_{j+1} = (raw_{j+1} - x_j L[j+1][j]) / L_{j+1,j+1}.
__device__ __forceinline__ void block_solve(double (&x)[CH_NB / 4], BlockLds &S)
{
    constexpr int NB = CH_NB, H = 4, QX = NB / H;
    const int tid = threadIdx.x, rr = tid & 63, h = tid >> 6;
    if (h == 0) S.X0[0][rr] = x[0];
    if (h == 1) S.X1[0][rr] = x[0];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NB; j += 2) {
        const int pb = (j >> 1) & 1;
        const double xj = S.X0[pb][rr] * S.piv[j];
        const double xj1 = fma(-xj, S.Lc[j][j + 1], S.X1[pb][rr]) * S.piv[j + 1];
#pragma unroll
        for (int q = 0; q < QX; ++q) { const int c = h + H * q; if (c > j + 1) x[q] = fma(-xj, S.Lc[j][c], fma(-xj1, S.Lc[j + 1][c], x[q])); }
        if (h == (j % H)) x[j / H] = xj;
        if (h == ((j + 1) % H)) x[(j + 1) / H] = xj1;
        if (j + 2 < NB) {
            if (h == ((j + 2) % H)) S.X0[pb ^ 1][rr] = x[(j + 2) / H];
            if (h == ((j + 3) % H)) S.X1[pb ^ 1][rr] = x[(j + 3) / H];
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
chol_panel_kernel(double *__restrict__ A, int ld, int nrows, int k0, double *__restrict__ Dinv, int *__restrict__ info)
{
    constexpr int NB = CH_NB, G = 256 / NB, Q = NB / G, H = 4, QX = NB / H;
    __shared__ BlockLds S;
    const int tid = threadIdx.x;
    {
        const int r = tid % NB, g = tid / NB;
        double a[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) { const int c = g + G * q; const double v = A[(size_t)(k0 + c) * ld + k0 + r]; a[q] = (c <= r) ? v : 0.0; }
        const bool ok = block_factor(a, S);
        if (blockIdx.x == 0) {
            if (!ok && tid == 0) atomicMax(info, k0 + 1);
#pragma unroll
            for (int q = 0; q < Q; ++q) { const int c = g + G * q; A[(size_t)(k0 + c) * ld + k0 + r] = (c <= r) ? S.Lc[c][r] : 0.0; }
        }
    }
    // ---- solve X L' = B for 64 rows (workgroup 0: B = I, giving X = L^-T, i.e. the inverse transposed)
    const int rr = tid & 63, h = tid >> 6;
    const int row = k0 + NB + 64 * ((int)blockIdx.x - 1) + rr;
    const bool diag_wg = blockIdx.x == 0;
    const bool live = diag_wg ? rr < NB : row < nrows;
    double x[QX];
#pragma unroll
    for (int q = 0; q < QX; ++q) {
        const int c = h + H * q;
        if (diag_wg) x[q] = (c == rr) ? 1.0 : 0.0;
        else x[q] = A[(size_t)(k0 + c) * ld + (live ? row : k0)];
    }
    block_solve(x, S);
    if (live) {
#pragma unroll
        for (int q = 0; q < QX; ++q) {
            const int c = h + H * q;
            if (diag_wg) Dinv[(size_t)(k0 / NB) * NB * NB + c * NB + rr] = x[q];      // Linv[c][rr] = X[rr][c]
            else A[(size_t)(k0 + c) * ld + row] = x[q];
        }
    }
}

// trailing update after block column k0: tile (ti, tj), ti >= tj, of 64 x 64 over rows >= k0 + NB (< nrows) and
// columns >= k0 + NB (< ncols)
__global__ void __launch_bounds__(256)
chol_update_kernel(double *__restrict__ A, int ld, int nrows, int ncols, int k0)
{
    __shared__ double Pi[32][64 + 1], Pj[32][64 + 1];
    // linear tile index -> (ti, tj) in the lower triangle
    int t = blockIdx.x, ti = 0;
    while (t > ti) { t -= ti + 1; ++ti; }
    const int tj = t;
    const int base = k0 + CH_NB;
    const int i0 = base + 64 * ti, j0 = base + 64 * tj;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // rows tx + 16 u, columns ty + 16 v
    double cv[4][4];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = i0 + tx + 16 * u, col = j0 + ty + 16 * v;
            cv[u][v] = (row < nrows && col < ncols && row >= col) ? A[(size_t)col * ld + row] : 0.0;
        }
    for (int kk = 0; kk < CH_NB; kk += 32) {
        if (kk) __syncthreads();
        for (int e = threadIdx.x; e < 32 * 64; e += 256) {
            const int c = e >> 6, rr = e & 63;
            Pi[c][rr] = (i0 + rr < nrows) ? A[(size_t)(k0 + kk + c) * ld + i0 + rr] : 0.0;
            Pj[c][rr] = (j0 + rr < ncols) ? A[(size_t)(k0 + kk + c) * ld + j0 + rr] : 0.0;
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
}

// L' x = y with y = row nr of the factored array; x -> out[0..n).  One workgroup of 1024.  LDS: x[nr] | d[NB]
constexpr int CH_SOLVE_THREADS = 1024;
__global__ void __launch_bounds__(CH_SOLVE_THREADS)
chol_back_kernel(const double *__restrict__ L, int ld, int nr, int n, const double *__restrict__ Dinv, double *__restrict__ out)
{
    constexpr int NB = CH_NB, CW = NB / 16;       // columns per wave
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *x = lds, *dsum = lds + nr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < nr; i += blockDim.x) x[i] = L[(size_t)i * ld + nr];
    __syncthreads();
    for (int k0 = nr - NB; k0 >= 0; k0 -= NB) {
        // wave w: columns k0 + 2 w, k0 + 2 w + 1 of L dotted with the part of x already solved
        {
            static_assert(CW == 2, "two columns per wave");
            const int c0 = k0 + 2 * wave;
            double p0 = 0.0, p1 = 0.0;
            const double *L0 = L + (size_t)c0 * ld, *L1 = L0 + ld;
            int j = k0 + NB + lane;
            for (; j + 192 < nr; j += 256) {              // four independent row strips in flight per pass
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
            const double *D = Dinv + (size_t)(k0 / NB) * NB * NB;
            double acc = 0.0;
#pragma unroll 8
            for (int c = 0; c < NB; ++c) acc = fma(D[c * NB + lane], x[k0 + c] - dsum[c], acc);
            x[k0 + lane] = acc;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += blockDim.x) out[i] = x[i];
}

}  // namespace cfmm
