// Dense Cholesky factorisation and solve for the n x n dual Hessian of the second-order iteration
// (n = tokens, 10^2 .. a few 10^3), fp64, gfx950 only.  Written here rather than taken from rocSOLVER:
// at this size the vendor path costs ~3.5 ms per factorisation (and minutes of one-time library
// initialisation on a fresh box) where the problem is launch-latency bound at a few hundred microseconds.
//
// Layout: column-major, lower triangle.  nr = n rounded up to the block size 32 (padding rows / columns
// carry an identity diagonal); the right-hand side rides along as ONE EXTRA ROW at index nr, so the
// factorisation forward-substitutes it for free (row nr of L is y = L^-1 b); ld = nr + 32.
// Right-looking, two launches per block column:
//   chol_panel_kernel   every workgroup (256 threads) factors the 32 x 32 diagonal block itself, operands in
//                       registers, one column exchanged through LDS per elimination step, then solves its
//                       own 64 rows of the panel against it column by column.  Workgroup 0 writes the factor
//                       back and leaves the inverse of the diagonal block in `Dinv` for the back substitution.
//                       (A one-wave variant with row-per-lane registers and v_readlane broadcasts needed no
//                       barrier but issued ~3300 instructions from a single wave: 24 us per launch.)
//   chol_update_kernel  trailing update C_ij -= P_i P_j' on the lower 64 x 64 tiles (C prefetched into
//                       registers before the panel is staged in LDS).
// chol_back_kernel: L' x = y, one workgroup, dot form (every pass over L reads contiguous columns: wave w
// owns two columns of the block), the diagonal block applied as a 32 x 32 mat-vec with its inverse.
#pragma once
#include "kernels.hpp"

namespace cfmm {

constexpr int CH_NB = 32;

__device__ __forceinline__ double lane_bcast(double v, int l)        // l uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// one block column: diagonal factor + panel solve, 256 threads, operands in registers, one LDS column
// exchanged per elimination step (one barrier per step).  Workgroup 0 is the diagonal block's: it writes the
// factor back and leaves the block's inverse in `Dinv` (row-major [i][c]) for the back substitution;
// workgroup w >= 1 owns panel rows k0 + 32 + 64 (w - 1) .. + 63 and redoes the (cheap) factorisation itself
// instead of waiting for another launch.
__global__ void __launch_bounds__(256)
chol_panel_kernel(double *__restrict__ A, int ld, int nrows, int k0, double *__restrict__ Dinv, int *__restrict__ info)
{
    __shared__ double Lc[CH_NB][CH_NB + 1];      // Lc[j][r]: column j of the block, rows r >= j (unscaled while factoring)
    __shared__ double Xs[CH_NB][64 + 1];         // Xs[j][row]: solved column j of this workgroup's panel rows
    __shared__ double piv[CH_NB];                // 1 / L[j][j]
    const int tid = threadIdx.x;
    // ---- factor.  thread (r, g) keeps the block's elements (r, c = g + 8 q), q = 0..3
    {
        const int r = tid & 31, g = tid >> 5;
        double a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int c = g + 8 * q; const double v = A[(size_t)(k0 + c) * ld + k0 + r]; a[q] = (c <= r) ? v : 0.0; }
        if (g == 0) Lc[0][r] = a[0];
        __syncthreads();
        bool ok = true;
#pragma unroll
        for (int j = 0; j < CH_NB; ++j) {
            const double p = Lc[j][j];
            const bool pos = p > 0.0 && p < 1.7976931348623157e308;
            ok = ok && pos;
            const double lrj = Lc[j][r] * rcp_nr(pos ? p : 1.0);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int c = g + 8 * q; if (c > j && c <= r) a[q] = fma(-lrj, Lc[j][c], a[q]); }
            if (j + 1 < CH_NB && g == ((j + 1) & 7)) Lc[j + 1][r] = a[(j + 1) >> 3];     // column j + 1 is final: publish it
            __syncthreads();
        }
        if (tid < CH_NB) { const double p = Lc[tid][tid]; piv[tid] = rsqrt_nr(p > 0.0 && p < 1.7976931348623157e308 ? p : 1.0); }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int c = g + 8 * q; if (c < r) Lc[c][r] *= piv[c]; }      // L[r][c]
        if (g == 0) Lc[r][r] *= piv[r];                                                              // sqrt(p)
        __syncthreads();
        if (blockIdx.x == 0) {
            if (!ok && tid == 0) atomicMax(info, k0 + 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int c = g + 8 * q; A[(size_t)(k0 + c) * ld + k0 + r] = (c <= r) ? Lc[c][r] : 0.0; }
        }
    }
    // ---- solve X L' = B for 64 rows (workgroup 0: B = I on 32 rows, giving X = L^-T, i.e. the inverse transposed).
    //      thread (rr, h) keeps x[rr][c = h + 4 q], q = 0..7
    const int rr = tid & 63, h = tid >> 6;
    const int row = k0 + CH_NB + 64 * ((int)blockIdx.x - 1) + rr;
    const bool diag_wg = blockIdx.x == 0;
    const bool live = diag_wg ? rr < CH_NB : row < nrows;
    double x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = h + 4 * q;
        if (diag_wg) x[q] = (c == rr) ? 1.0 : 0.0;
        else x[q] = A[(size_t)(k0 + c) * ld + (live ? row : k0)];
    }
#pragma unroll
    for (int j = 0; j < CH_NB; ++j) {
        if (h == (j & 3)) { const double xj = x[j >> 2] * piv[j]; x[j >> 2] = xj; Xs[j][rr] = xj; }
        __syncthreads();
        const double xj = Xs[j][rr];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int c = h + 4 * q; if (c > j) x[q] = fma(-xj, Lc[j][c], x[q]); }
    }
    if (live) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = h + 4 * q;
            if (diag_wg) Dinv[(size_t)(k0 / CH_NB) * CH_NB * CH_NB + c * CH_NB + rr] = x[q];      // Linv[c][rr] = X[rr][c]
            else A[(size_t)(k0 + c) * ld + row] = x[q];
        }
    }
}

// trailing update after block column k0: tile (ti, tj), ti >= tj, of 64 x 64 over rows >= k0 + 32 (< nrows) and
// columns >= k0 + 32 (< ncols)
__global__ void __launch_bounds__(256)
chol_update_kernel(double *__restrict__ A, int ld, int nrows, int ncols, int k0)
{
    __shared__ double Pi[CH_NB][64 + 1], Pj[CH_NB][64 + 1];
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
    for (int e = threadIdx.x; e < CH_NB * 64; e += 256) {
        const int c = e >> 6, rr = e & 63;
        Pi[c][rr] = (i0 + rr < nrows) ? A[(size_t)(k0 + c) * ld + i0 + rr] : 0.0;
        Pj[c][rr] = (j0 + rr < ncols) ? A[(size_t)(k0 + c) * ld + j0 + rr] : 0.0;
    }
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < CH_NB; ++c) {
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
}

// L' x = y with y = row nr of the factored array; x -> out[0..n).  One workgroup of 1024.  LDS: x[nr] | d[32]
constexpr int CH_SOLVE_THREADS = 1024;
__global__ void __launch_bounds__(CH_SOLVE_THREADS)
chol_back_kernel(const double *__restrict__ L, int ld, int nr, int n, const double *__restrict__ Dinv, double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *x = lds, *dsum = lds + nr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < nr; i += blockDim.x) x[i] = L[(size_t)i * ld + nr];
    __syncthreads();
    for (int k0 = nr - CH_NB; k0 >= 0; k0 -= CH_NB) {
        // wave w: columns k0 + 2 w, k0 + 2 w + 1 of L dotted with the part of x already solved
        {
            const int c0 = k0 + 2 * wave;
            double p0 = 0.0, p1 = 0.0;
            const double *L0 = L + (size_t)c0 * ld, *L1 = L0 + ld;
            int j = k0 + CH_NB + lane;
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
        if (wave == 0 && lane < CH_NB) {
            // x_r = sum_{c >= r} Linv[c][r] (y_c - d_c)
            const double *D = Dinv + (size_t)(k0 / CH_NB) * CH_NB * CH_NB;
            double acc = 0.0;
#pragma unroll 8
            for (int c = 0; c < CH_NB; ++c) acc = fma(D[c * CH_NB + lane], x[k0 + c] - dsum[c], acc);
            x[k0 + lane] = acc;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += blockDim.x) out[i] = x[i];
}

}  // namespace cfmm
