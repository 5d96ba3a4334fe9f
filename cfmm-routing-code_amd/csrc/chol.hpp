// Dense Cholesky factorisation and solve for the n x n dual Hessian of the second-order iteration
// (n = tokens, 10^2 .. a few 10^3), fp64, gfx950 only.  Written here rather than taken from rocSOLVER:
// at this size the vendor path costs ~3.5 ms per factorisation (and minutes of one-time library
// initialisation on a fresh box) where the problem is launch-latency bound at a few hundred microseconds.
//
// Layout: column-major, lower triangle, leading dimension ld = n rounded up to the block size (the
// padding rows / columns carry an identity diagonal).  Right-looking, block size 32, two launches per
// block column:
//   chol_panel_kernel   every wave factors the 32 x 32 diagonal block itself, row-per-lane in registers
//                       (cross-lane traffic through v_readlane: no LDS, no barrier), then solves its own
//                       64 rows of the panel against it; wave 0 of workgroup 0 writes the factor back.
//   chol_update_kernel  trailing update C_ij -= P_i P_j' on the lower 64 x 64 tiles, panel staged in LDS.
// chol_solve_kernel: one workgroup, forward substitution in axpy form and backward substitution in dot
// form, so that every pass over L reads contiguous columns.
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

// lanes r and r + 32 both hold row r of the diagonal block (a[c] = A[r][c], c <= r): in-register Cholesky.
// Returns false on a non-positive pivot (the block is then garbage).
__device__ __forceinline__ bool diag_factor(double (&a)[CH_NB], int r)
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < CH_NB; ++j) {
        const double ajj = lane_bcast(a[j], j);
        if (!(ajj > 0.0)) ok = false;
        const double inv = 1.0 / sqrt(ajj > 0.0 ? ajj : 1.0);
        a[j] = a[j] * inv;                                   // column j final (lane j: sqrt(ajj))
#pragma unroll
        for (int c = j + 1; c < CH_NB; ++c) a[c] = fma(-a[j], lane_bcast(a[j], c), a[c]);
        SCHED_FENCE();                                       // (keeps the scalar broadcasts of later columns from being hoisted and spilled)
    }
    return ok;
}

// one block column: diagonal factor + panel solve.  64 threads per workgroup; workgroup w owns panel rows
// k0 + 32 + 64 w + lane.
__global__ void __launch_bounds__(64)
chol_panel_kernel(double *__restrict__ A, int ld, int n, int k0, int *__restrict__ info)
{
    const int lane = threadIdx.x, r = lane & 31;
    double l[CH_NB];
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) { const double v = A[(size_t)(k0 + c) * ld + k0 + r]; l[c] = (c <= r) ? v : 0.0; }   // (unconditional loads: no exec-mask juggling)
    const bool ok = diag_factor(l, r);
    if (blockIdx.x == 0) {
        if (!ok && lane == 0) atomicMax(info, k0 + 1);
        if (lane < 32) {
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) A[(size_t)(k0 + c) * ld + k0 + r] = (c <= r) ? l[c] : 0.0;
        }
    }
    // (opaque copy: otherwise the 496 scalar broadcasts of the factorisation are kept alive for the solve below -- ~1000 spilled SGPRs)
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) asm volatile("" : "+v"(l[c]));
    const int row = k0 + CH_NB + 64 * blockIdx.x + lane;
    if (k0 + CH_NB + 64 * (int)blockIdx.x >= n) return;        // (uniform) nothing below the diagonal block for this workgroup
    const bool live = row < n;
    const size_t rr = live ? row : (size_t)k0;
    double x[CH_NB];
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) x[c] = A[(size_t)(k0 + c) * ld + rr];
    // x L' = a  ->  x[c] = (a[c] - sum_{j<c} x[j] L[c][j]) / L[c][c];  L[c][j] sits in lane c, register j
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) {
        double acc = x[c];
#pragma unroll
        for (int j = 0; j < c; ++j) acc = fma(-x[j], lane_bcast(l[j], c), acc);
        x[c] = acc / lane_bcast(l[c], c);
        SCHED_FENCE();
    }
    if (live) {
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) A[(size_t)(k0 + c) * ld + row] = x[c];
    }
}

// trailing update after block column k0: tile (ti, tj), ti >= tj, of 64 x 64 over rows / columns >= k0 + 32
__global__ void __launch_bounds__(256)
chol_update_kernel(double *__restrict__ A, int ld, int n, int k0)
{
    __shared__ double Pi[CH_NB][64 + 1], Pj[CH_NB][64 + 1];
    // linear tile index -> (ti, tj) in the lower triangle
    int t = blockIdx.x, ti = 0;
    while (t > ti) { t -= ti + 1; ++ti; }
    const int tj = t;
    const int base = k0 + CH_NB;
    const int i0 = base + 64 * ti, j0 = base + 64 * tj;
    for (int e = threadIdx.x; e < CH_NB * 64; e += 256) {
        const int c = e >> 6, rr = e & 63;
        Pi[c][rr] = (i0 + rr < n) ? A[(size_t)(k0 + c) * ld + i0 + rr] : 0.0;
        Pj[c][rr] = (j0 + rr < n) ? A[(size_t)(k0 + c) * ld + j0 + rr] : 0.0;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // rows tx + 16 u, columns ty + 16 v
    double acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
#pragma unroll 8
    for (int c = 0; c < CH_NB; ++c) {
        double pi[4], pj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { pi[u] = Pi[c][tx + 16 * u]; pj[u] = Pj[c][ty + 16 * u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] = fma(pi[u], pj[v], acc[u][v]);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int col = j0 + ty + 16 * v;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = i0 + tx + 16 * u;
            if (row < n && col < n && row >= col) A[(size_t)col * ld + row] -= acc[u][v];
        }
    }
}

// L L' x = b in place (b has room for ld entries).  One workgroup.  LDS: y[ld] | d[32]
constexpr int CH_SOLVE_THREADS = 1024;
__global__ void __launch_bounds__(CH_SOLVE_THREADS)
chol_solve_kernel(const double *__restrict__ L, int ld, int n, double *__restrict__ b)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *y = lds, *dsum = lds + ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31;
    const int nblk = ld / CH_NB;
    for (int i = tid; i < ld; i += blockDim.x) y[i] = i < n ? b[i] : 0.0;
    __syncthreads();
    // ---- forward: L y = b, block column by block column (axpy form)
    for (int kb = 0; kb < nblk; ++kb) {
        const int k0 = kb * CH_NB;
        if (wave == 0) {
            double l[CH_NB];
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) { const double v = L[(size_t)(k0 + c) * ld + k0 + r]; l[c] = (c <= r) ? v : 0.0; }
            double t = y[k0 + r];
            double mine = 0.0;
#pragma unroll
            for (int j = 0; j < CH_NB; ++j) {
                const double yj = lane_bcast(t, j) / lane_bcast(l[j], j);
                if (r == j) mine = yj;
                t = fma(-l[j], yj, t);               // lanes r > j (l[j] = 0 above the diagonal)
            }
            if (lane < 32) y[k0 + r] = mine;
        }
        __syncthreads();
        for (int i = k0 + CH_NB + tid; i < ld; i += blockDim.x) {
            double acc = 0.0;
#pragma unroll 8
            for (int c = 0; c < CH_NB; ++c) acc = fma(L[(size_t)(k0 + c) * ld + i], y[k0 + c], acc);
            y[i] -= acc;
        }
        __syncthreads();
    }
    // ---- backward: L' x = y, block column by block column from the last (dot form)
    for (int kb = nblk - 1; kb >= 0; --kb) {
        const int k0 = kb * CH_NB;
        if (tid < CH_NB) dsum[tid] = 0.0;
        __syncthreads();
        if (k0 + CH_NB < ld) {
            double p[CH_NB];
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) p[c] = 0.0;
            for (int j = k0 + CH_NB + tid; j < ld; j += blockDim.x) {
                const double xj = y[j];
#pragma unroll
                for (int c = 0; c < CH_NB; ++c) p[c] = fma(L[(size_t)(k0 + c) * ld + j], xj, p[c]);
            }
            if (k0 + CH_NB + wave * 64 < ld) {                 // (uniform per wave) waves with no row skip the reduction
#pragma unroll
                for (int c = 0; c < CH_NB; ++c) {
                    const double s = wave_allsum(p[c]);
                    if (lane == 0) unsafeAtomicAdd(&dsum[c], s);
                }
            }
        }
        __syncthreads();
        if (wave == 0) {
            // lane c holds column c of the diagonal block: a[j] = L[j][c], j >= c
            double a[CH_NB];
#pragma unroll
            for (int j = 0; j < CH_NB; ++j) { const double v = L[(size_t)(k0 + r) * ld + k0 + j]; a[j] = (j >= r) ? v : 0.0; }
            double t = y[k0 + r] - dsum[r];
            double mine = 0.0;
#pragma unroll
            for (int j = CH_NB - 1; j >= 0; --j) {
                const double xj = lane_bcast(t, j) / lane_bcast(a[j], j);
                if (r == j) mine = xj;
                t = fma(-a[j], xj, t);               // lanes c < j
            }
            if (lane < 32) y[k0 + r] = mine;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += blockDim.x) b[i] = y[i];
}

}  // namespace cfmm
