// Token-block ordering of a bucket's pools, on the device, once per upload.
//
// The evaluation gives every workgroup a contiguous range of each bucket's wave-tiles (kernels.hpp:
// eval_tiles_and_flush).  If the pools arrive in arbitrary order, the ~4000 pools of one workgroup touch EVERY token: its
// LDS psi tile is dense, the flush sends n atomics per workgroup (256 k - 512 k per launch), the nu gathers and psi scatters
// roam over the whole tile.  Ordered by the BLOCK PAIR of their first two tokens -- tokens cut into 16 blocks, key =
// (lower block, higher block): 136 keys -- the pools of one workgroup's range touch a few hundred tokens instead.
// C4 shard (1.25e6 constant-product pools / 2000 tokens): evaluation 16.1 -> 10.5 us, outer iteration 27.5 -> 23.1 us.
//
// A counting sort with 256 keys: histogram, exclusive scan, scatter (workgroup-local ranks through LDS, one global
// reservation per workgroup and key).  The order INSIDE a key is whatever the reservations make it -- it carries no
// meaning (psi is a sum) -- and perm[position] = original index lets the tenders go back out in the caller's order.
// The sort runs lazily, in front of the first kernel that reads the pools (cfmm_hip.hip: pools_ready), from the arena
// the upload landed the columns in into a NEW arena (+ the permutation, + the 256 counters); the landing arena is freed
// by release_landed of the same context, behind that entry point's own synchronisation.  A context and its clones share
// the pool store: its landing list and pending flags are guarded by PoolStore::mu.
#pragma once
#include "kernels.hpp"

namespace cfmm {

constexpr int RO_NB = 16, RO_KEYS = RO_NB * RO_NB, RO_THREADS = 1024, RO_PER = 1;
// below these a bucket stays as it is (perm = null); K-asset pools gain little (only two of their K legs are localised)
constexpr long long RO_MIN_POOLS = 16384, RO_MIN_POOLS_N = 65536;

__device__ __forceinline__ int ro_key(int ta, int tb, int bsz)
{
    const int a = ta / bsz, b = tb / bsz;
    return (a < b ? a : b) * RO_NB + (a < b ? b : a);
}

// K = 2: ia / ib columns; K >= 3: pool-major legs, the first two legs of pool i at idx[i K], idx[i K + 1]
template <int K>
__global__ void __launch_bounds__(RO_THREADS)
ro_hist_kernel(const int *__restrict__ ia, const int *__restrict__ ib, long long m, int bsz, unsigned *__restrict__ hist)
{
    __shared__ unsigned h[RO_KEYS];
    if (threadIdx.x < RO_KEYS) h[threadIdx.x] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
        const int ta = K == 2 ? ia[i] : ia[i * K], tb = K == 2 ? ib[i] : ia[i * K + 1];
        atomicAdd(&h[ro_key(ta, tb, bsz)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < RO_KEYS && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

__global__ void __launch_bounds__(64)
ro_scan_kernel(unsigned *hist)                    // counts -> exclusive offsets (the scatter's cursors)
{
    if (threadIdx.x == 0) {
        unsigned c = 0;
        for (int k = 0; k < RO_KEYS; ++k) { const unsigned v = hist[k]; hist[k] = c; c += v; }
    }
}

struct Cols2 { double *Ra, *Rb, *fee, *param; int *ia, *ib; };
struct ColsN { int *idx; double *R, *w, *fee, *lfee, *lrw; };

__global__ void __launch_bounds__(RO_THREADS)
ro_scatter2_kernel(Bucket2 s, Cols2 d, int *__restrict__ perm, int bsz, unsigned *__restrict__ cursor)
{
    __shared__ unsigned cnt[RO_KEYS], base[RO_KEYS];
    if (threadIdx.x < RO_KEYS) cnt[threadIdx.x] = 0;
    __syncthreads();
    const long long i0 = (long long)blockIdx.x * (RO_THREADS * RO_PER);
    int key[RO_PER]; unsigned rank[RO_PER];
#pragma unroll
    for (int r = 0; r < RO_PER; ++r) {
        const long long i = i0 + r * RO_THREADS + threadIdx.x;
        key[r] = -1; rank[r] = 0;
        if (i < s.m) { key[r] = ro_key(s.ia[i], s.ib[i], bsz); rank[r] = atomicAdd(&cnt[key[r]], 1u); }
    }
    __syncthreads();
    if (threadIdx.x < RO_KEYS) { const unsigned c = cnt[threadIdx.x]; base[threadIdx.x] = c ? atomicAdd(&cursor[threadIdx.x], c) : 0u; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RO_PER; ++r) {
        if (key[r] < 0) continue;
        const long long i = i0 + r * RO_THREADS + threadIdx.x;
        const long long p = (long long)base[key[r]] + rank[r];
        d.Ra[p] = s.Ra[i]; d.Rb[p] = s.Rb[i]; d.fee[p] = s.fee[i];
        if (s.param) d.param[p] = s.param[i];
        d.ia[p] = s.ia[i]; d.ib[p] = s.ib[i];
        perm[p] = (int)i;
    }
}

template <int K>
__global__ void __launch_bounds__(RO_THREADS)
ro_scatterN_kernel(BucketN s, ColsN d, int *__restrict__ perm, int bsz, unsigned *__restrict__ cursor)
{
    __shared__ unsigned cnt[RO_KEYS], base[RO_KEYS];
    if (threadIdx.x < RO_KEYS) cnt[threadIdx.x] = 0;
    __syncthreads();
    const long long i0 = (long long)blockIdx.x * (RO_THREADS * RO_PER);
    int key[RO_PER]; unsigned rank[RO_PER];
#pragma unroll
    for (int r = 0; r < RO_PER; ++r) {
        const long long i = i0 + r * RO_THREADS + threadIdx.x;
        key[r] = -1; rank[r] = 0;
        if (i < s.m) { key[r] = ro_key(s.idx[i * K], s.idx[i * K + 1], bsz); rank[r] = atomicAdd(&cnt[key[r]], 1u); }
    }
    __syncthreads();
    if (threadIdx.x < RO_KEYS) { const unsigned c = cnt[threadIdx.x]; base[threadIdx.x] = c ? atomicAdd(&cursor[threadIdx.x], c) : 0u; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RO_PER; ++r) {
        if (key[r] < 0) continue;
        const long long i = i0 + r * RO_THREADS + threadIdx.x;
        const long long p = (long long)base[key[r]] + rank[r];
#pragma unroll
        for (int j = 0; j < K; ++j) { d.idx[p * K + j] = s.idx[i * K + j]; d.R[p * K + j] = s.R[i * K + j]; d.w[p * K + j] = s.w[i * K + j]; d.lrw[p * K + j] = s.lrw[i * K + j]; }
        d.fee[p] = s.fee[i]; d.lfee[p] = s.lfee[i];
        perm[p] = (int)i;
    }
}

}  // namespace cfmm
