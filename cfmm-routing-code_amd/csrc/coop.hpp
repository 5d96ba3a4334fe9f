// Mid-size networks (BASELINE config 2: 1e4 pools / 100 tokens): the whole solve in ONE launch of a handful of
// cooperating workgroups.
//
// A network of this size is launch-latency bound in the grid-wide path: 0.3 MB of pool data per evaluation, ~10 us per
// outer iteration of which the evaluation proper is one.  tiny.hpp removes the launches for what ONE workgroup can
// evaluate (<= 64 wave-tiles); here G workgroups (2..32, one wave-tile or two per wave) share the tiles and everything
// else stays as in tiny.hpp: the psi tile never leaves LDS through an accumulator, the projected L-BFGS step is taken by
// ONE WAVE with its state in registers (onewave.hpp, two tokens per lane: <= 128 tokens) -- by wave 0 of EVERY workgroup,
// redundantly and bit-identically, so that no price vector has to travel.  What does travel, once per iteration, is each
// workgroup's partial psi (n + 1 doubles, the diagonal metric too on the first evaluation): an all-gather through a slab
// in global memory,
//     payload: 8-byte agent-scope relaxed atomic stores (write-through) -> every thread drains its own (s_waitcnt vmcnt(0))
//              -> barrier -> ONE agent-scope flag store carrying the iteration's epoch;
//     readers: lanes of wave 0 poll the G flags (agent-scope relaxed loads, s_sleep between polls, bounded), then read the
//              G partials with agent-scope loads and add them in workgroup order (the same bits everywhere)
// -- MI355X_MICROARCH.md's "8-byte agent atomics on both sides" form: no fences, ~1-2 us per exchange.  Two slab
// parities: a workgroup can run at most one exchange ahead of the slowest.  The flags are monotone over the life of the
// context (epoch base per solve), so nothing is cleared between solves.
// The workgroups must be co-resident (they wait for each other): G <= 32 of 256 CUs on an otherwise idle device always
// are; a poll that runs out of its bound ends the solve with status 4 and the host repeats it through the grid-wide path.
//                                                                        reference: arbitrage.py:82 (prob.solve())
#pragma once
#include "onewave.hpp"

namespace cfmm {

constexpr int COOP_N = 128;                // tokens (and price groups): two per lane of wave 0
constexpr int COOP_THREADS = 512;          // 8 waves per workgroup walk the tiles
constexpr int COOP_MAX_WGS = 32;
constexpr int COOP_MAX_TILES = COOP_MAX_WGS * (COOP_THREADS / 64) * 4;       // up to four wave-tiles per wave
constexpr int COOP_STATUS_TIMEOUT = 4;

struct CoopArgs {
    unsigned long long *slab;              // [2][G][stride] 8-byte words: psi[n] | sum arb | pad | diag[n]
    unsigned long long *flags;             // [2][COOP_MAX_WGS]: epoch of the last partial workgroup w delivered into parity p
    int stride;
    unsigned long long epoch0;             // flag values of this solve are epoch0 + 1 + iteration
};
__host__ __device__ inline int coop_stride(int n) { return 2 * ((n + 1) & ~1) + 2; }

// LDS (doubles): eval_kernel<WITH_D>'s carve | exchange strips | q[128] | q2[128] | S[MAX_MEMORY][128] | Y[MAX_MEMORY][128] | ctl
__host__ __device__ inline int coop_lds_doubles(int n)
{
    return eval_lds_doubles(n, true) + 2 * 64 * (COOP_THREADS / 64) + 2 * COOP_N + 2 * MAX_MEMORY * COOP_N + 4;
}

__device__ __forceinline__ void agent_store(unsigned long long *p, double v)
{
    __hip_atomic_store(p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double agent_load(const unsigned long long *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__global__ void __launch_bounds__(COOP_THREADS)
solve_coop_kernel(EvalArgs ev, UpdArgs a, CoopArgs c, int iters)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int W = WaveUpdate<2>::W;
    const int n = ev.n, ng = a.ng, M = a.M;
    const int G = (int)gridDim.x, wg = (int)blockIdx.x;
    const int tid = threadIdx.x, L = tid & 63, wave = wuni((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
    const int tile = eval_tile_doubles(n, false);
    double *psi_s = lds, *diag_s = lds + tile;
    double *nu_s = lds + 2 * tile;                      // [n + 1]
    double *fpart = nu_s + n + 2;                       // [16]
    int *next_tile = reinterpret_cast<int *>(fpart + 16);
    double *strips = lds + eval_lds_doubles(n, true);
    double2 *xs = reinterpret_cast<double2 *>(strips) + 64 * wave;
    double *q = strips + 2 * 64 * (COOP_THREADS / 64);  // [128] group sums / the exchange of the trial point
    double *q2 = q + W;
    double *Sh = q2 + W, *Yh = Sh + MAX_MEMORY * W;     // history pairs, one row of 128 per slot
    int *status_s = reinterpret_cast<int *>(Yh + MAX_MEMORY * W);

    WaveUpdate<2> u;                                    // wave 0 of EVERY workgroup: the same state, the same steps
    if (wave == 0) u.load(a, L, n, ng);
    for (int j = tid; j < 2 * MAX_MEMORY * W; j += blockDim.x) Sh[j] = 0.0;
    for (int j = tid; j <= n; j += blockDim.x) nu_s[j] = a.nu[j];
    if (tid == 0) *status_s = 0;
    const int np = (n + 1) & ~1;

    for (int it = 0; it < iters; ++it) {
        const bool first = it == 0;
        for (int j = tid; j < 2 * tile; j += blockDim.x) lds[j] = 0.0;
        if (tid < 64) build_tile_table(ev, next_tile, tid);          // (this workgroup's share of the tiles; re-arms the ticket counter)
        __syncthreads();
        if (first) eval_tiles_and_flush<true, false, false, false, false>(ev, nullptr, nu_s, psi_s, diag_s, fpart, next_tile, xs);
        else eval_tiles_and_flush<false, false, false, false, false>(ev, nullptr, nu_s, psi_s, diag_s, fpart, next_tile, xs);
        // ---- publish this workgroup's partial (a barrier has been passed: tile and per-wave partials are complete) ---------
        const int par = it & 1;
        unsigned long long *mine = c.slab + ((size_t)par * G + wg) * c.stride;
        for (int j = tid; j < n; j += blockDim.x) {
            agent_store(mine + j, psi_s[j]);
            if (first) agent_store(mine + np + 2 + j, diag_s[j]);
        }
        if (tid == 0) {
            double f = 0.0;
            for (int w = 0; w < nw; ++w) f += fpart[w];
            agent_store(mine + np, f);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every thread: its own stores have left
        __syncthreads();
        const unsigned long long target = c.epoch0 + 1ull + (unsigned long long)it;
        if (tid == 0) __hip_atomic_store(c.flags + par * COOP_MAX_WGS + wg, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- wave 0: wait for everybody's partial of this iteration, add them in workgroup order, take the step ----------
        if (wave == 0) {
            bool ok = true;
            if (L < G) {
                const unsigned long long *f = c.flags + par * COOP_MAX_WGS + L;
                long spins = 0;
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1L << 22)) { ok = false; break; }      // (~ a second: a workgroup is not running)
                }
            }
            ok = __all(ok);
            asm volatile("" ::: "memory");               // (the partials are read after the flags have been seen)
            double psi[2] = {0.0, 0.0}, dg[2] = {0.0, 0.0}, fpools = 0.0;
            const unsigned long long *base = c.slab + (size_t)par * G * c.stride;
            const int j0 = L < n ? L : 0, j1 = L + 64 < n ? L + 64 : 0;          // (lanes without a token read entry 0 and drop it)
            for (int w0 = 0; w0 < G; w0 += 8) {          // eight partials' loads in flight at a time, added in workgroup order
                double t0[8], t1[8], tf[8], d0[8], d1[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int w = w0 + k < G ? w0 + k : G - 1;
                    const unsigned long long *p = base + (size_t)w * c.stride;
                    t0[k] = agent_load(p + j0); t1[k] = agent_load(p + j1); tf[k] = agent_load(p + np);
                    d0[k] = first ? agent_load(p + np + 2 + j0) : 0.0; d1[k] = first ? agent_load(p + np + 2 + j1) : 0.0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) if (w0 + k < G) { psi[0] += t0[k]; psi[1] += t1[k]; fpools += tf[k]; dg[0] += d0[k]; dg[1] += d1[k]; }
            }
            if (!(L < n)) { psi[0] = 0.0; dg[0] = 0.0; }
            if (!(L + 64 < n)) { psi[1] = 0.0; dg[1] = 0.0; }
            if (!ok) u.st.status = COOP_STATUS_TIMEOUT;
            else u.step(a, M, psi, dg, fpools, Sh, Yh, q, q2, nu_s);
            if (L == 0) *status_s = u.st.status;
        }
        __syncthreads();
        if (*status_s != 0) break;
    }
    if (wave == 0 && wg == 0) u.store(a);
}

}  // namespace cfmm
