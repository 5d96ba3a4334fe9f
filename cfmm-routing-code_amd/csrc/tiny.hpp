// The reference's own problem sizes: 5 pools over 3-4 tokens (arbitrage.py:5-28, liquidation.py:5-28, two-asset.py:7-32).
//
// In the grid-wide path such a solve is pure launch latency: ~11 us per outer iteration (two dependent launches) for a
// fraction of a microsecond of work, 300 evaluations = 3.5 ms.  Here ONE workgroup runs the WHOLE outer loop in ONE
// launch and nothing of the loop touches global memory except the pool columns (cache hits):
//   * every wave walks wave-tiles exactly as in eval_kernel (same tile code), but the psi tile STAYS in LDS -- no flush;
//   * wave 0 then takes the projected L-BFGS step as ONE WAVE: token j / price group r live on lane j / r (<= 64 of each),
//     every vector of the update is one register per lane, every reduction a DPP butterfly (no barrier, no LDS, no
//     memory), the history pairs sit in LDS; it writes the next trial prices into the LDS price table;
//   * one barrier, next evaluation.
// Same iteration as update_generic_body / oracle_step (price ties, bounds, any memory <= MAX_MEMORY), different layout.
// The state reaches global memory once, when the solve ends.
#pragma once
#include "onewave.hpp"

namespace cfmm {

constexpr int TINY_N = 64;                 // tokens (and price groups): one per lane of wave 0
constexpr int TINY_THREADS = 512;          // <= 8 waves walk the tiles (256 VGPRs per thread: tile code + update in one kernel)
constexpr int TINY_MAX_TILES = 64;

// LDS (doubles): eval_kernel<WITH_D>'s carve | exchange strips | q[64] | q2[64] | S[MAX_MEMORY][64] | Y[MAX_MEMORY][64] | ctl
__host__ __device__ inline int tiny_lds_doubles(int n)
{
    return eval_lds_doubles(n, true) + 2 * 64 * (TINY_THREADS / 64) + 2 * 64 + 2 * MAX_MEMORY * 64 + 4;
}

// BATCH (cfmm_solve_sweep: the 50-point sweep of two-asset.py:34-100 in ONE launch): workgroup b solves point b of a sweep over the
// SAME pools -- its own utility, price ties, start prices, tolerance and budget and its own tied-pool flags of the constant-sum
// bucket (a0.batch[b], device memory; UpdArgs::pool_flags).  Everything else is the one solve above.
template <bool BATCH = false>
__global__ void __launch_bounds__(TINY_THREADS)
solve_tiny_kernel(EvalArgs ev_in, UpdArgs a0, int iters_in)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const UpdArgs &a = upd_args<BATCH>(a0);
    const int iters = BATCH ? a.max_evals + 1 : iters_in;
    // (the point's own flags: a copy of the argument block with ONE pointer replaced -- every other field still comes out of the
    //  kernel arguments; the tile-range table, which indexes tile_end dynamically, is built from the arguments themselves)
    EvalArgs ev = ev_in;
    if (BATCH) ev.b2[2].flags = a.pool_flags;
    const int n = ev_in.n, ng = a.ng, M = a.M;
    const int tid = threadIdx.x, L = tid & 63, wave = wuni((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
    const int tile = eval_tile_doubles(n, false);
    double *psi_s = lds, *diag_s = lds + tile;
    double *nu_s = lds + 2 * tile;                      // [n + 1]
    double *fpart = nu_s + n + 2;                       // [16]
    int *next_tile = reinterpret_cast<int *>(fpart + 16);
    double *strips = lds + eval_lds_doubles(n, true);
    double2 *xs = reinterpret_cast<double2 *>(strips) + 64 * wave;
    double *q = strips + 2 * 64 * (TINY_THREADS / 64);  // [64] group sums / the exchange of the trial point
    double *q2 = q + 64;
    double *Sh = q2 + 64, *Yh = Sh + MAX_MEMORY * 64;   // history pairs, one row of 64 per slot
    int *status_s = reinterpret_cast<int *>(Yh + MAX_MEMORY * 64);

    WaveUpdate<1> u;                                    // wave 0's registers: lane r = group variable r, lane j = token j (onewave.hpp)
    if (wave == 0) u.load(a, L, n, ng);
    for (int j = tid; j < 2 * MAX_MEMORY * 64; j += blockDim.x) Sh[j] = 0.0;
    for (int j = tid; j <= n; j += blockDim.x) nu_s[j] = a.nu[j];
    if (tid == 0) *status_s = 0;

    for (int it = 0; it < iters; ++it) {
        for (int j = tid; j < 2 * tile; j += blockDim.x) lds[j] = 0.0;
        if (tid < 64) build_tile_table(ev_in, next_tile, tid, BATCH);   // (also re-arms the ticket counter; BATCH: every workgroup walks ALL tiles)
        __syncthreads();
        if (it == 0) eval_tiles_and_flush<true, false, false, false, false>(ev, nullptr, nu_s, psi_s, diag_s, fpart, next_tile, xs);
        else eval_tiles_and_flush<false, false, false, false, false>(ev, nullptr, nu_s, psi_s, diag_s, fpart, next_tile, xs);
        // (a barrier has been passed: the tiles and the per-wave partials of sum arb are complete)
        if (wave == 0) {
            const double psi[1] = {u.tin[0] ? psi_s[L] : 0.0};
            const double dg[1] = {(u.st.first != 0 && u.tin[0]) ? diag_s[L] : 0.0};
            const double fpools = wave_allsum(L < nw ? fpart[L] : 0.0);
            u.step(a, M, psi, dg, fpools, Sh, Yh, q, q2, nu_s);
            if (L == 0) *status_s = u.st.status;
        }
        __syncthreads();
        if (*status_s != 0) break;
    }
    if (wave == 0) u.store(a);
}

}  // namespace cfmm
