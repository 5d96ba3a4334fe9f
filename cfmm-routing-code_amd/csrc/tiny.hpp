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
#include "kernels.hpp"

namespace cfmm {

constexpr int TINY_N = 64;                 // tokens (and price groups): one per lane of wave 0
constexpr int TINY_THREADS = 512;          // <= 8 waves walk the tiles (256 VGPRs per thread: tile code + update in one kernel)
constexpr int TINY_MAX_TILES = 64;

// LDS (doubles): eval_kernel<WITH_D>'s carve | exchange strips | q[64] | q2[64] | S[MAX_MEMORY][64] | Y[MAX_MEMORY][64] | ctl
__host__ __device__ inline int tiny_lds_doubles(int n)
{
    return eval_lds_doubles(n, true) + 2 * 64 * (TINY_THREADS / 64) + 2 * 64 + 2 * MAX_MEMORY * 64 + 4;
}

__device__ __forceinline__ int tuni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double tuni(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ double lane_value(double v, int l)          // l wave-uniform
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

__global__ void __launch_bounds__(TINY_THREADS)
solve_tiny_kernel(EvalArgs ev, UpdArgs a, int iters)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = ev.n, ng = a.ng, M = a.M;
    const int tid = threadIdx.x, L = tid & 63, wave = tuni((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
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

    const bool tin = L < n, gin = L < ng, ties = ng != n;
    // wave 0's registers: lane r = group variable r, lane j = token j
    DevState st = {};
    double s = 0.0, s_t = 0.0, Gs = 0.0, d = 0.0, Ds = 0.0, glo = 0.0, ghi = 0.0;
    double cj = 0.0, hj = 0.0, offj = 0.0, nuj = 0.0, psi_a = 0.0, nu_a = 0.0, rho_l = 0.0;
    int ct = 0, grp = 0;
    if (wave == 0) {
        st = *a.st;
        if (gin) { s = a.s[L]; s_t = a.s_t[L]; d = a.d[L]; glo = a.glo[L]; ghi = a.ghi[L]; }
        if (tin) { cj = a.c[L]; hj = a.h[L]; offj = a.off[L]; ct = a.ctype[L]; grp = a.grp[L]; nuj = a.nu[L]; nu_a = a.nu_acc[L]; }
        st.status = tuni(st.status); st.evals = tuni(st.evals); st.iters = tuni(st.iters); st.first = tuni(st.first);
        st.hist = tuni(st.hist); st.head = tuni(st.head); st.nrej = tuni(st.nrej);
        st.f = tuni(st.f); st.t_step = tuni(st.t_step);
    }
    for (int j = tid; j < 2 * MAX_MEMORY * 64; j += blockDim.x) Sh[j] = 0.0;
    for (int j = tid; j <= n; j += blockDim.x) nu_s[j] = a.nu[j];
    if (tid == 0) *status_s = 0;

    for (int it = 0; it < iters; ++it) {
        for (int j = tid; j < 2 * tile; j += blockDim.x) lds[j] = 0.0;
        if (tid < 64) build_tile_table(ev, next_tile, tid);          // (also re-arms the ticket counter)
        __syncthreads();
        if (it == 0) eval_tiles_and_flush<true, false, false, false, false>(ev, nullptr, nu_s, psi_s, diag_s, fpart, next_tile, xs);
        else eval_tiles_and_flush<false, false, false, false, false>(ev, nullptr, nu_s, psi_s, diag_s, fpart, next_tile, xs);
        // (a barrier has been passed: the tiles and the per-wave partials of sum arb are complete)
        if (wave == 0) {
            const bool first = st.first != 0;
            // ---- A. residuals, group gradient at the trial point ---------------------------------------------
            const double psi = tin ? psi_s[L] : 0.0;
            const double dg = (first && tin) ? diag_s[L] : 0.0;
            const double fpools = wave_allsum(L < nw ? fpart[L] : 0.0);
            const double rj = psi + hj;
            const double f_t = wave_allsum(tin ? (nuj - cj) * hj : 0.0) + fpools;
            const double gapv = wave_allsum(tin ? (nuj - cj) * rj : 0.0);
            const double viol = wave_allmax(!tin ? 0.0 : (ct == 0 ? fmax(-rj, 0.0) : (ct == 1 ? fabs(rj) : 0.0)));
            const double scale = wave_allmax(tin ? fmax(fabs(psi), fabs(hj)) : 0.0);
            double Gs_t;
            if (ties) {                                  // group sums through LDS (one wave: its LDS operations execute in order)
                q[L] = 0.0; q2[L] = 0.0;
                if (tin) { unsafeAtomicAdd(&q[grp], nuj * rj); if (first) unsafeAtomicAdd(&q2[grp], dg); }
                Gs_t = gin ? q[L] : 0.0;
                if (first) Ds = gin ? q2[L] : 0.0;
            } else {
                Gs_t = tin ? nuj * rj : 0.0;
                if (first) Ds = dg;
            }
            st.evals += 1;

            // ---- B. accept test --------------------------------------------------------------------------------
            const double ds = gin ? s_t - s : 0.0;
            bool accept = first;
            if (!first) {
                const double dd0 = wave_allsum(Gs * ds), dd1 = wave_allsum(Gs_t * ds);
                accept = (f_t == f_t) && ((f_t <= st.f + a.armijo * dd0) ||
                                          (f_t <= st.f + 1e-11 * fmax(1.0, fabs(st.f)) && dd1 <= 0.8 * fabs(dd0)));
            }
            if (!accept) {
                st.t_step *= 0.5;
                st.nrej += 1;
                if (st.t_step < 1e-9) st.status = 2;
            } else {
                // ---- C. curvature pair, move the accepted point ----------------------------------------------
                if (!first) {
                    const double y1 = gin ? Gs_t - Gs : 0.0;
                    Sh[st.head * 64 + L] = ds; Yh[st.head * 64 + L] = y1;
                    const double sy = wave_allsum(ds * y1), ss = wave_allsum(ds * ds), yy = wave_allsum(y1 * y1);
                    if (sy > 1e-12 * sqrt(ss) * sqrt(yy)) {
                        if (L == st.head) rho_l = 1.0 / sy;
                        st.head = (st.head + 1) % M;
                        if (st.hist < M) st.hist += 1;
                    }
                    st.iters += 1;
                }
                s = s_t; Gs = Gs_t; psi_a = psi; nu_a = nuj;
                st.f = f_t; st.first = 0;
                st.gap = fabs(gapv) / fmax(1.0, fabs(f_t));
                st.infeas = viol / fmax(scale, 1e-300);
                st.primal = f_t - gapv;               // c'psi = g - (nu - c)'(psi + h)
                {
                    double v = Gs;
                    if (glo == ghi) v = 0.0;
                    else if (s <= glo + 1e-14) v = fmin(Gs, 0.0);
                    else if (s >= ghi - 1e-14) v = fmax(Gs, 0.0);
                    st.pg = wave_allsum(gin ? fabs(v) : 0.0) / fmax(1.0, fabs(f_t));
                }
                const bool conv = a.pg_rule ? (st.pg <= a.tol_gap) : (st.gap <= a.tol_gap && st.infeas <= a.tol_infeas);
                if (conv) {
                    st.status = 1;
                } else {
                    // ---- D. two-loop recursion with the diagonal metric ------------------------------------
                    const bool active = !gin || is_active(s, glo, ghi, Gs);
                    double qv = active ? 0.0 : Gs;
                    const double gp = wave_allsum(qv * qv);
                    double alpha[MAX_MEMORY];
#pragma unroll
                    for (int k = 0; k < MAX_MEMORY; ++k) {
                        alpha[k] = 0.0;
                        if (k < st.hist) {
                            const int i = (st.head - 1 - k + 2 * M) % M;
                            const double al = lane_value(rho_l, i) * wave_allsum(Sh[i * 64 + L] * qv);
                            alpha[k] = al;
                            qv -= al * Yh[i * 64 + L];
                        }
                    }
                    const double H = Ds + fmax(Gs, 0.0);
                    qv = (gin && H > 0.0) ? qv / H : 0.0;
#pragma unroll
                    for (int k = MAX_MEMORY - 1; k >= 0; --k) {
                        if (k < st.hist) {
                            const int i = (st.head - 1 - k + 2 * M) % M;
                            const double beta = lane_value(rho_l, i) * wave_allsum(Yh[i * 64 + L] * qv);
                            qv += Sh[i * 64 + L] * (alpha[k] - beta);
                        }
                    }
                    double dv = active ? 0.0 : -qv;
                    const double dsum = wave_allsum(dv * Gs);
                    double dmx = wave_allmax(fabs(dv));
                    if (!(dsum < 0.0) && gp > 0.0) {        // not a descent direction: restart
                        st.hist = 0;
                        dv = (active || !(H > 0.0)) ? 0.0 : -Gs / H;
                        dmx = wave_allmax(fabs(dv));
                    }
                    d = dv;
                    st.t_step = (dmx > a.max_step) ? a.max_step / dmx : 1.0;
                }
            }
            // ---- E. next trial point: straight into the LDS price table ------------------------------------------
            if (st.status == 0) {
                double v = s + st.t_step * d;
                v = fmax(v, glo);
                v = fmin(v, ghi);
                s_t = gin ? v : 0.0;
                q[L] = s_t;                              // (same wave: ordered)
                nuj = tin ? exp(q[grp] + offj) : 0.0;
                if (tin) nu_s[L] = nuj;
                if (st.evals >= a.max_evals) st.status = 3;
            }
            st.status = tuni(st.status); st.hist = tuni(st.hist); st.head = tuni(st.head);
            if (L == 0) *status_s = st.status;
        }
        __syncthreads();
        if (*status_s != 0) break;
    }
    if (wave == 0) {                                     // the state reaches global memory once
        if (tin) { a.nu_acc[L] = nu_a; a.psi_acc[L] = psi_a; a.nu[L] = nuj; }
        if (gin) { a.s[L] = s; a.s_t[L] = s_t; a.Gs[L] = Gs; a.d[L] = d; a.Ds[L] = Ds; }
        if (L == 0) {
            if (st.status == 0) st.status = 3;           // (the launch's budget is the solve's)
            *a.st = st; a.nu[n] = 1.0; report_progress(a, st);
        }
    }
}

}  // namespace cfmm
