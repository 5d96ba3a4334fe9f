// HIP kernels of the dual-decomposition hot path (gfx950 / MI355X, wave64, fp64, no MFMA).
//
//   eval2_kernel<KIND>, evaln_kernel<K>   one dual evaluation of a pool bucket: stream the SoA
//        columns (coalesced, once), gather nu from an LDS copy, solve the pool, scatter-add
//        A_i(L_i - D_i) into an LDS tile of psi (ds_add_f64), then flush the tile to one of
//        `nslices` global accumulators (global_atomic_add_f64).        reference: arbitrage.py:54
//   update_kernel                         consumes the accumulators (after the all-reduce when
//        pool-sharded) and performs one step of the projected L-BFGS iteration on log-prices:
//        the on-device "nu update".                                     reference: arbitrage.py:82
//   trades2_kernel / tradesn_kernel       materialise Delta_i, Lambda_i at the accepted prices
//        (once per solve).                                              reference: two-asset.py:94,98
#pragma once
#include "pool_math.hpp"

namespace cfmm {

constexpr int EVAL_THREADS = 1024;    // two-asset kernels: <= 128 VGPRs is plenty
constexpr int EVALN_THREADS = 512;    // K-asset kernels: K-sized register arrays want up to 256 VGPRs
constexpr int UPD_THREADS = 1024;
constexpr int MAX_MEMORY = 16;

struct DevState {
    int status, evals, iters, first, hist, head, nrej, pad;
    double f, t_step, gap, infeas, primal, pg;
};

struct Bucket2 {
    long long m;
    const double *Ra, *Rb, *fee, *param;
    const int *ia, *ib, *flags;
};

struct BucketN {
    long long m;
    const int *idx;
    const double *R, *w, *fee;
};

// accumulator slice layout: [0,n) psi | [n] sum arb | [n+8, 2n+8) diag
__host__ __device__ inline int acc_stride(int n) { return 2 * n + 8; }

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}

// ------------------------------------------------------------------------------------------
// Prologue / epilogue shared by the evaluation kernels
// LDS: nu_s[n] | psi_s[n] | (diag_s[n]) | fpart[16]
// ------------------------------------------------------------------------------------------
template <bool WITH_D>
__device__ __forceinline__ void eval_prologue(double *lds, const double *__restrict__ nu, int n,
                                              double *&nu_s, double *&psi_s, double *&diag_s)
{
    nu_s = lds;
    psi_s = lds + n;
    diag_s = lds + 2 * n;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        nu_s[j] = nu[j];
        psi_s[j] = 0.0;
        if (WITH_D) diag_s[j] = 0.0;
    }
    __syncthreads();
}

template <bool WITH_D>
__device__ __forceinline__ void eval_epilogue(double fsum, double *psi_s, double *diag_s, double *fpart,
                                              int n, double *__restrict__ acc, int nslices)
{
    fsum = wave_sum(fsum);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    if (lane == 0) fpart[wave] = fsum;
    __syncthreads();
    double *base = acc + (size_t)(blockIdx.x % nslices) * acc_stride(n);
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const double v = psi_s[j];
        if (v != 0.0) unsafeAtomicAdd(&base[j], v);
        if (WITH_D) {
            const double dv = diag_s[j];
            if (dv != 0.0) unsafeAtomicAdd(&base[n + 8 + j], dv);
        }
    }
    if (threadIdx.x == 0) {
        double f = 0.0;
        for (int w = 0; w < nw; ++w) f += fpart[w];
        if (f != 0.0) unsafeAtomicAdd(&base[n], f);
    }
}

// ------------------------------------------------------------------------------------------
// two-asset buckets: one lane = one pool; 32 B (CP2, SUM2) or 40 B (W2, CURVE2) per pool
// ------------------------------------------------------------------------------------------
template <int KIND, bool WITH_D>
__global__ void __launch_bounds__(EVAL_THREADS)
eval2_kernel(Bucket2 b, const double *__restrict__ nu, int n, double *__restrict__ acc, int nslices,
             const DevState *__restrict__ st)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (st && st->status != 0) return;
    double *nu_s, *psi_s, *diag_s;
    eval_prologue<WITH_D>(lds, nu, n, nu_s, psi_s, diag_s);
    double *fpart = lds + (WITH_D ? 3 : 2) * n;

    double fsum = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < b.m; i += stride) {
        const double Ra = b.Ra[i], Rb = b.Rb[i], g = b.fee[i];
        const int ia = b.ia[i], ib = b.ib[i];
        const double pa = nu_s[ia], pb = nu_s[ib];
        Y2 y;
        if (KIND == 0) y = pool_cp2(Ra, Rb, g, pa, pb);
        else if (KIND == 1) y = pool_w2(Ra, Rb, g, b.param[i], pa, pb);
        else if (KIND == 2) { y = pool_sum2(Ra, Rb, g, pa, pb); if (b.flags && b.flags[i]) { y.ya = 0.0; y.yb = 0.0; } }
        else y = pool_curve2(Ra, Rb, g, b.param[i], pa, pb);
        if (y.ya != 0.0 || y.yb != 0.0) {
            unsafeAtomicAdd(&psi_s[ia], y.ya);
            unsafeAtomicAdd(&psi_s[ib], y.yb);
            fsum += pa * y.ya + pb * y.yb;
        }
        if (WITH_D) {
            double da = 0.0, db = 0.0;
            if (KIND == 0) { da = 0.5 * pa * Ra; db = 0.5 * pb * Rb; }
            else if (KIND == 1) { const double wa = b.param[i]; da = (1.0 - wa) * pa * Ra; db = wa * pb * Rb; }
            else if (KIND == 3) curve_diag(Ra, Rb, b.param[i], pa, pb, da, db);
            if (KIND != 2) { unsafeAtomicAdd(&diag_s[ia], da); unsafeAtomicAdd(&diag_s[ib], db); }
        }
    }
    __syncthreads();
    eval_epilogue<WITH_D>(fsum, psi_s, diag_s, fpart, n, acc, nslices);
}

// ------------------------------------------------------------------------------------------
// K-asset geo-mean buckets, slot-major ("size-class SoA"): column j of pool i at [j*m + i], so
// each of the 3K loads per lane is coalesced across the wave; 12 + 20 K bytes per pool.
// ------------------------------------------------------------------------------------------
template <int K, bool WITH_D>
__global__ void __launch_bounds__(EVALN_THREADS)
evaln_kernel(BucketN b, const double *__restrict__ nu, int n, double *__restrict__ acc, int nslices,
             const DevState *__restrict__ st)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (st && st->status != 0) return;
    double *nu_s, *psi_s, *diag_s;
    eval_prologue<WITH_D>(lds, nu, n, nu_s, psi_s, diag_s);
    double *fpart = lds + (WITH_D ? 3 : 2) * n;

    double fsum = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < b.m; i += stride) {
        double R[K], w[K], p[K], y[K];
        int t[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            t[j] = b.idx[(size_t)j * b.m + i];
            R[j] = b.R[(size_t)j * b.m + i];
            w[j] = b.w[(size_t)j * b.m + i];
        }
        const double g = b.fee[i];
#pragma unroll
        for (int j = 0; j < K; ++j) p[j] = nu_s[t[j]];
        pool_geomean_n<K>(R, w, g, p, y);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (y[j] != 0.0) { unsafeAtomicAdd(&psi_s[t[j]], y[j]); fsum += p[j] * y[j]; }
            if (WITH_D) unsafeAtomicAdd(&diag_s[t[j]], (1.0 - w[j]) * p[j] * R[j]);
        }
    }
    __syncthreads();
    eval_epilogue<WITH_D>(fsum, psi_s, diag_s, fpart, n, acc, nslices);
}

// ------------------------------------------------------------------------------------------
// trade materialisation (once per solve): Delta = max(-y,0), Lambda = max(y,0), slot-major
// ------------------------------------------------------------------------------------------
template <int KIND>
__global__ void __launch_bounds__(256)
trades2_kernel(Bucket2 b, const double *__restrict__ nu, double *__restrict__ delta, double *__restrict__ lambda)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.m) return;
    const double Ra = b.Ra[i], Rb = b.Rb[i], g = b.fee[i];
    const double pa = nu[b.ia[i]], pb = nu[b.ib[i]];
    Y2 y;
    if (KIND == 0) y = pool_cp2(Ra, Rb, g, pa, pb);
    else if (KIND == 1) y = pool_w2(Ra, Rb, g, b.param[i], pa, pb);
    else if (KIND == 2) { y = pool_sum2(Ra, Rb, g, pa, pb); if (b.flags && b.flags[i]) { y.ya = 0.0; y.yb = 0.0; } }
    else y = pool_curve2(Ra, Rb, g, b.param[i], pa, pb);
    delta[i] = fmax(-y.ya, 0.0);  delta[b.m + i] = fmax(-y.yb, 0.0);
    lambda[i] = fmax(y.ya, 0.0);  lambda[b.m + i] = fmax(y.yb, 0.0);
}

template <int K>
__global__ void __launch_bounds__(256)
tradesn_kernel(BucketN b, const double *__restrict__ nu, double *__restrict__ delta, double *__restrict__ lambda)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.m) return;
    double R[K], w[K], p[K], y[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        R[j] = b.R[(size_t)j * b.m + i];
        w[j] = b.w[(size_t)j * b.m + i];
        p[j] = nu[b.idx[(size_t)j * b.m + i]];
    }
    pool_geomean_n<K>(R, w, b.fee[i], p, y);
#pragma unroll
    for (int j = 0; j < K; ++j) {
        delta[(size_t)j * b.m + i] = fmax(-y[j], 0.0);
        lambda[(size_t)j * b.m + i] = fmax(y[j], 0.0);
    }
}

// ------------------------------------------------------------------------------------------
// fold the accumulator slices into slice 0 (used before the RCCL all-reduce and by eval_dual)
// ------------------------------------------------------------------------------------------
__global__ void fold_kernel(double *__restrict__ acc, int n, int nslices, int with_d, const DevState *st)
{
    if (st && st->status != 0) return;
    const int len = with_d ? acc_stride(n) : n + 1;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= len) return;
    double v = acc[j];
    for (int s = 1; s < nslices; ++s) {
        v += acc[(size_t)s * acc_stride(n) + j];
        acc[(size_t)s * acc_stride(n) + j] = 0.0;
    }
    acc[j] = v;
}

// ------------------------------------------------------------------------------------------
// The nu update: one workgroup, one step of projected L-BFGS on the group variables s
// (log nu_j = s[grp[j]] + off[j]).  Mirrors oracle/cfmm_oracle.c:oracle_step.
// ------------------------------------------------------------------------------------------
struct UpdArgs {
    int n, ng, M, nslices;
    double *acc;
    const double *c, *h, *off, *glo, *ghi;
    const int *ctype, *grp;
    double *nu, *nu_acc, *psi_acc, *psi_t;
    double *s, *s_t, *Gs, *Gs_t, *d, *Ds, *S, *Y, *rho;
    DevState *st;
    double tol_gap, tol_infeas, armijo, max_step;
    int max_evals, pg_rule;
};

// block-wide reduction of NV sums and NM maxima at once; result broadcast to every thread
template <int NV, int NM>
__device__ __forceinline__ void block_reduce(double (&v)[NV], double (&mx)[NM == 0 ? 1 : NM], double *scratch /* >= 16*(NV+NM) */)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) { const double r = wave_sum(v[k]); if (lane == 0) scratch[k * 16 + wave] = r; }
#pragma unroll
    for (int k = 0; k < NM; ++k) { const double r = wave_max(mx[k]); if (lane == 0) scratch[(NV + k) * 16 + wave] = r; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) { double r = 0.0; for (int w = 0; w < nw; ++w) r += scratch[k * 16 + w]; v[k] = r; }
#pragma unroll
    for (int k = 0; k < NM; ++k) { double r = scratch[(NV + k) * 16]; for (int w = 1; w < nw; ++w) r = fmax(r, scratch[(NV + k) * 16 + w]); mx[k] = r; }
    __syncthreads();
}

__device__ __forceinline__ bool is_active(double s, double lo, double hi, double G)
{
    return (s <= lo + 1e-14 && G > 0.0) || (s >= hi - 1e-14 && G < 0.0) || (lo == hi);
}

__global__ void __launch_bounds__(UPD_THREADS)
update_kernel(UpdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *q = lds;                         // [ng]
    double *q2 = lds + a.ng;                 // [ng]
    double *scratch = lds + 2 * a.ng;        // [16*8]
    DevState st = *a.st;
    if (st.status != 0) return;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n = a.n, ng = a.ng, M = a.M;
    const int stride = acc_stride(n);
    const bool ties = (ng != n);

    // ---- A. fold slices, residuals, group gradient at the trial point --------------------
    if (ties) { for (int r = tid; r < ng; r += nt) { q[r] = 0.0; q2[r] = 0.0; } }
    __syncthreads();
    double sums[2] = {0.0, 0.0};             // f_lin, gapv
    double maxs[2] = {0.0, 0.0};             // viol, scale
    for (int j = tid; j < n; j += nt) {
        double psi = 0.0, dg = 0.0;
        for (int sl = 0; sl < a.nslices; ++sl) {
            double *base = a.acc + (size_t)sl * stride;
            psi += base[j]; base[j] = 0.0;
            if (st.first) { dg += base[n + 8 + j]; base[n + 8 + j] = 0.0; }
        }
        a.psi_t[j] = psi;
        const double nuj = a.nu[j], hj = a.h[j], cj = a.c[j];
        const double rj = psi + hj;
        sums[0] += (nuj - cj) * hj;
        sums[1] += (nuj - cj) * rj;
        const int ct = a.ctype[j];
        maxs[0] = fmax(maxs[0], ct == 0 ? fmax(-rj, 0.0) : (ct == 1 ? fabs(rj) : 0.0));
        maxs[1] = fmax(maxs[1], fmax(fabs(psi), fabs(hj)));
        if (ties) {                          // group sums through LDS (ds_add_f64)
            unsafeAtomicAdd(&q[a.grp[j]], nuj * rj);
            if (st.first) unsafeAtomicAdd(&q2[a.grp[j]], dg);
        } else {
            a.Gs_t[j] = nuj * rj;
            if (st.first) a.Ds[j] = dg;
        }
    }
    double fpools = 0.0;
    if (tid == 0) {
        for (int sl = 0; sl < a.nslices; ++sl) { double *base = a.acc + (size_t)sl * stride; fpools += base[n]; base[n] = 0.0; }
    }
    sums[0] += fpools;                       // f_t = sum arb + (nu - c)'h
    block_reduce<2, 2>(sums, maxs, scratch);
    if (ties) {
        for (int r = tid; r < ng; r += nt) { a.Gs_t[r] = q[r]; if (st.first) a.Ds[r] = q2[r]; }
        __syncthreads();
    }
    const double f_t = sums[0], gapv = sums[1], viol = maxs[0], scale = maxs[1];
    st.evals += 1;

    // ---- B. accept test ------------------------------------------------------------------
    bool accept = st.first != 0;
    if (!st.first) {
        double dd[2] = {0.0, 0.0};
        double dummy[1] = {0.0};
        for (int r = tid; r < ng; r += nt) { const double ds = a.s_t[r] - a.s[r]; dd[0] += a.Gs[r] * ds; dd[1] += a.Gs_t[r] * ds; }
        block_reduce<2, 0>(dd, dummy, scratch);
        accept = (f_t == f_t) && ((f_t <= st.f + a.armijo * dd[0]) ||
                                  (f_t <= st.f + 1e-11 * fmax(1.0, fabs(st.f)) && dd[1] <= 0.8 * fabs(dd[0])));
    }

    if (!accept) {
        st.t_step *= 0.5;
        st.nrej += 1;
        if (st.t_step < 1e-9) st.status = 2;
    } else {
        // ---- C. curvature pair, move the accepted point ----------------------------------
        if (!st.first) {
            double *sv = a.S + (size_t)st.head * n, *yv = a.Y + (size_t)st.head * n;
            double t3[3] = {0.0, 0.0, 0.0};
            double dummy[1] = {0.0};
            for (int r = tid; r < ng; r += nt) {
                const double s1 = a.s_t[r] - a.s[r], y1 = a.Gs_t[r] - a.Gs[r];
                sv[r] = s1; yv[r] = y1;
                t3[0] += s1 * y1; t3[1] += s1 * s1; t3[2] += y1 * y1;
            }
            block_reduce<3, 0>(t3, dummy, scratch);
            if (t3[0] > 1e-12 * sqrt(t3[1]) * sqrt(t3[2])) {
                if (tid == 0) a.rho[st.head] = 1.0 / t3[0];
                st.head = (st.head + 1) % M;
                if (st.hist < M) st.hist += 1;
            }
            st.iters += 1;
        }
        for (int r = tid; r < ng; r += nt) { a.s[r] = a.s_t[r]; a.Gs[r] = a.Gs_t[r]; }
        for (int j = tid; j < n; j += nt) { a.psi_acc[j] = a.psi_t[j]; a.nu_acc[j] = a.nu[j]; }
        st.f = f_t; st.first = 0;
        st.gap = fabs(gapv) / fmax(1.0, fabs(f_t));
        st.infeas = viol / fmax(scale, 1e-300);
        st.primal = f_t - gapv;               // c'psi = g - (nu - c)'(psi + h)
        // value of the projected reduced gradient: sum_r |P(Gs)_r| / max(1,|f|)  (>= gap)
        {
            double pgs[1] = {0.0};
            double dummy[1] = {0.0};
            for (int r = tid; r < ng; r += nt) {
                const double G = a.Gs[r], sr = a.s[r];
                double v = G;
                if (a.glo[r] == a.ghi[r]) v = 0.0;
                else if (sr <= a.glo[r] + 1e-14) v = fmin(G, 0.0);
                else if (sr >= a.ghi[r] - 1e-14) v = fmax(G, 0.0);
                pgs[0] += fabs(v);
            }
            __syncthreads();
            block_reduce<1, 0>(pgs, dummy, scratch);
            st.pg = pgs[0] / fmax(1.0, fabs(f_t));
        }
        const bool conv = a.pg_rule ? (st.pg <= a.tol_gap) : (st.gap <= a.tol_gap && st.infeas <= a.tol_infeas);
        if (conv) {
            st.status = 1;
        } else {
            // ---- D. two-loop recursion with the diagonal metric --------------------------
            double gp[1] = {0.0};
            double dummy[1] = {0.0};
            for (int r = tid; r < ng; r += nt) {
                const double G = a.Gs[r];
                const double v = is_active(a.s[r], a.glo[r], a.ghi[r], G) ? 0.0 : G;
                q[r] = v; gp[0] += v * v;
            }
            __syncthreads();
            block_reduce<1, 0>(gp, dummy, scratch);
            double alpha[MAX_MEMORY];
#pragma unroll
            for (int k = 0; k < MAX_MEMORY; ++k) {
                if (k < st.hist) {
                    const int i = (st.head - 1 - k + 2 * M) % M;
                    const double *sv = a.S + (size_t)i * n, *yv = a.Y + (size_t)i * n;
                    double dt[1] = {0.0};
                    for (int r = tid; r < ng; r += nt) dt[0] += sv[r] * q[r];
                    block_reduce<1, 0>(dt, dummy, scratch);
                    const double al = a.rho[i] * dt[0];
                    alpha[k] = al;
                    for (int r = tid; r < ng; r += nt) q[r] -= al * yv[r];
                }
            }
            for (int r = tid; r < ng; r += nt) {
                const double H = a.Ds[r] + fmax(a.Gs[r], 0.0);
                q[r] = H > 0.0 ? q[r] / H : 0.0;
            }
#pragma unroll
            for (int k = MAX_MEMORY - 1; k >= 0; --k) {
                if (k < st.hist) {
                    const int i = (st.head - 1 - k + 2 * M) % M;
                    const double *sv = a.S + (size_t)i * n, *yv = a.Y + (size_t)i * n;
                    double dt[1] = {0.0};
                    for (int r = tid; r < ng; r += nt) dt[0] += yv[r] * q[r];
                    block_reduce<1, 0>(dt, dummy, scratch);
                    const double beta = a.rho[i] * dt[0];
                    for (int r = tid; r < ng; r += nt) q[r] += sv[r] * (alpha[k] - beta);
                }
            }
            double dsum[1] = {0.0};
            double dmx[1] = {0.0};
            for (int r = tid; r < ng; r += nt) {
                const double G = a.Gs[r];
                const double dv = is_active(a.s[r], a.glo[r], a.ghi[r], G) ? 0.0 : -q[r];
                a.d[r] = dv; dsum[0] += dv * G; dmx[0] = fmax(dmx[0], fabs(dv));
            }
            block_reduce<1, 1>(dsum, dmx, scratch);
            if (!(dsum[0] < 0.0) && gp[0] > 0.0) {        // not a descent direction: restart
                st.hist = 0;
                dmx[0] = 0.0;
                double z1[1] = {0.0};
                for (int r = tid; r < ng; r += nt) {
                    const double G = a.Gs[r];
                    const double H = a.Ds[r] + fmax(G, 0.0);
                    const double dv = (is_active(a.s[r], a.glo[r], a.ghi[r], G) || !(H > 0.0)) ? 0.0 : -G / H;
                    a.d[r] = dv; dmx[0] = fmax(dmx[0], fabs(dv));
                }
                block_reduce<1, 1>(z1, dmx, scratch);
            }
            st.t_step = (dmx[0] > a.max_step) ? a.max_step / dmx[0] : 1.0;
        }
    }
    __syncthreads();

    // ---- E. next trial point ---------------------------------------------------------------
    if (st.status == 0) {
        for (int r = tid; r < ng; r += nt) {
            double v = a.s[r] + st.t_step * a.d[r];
            v = fmax(v, a.glo[r]);
            v = fmin(v, a.ghi[r]);
            a.s_t[r] = v;
        }
        __syncthreads();
        for (int j = tid; j < n; j += nt) a.nu[j] = exp(a.s_t[a.grp[j]] + a.off[j]);
        if (st.evals >= a.max_evals) st.status = 3;
    }
    if (tid == 0) *a.st = st;
}

// start of a solve: group variable = mean over members of (log nu0_j - off_j), clamped
__global__ void __launch_bounds__(UPD_THREADS)
start_kernel(UpdArgs a, const double *__restrict__ nu0)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *sum = lds, *cnt = lds + a.ng;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n = a.n, ng = a.ng;
    for (int r = tid; r < ng; r += nt) { sum[r] = 0.0; cnt[r] = 0.0; }
    __syncthreads();
    for (int j = tid; j < n; j += nt) {
        unsafeAtomicAdd(&sum[a.grp[j]], log(nu0[j]) - a.off[j]);
        unsafeAtomicAdd(&cnt[a.grp[j]], 1.0);
    }
    __syncthreads();
    for (int r = tid; r < ng; r += nt) {
        double v = sum[r] / fmax(cnt[r], 1.0);
        v = fmax(v, a.glo[r]);
        v = fmin(v, a.ghi[r]);
        sum[r] = v;
        a.s_t[r] = v; a.s[r] = v; a.d[r] = 0.0;
    }
    __syncthreads();
    for (int j = tid; j < n; j += nt) { const double v = exp(sum[a.grp[j]] + a.off[j]); a.nu[j] = v; a.nu_acc[j] = v; }
    if (tid == 0) {
        DevState st;
        st.status = 0; st.evals = 0; st.iters = 0; st.first = 1; st.hist = 0; st.head = 0; st.nrej = 0; st.pad = 0;
        st.f = 0.0; st.t_step = 1.0; st.gap = 0.0; st.infeas = 0.0; st.primal = 0.0; st.pg = 0.0;
        *a.st = st;
    }
}

}  // namespace cfmm
