// iter_kernel: ONE launch per outer iteration -- the nu update folded into the evaluation launch (gfx950, wave64, fp64).
//
// Round 1 ran an outer iteration as two dependent launches: eval_kernel (256 workgroups) -> update_gram_kernel (ONE
// workgroup: 255 CUs idle for 8.6 us) -> next eval_kernel, i.e. two launch boundaries (~2 us each) and a serial section
// per iteration: 28.5 us of device time for a 15-16 us evaluation.  Here EVERY workgroup of the evaluation launch
// performs the (cheap, latency-bound) update itself in its prologue -- redundantly and bit-identically: same code, same
// inputs, same reduction order -- from the accumulators the previous launch flushed, writes the new prices straight into
// its LDS copy and goes on to its tiles.  Workgroup 0 alone stores the new solver state.
//                                                                        reference: arbitrage.py:82 (prob.solve())
// No grid barrier, no flags: what one launch writes is only read by the NEXT launch.  That needs rotation:
//   * accumulators, three sets: launch t reads A[(t-1) % 3] (complete: flushed by launch t-1), flushes into A[t % 3],
//     and one of its workgroups zeroes A[(t+1) % 3] (last read by launch t-1, next flushed into by launch t+1);
//   * solver state (s, s_t, Gs, d, trial prices, DevState), three sets: read X[(t-1) % 3], write X[t % 3];
//   * the L-BFGS history is a ring of M + 1 slots for a window of M pairs, so the slot a launch writes is never one
//     another workgroup of the same launch still reads.
// The phase t % 3 is a kernel argument; captured graphs hold a multiple of three launches.
// Pool-sharded: launch t -> fold A[t % 3] -> ncclAllReduce(slice 0) -> launch t+1 reads that one slice.
//
// The update itself is the Gram form of kernels.hpp (update_gram_kernel; mirrors oracle/cfmm_oracle.c:oracle_step) with a
// history of 3 pairs: all 32 scalars the accept test, the stopping rule and the two-loop recursion need come out of ONE
// batched reduce-scatter.  To fit the evaluation kernel's register budget (128 VGPRs at 16 waves per CU; the
// stand-alone Gram kernel uses 223) the per-lane products are never alive together: values i and i + 16 are produced
// as a pair and immediately exchanged across row pairs (v_permlane16_swap), which leaves 16 running values; every thread
// owns E = 2 variables, the update runs in two halves that reload what they need instead of holding it, and its uniform
// scalars live in SGPRs (v_readfirstlane / v_readlane).
#pragma once
#include <utility>
#include "kernels.hpp"

namespace cfmm {

#ifndef ITER_E_SMALL_DEF
#define ITER_E_SMALL_DEF 2
#endif
constexpr int ITER_E_SMALL = ITER_E_SMALL_DEF;  // variables per thread of the in-launch update up to EVAL_THREADS tokens (2 beyond)
// history of the in-launch update: 3 pairs.  On the BASELINE configs memory 3 needs no more evaluations than 4 (mean over
// 12 instances 32.0 against 34.4; 2: 39.3, 5: 34.5, 6: 34.7) and makes the batch exactly 32 scalars -- one register-lean
// reduce-scatter, two vectors less to load in every workgroup
constexpr int ITER_MM = 3;                  // history pairs kept by the in-launch update
constexpr int ITER_P = ITER_MM + 1;         // + the new pair
// layout of the batched scalars (8 + 2 P + P (P - 1) / 2 + P (P + 1) / 2 = 32 for P = 4)
constexpr int GI_U = 8, GI_V = GI_U + ITER_P, GI_SY = GI_V + ITER_P, GI_YHY = GI_SY + ITER_P * (ITER_P - 1) / 2,
              GI_END = GI_YHY + ITER_P * (ITER_P + 1) / 2;
static_assert(GI_END <= 32, "the batch must fit one 32-value reduce-scatter");
constexpr int XS_HIST = 5;                  // s | s_t | Gs | d | trial prices | S window (ITER_MM, newest first) | Y window, per state set
constexpr int XS_VECS = XS_HIST + 2 * ITER_MM;
__host__ __device__ inline int iter_xvs(int n) { return (n + 3) & ~1; }      // vector stride of a state set: >= n + 2 (the stop flag rides at [n]), even

constexpr int ITER_HRING = 1024;            // slots of the per-launch progress ring (> two chunks of launches in flight: iters_per_graph <= 256)
__host__ __device__ inline unsigned long long iter_hring_word(int evals, int status, int launch)
{
    return (unsigned long long)((unsigned)evals & 0xffffffu) | ((unsigned long long)((unsigned)status & 0xffu) << 24) | ((unsigned long long)(unsigned)launch << 32);
}
struct IterArgs {
    EvalArgs ev;                    // tile space of the evaluation (ev.nu / ev.acc are not used here)
    int n, M, nread, phase;         // nread: accumulator slices to read (nslices, or 1 behind an all-reduce)
    int xvs, max_evals, pg_rule;
    int plain;                      // 1: h == 0, every token CFMM_GE, no upper bounds (the linear-utility arbitrage of arbitrage.py:57,77):
                                    //    three of the update's vectors need not be read
    double *acc3; long long acc_set;        // 3 sets of nslices * acc_stride(n) doubles
    double *xs; long long xs_set;           // 3 sets of XS_VECS * xvs doubles
    DevState *st3;                          // 3 sets
    const double *c, *h, *glo, *ghi;
    const int *ctype;
    double *Ds;
    double *nu, *nu_acc, *psi_acc;          // written by workgroup 0: trial prices (+ stop flag at [n]), accepted point
    double tol_gap, tol_infeas, armijo, max_step;
    unsigned long long *hstat;              // pinned HOST word (zero-copy): evals | status << 32, for the host's run-ahead control
    // pinned HOST ring (zero-copy; null = none), one slot per launch: evals | status << 24 | launch << 32, written by EVERY launch (the idle
    // ones behind the end of a solve too).  The pool-sharded host loop decides on the slot of a FIXED launch -- the last of the chunk
    // before the one it has just enqueued -- so every rank takes the same decision whatever its device's pace (round 6: no copy, no event)
    unsigned long long *hring;
    int launch;
    // pinned HOST mirrors (mapped; null = none): the accepted prices / net trade are stored there as well whenever a point is
    // accepted, the state record when the solve ends -- the host reads its result after one synchronisation, no copies
    double *h_nu_acc, *h_psi_acc;
    DevState *h_final;
};

template <int E> __device__ __forceinline__ void ldE(const double *p, int first, double (&v)[E]);
template <> __device__ __forceinline__ void ldE<1>(const double *p, int first, double (&v)[1]) { v[0] = p[first]; }
template <> __device__ __forceinline__ void ldE<2>(const double *p, int first, double (&v)[2])
{
    const double2 t = *reinterpret_cast<const double2 *>(p + first);
    v[0] = t.x; v[1] = t.y;
}
template <int E> __device__ __forceinline__ void ldEi(const int *p, int first, int (&v)[E]);
template <> __device__ __forceinline__ void ldEi<1>(const int *p, int first, int (&v)[1]) { v[0] = p[first]; }
template <> __device__ __forceinline__ void ldEi<2>(const int *p, int first, int (&v)[2])
{
    const int2 t = *reinterpret_cast<const int2 *>(p + first);
    v[0] = t.x; v[1] = t.y;
}
// store E adjacent doubles starting at `first`, never touching index >= len (first is a multiple of E)
template <int E> __device__ __forceinline__ void stE(double *p, int first, int len, const double (&v)[E])
{
    if (E == 2 && first + 1 < len) { *reinterpret_cast<double2 *>(p + first) = make_double2(v[0], v[E - 1]); return; }
    if (first < len) p[first] = v[0];
}

// store E adjacent doubles at `first` < n into a vector of a state set: its stride (iter_xvs) leaves room behind element
// n - 1, so the store is always whole (no tail variants: lanes beyond the end hold zeros)
template <int E> __device__ __forceinline__ void stX(double *p, int first, const double (&v)[E])
{
    if (E == 2) *reinterpret_cast<double2 *>(p + first) = make_double2(v[0], v[E - 1]);
    else p[first] = v[0];
}

// what one thread contributes to the batched scalars:
//   0 f_lin 1 gapv 2 Gs.ds 3 Gs_t.ds 4 s.y 5 s.s 6 y.y 7 pg | GI_U.. u_k = s_k.q0 | GI_V.. v_k = y_k.H0 q0
//   GI_SY.. SY[k][j] = s_k.y_j (k > j) | GI_YHY.. YHY[k][j] = y_k.H0 y_j (k >= j)          pairs newest first, 0 = the new one
template <int E>
struct GramIn {
    double g1[8];
    double S[ITER_P][E], Y[ITER_P][E], q0[E], H0[E], hq[E];
};
__host__ __device__ constexpr int sy_row(int c) { int k = 1; while (k * (k + 1) / 2 <= c) ++k; return k; }            // c = k(k-1)/2 + j, j < k
__host__ __device__ constexpr int yhy_row(int c) { int k = 0; while ((k + 1) * (k + 2) / 2 <= c) ++k; return k; }     // c = k(k+1)/2 + j, j <= k
template <int I, int E>
__device__ __forceinline__ double gram_val(const GramIn<E> &in)
{
    double r = 0.0;
    if constexpr (I < GI_U) r = in.g1[I];
    else if constexpr (I < GI_V) {
#pragma unroll
        for (int e = 0; e < E; ++e) r = fma(in.S[I - GI_U][e], in.q0[e], r);
    } else if constexpr (I < GI_SY) {
#pragma unroll
        for (int e = 0; e < E; ++e) r = fma(in.Y[I - GI_V][e], in.hq[e], r);
    } else if constexpr (I < GI_YHY) {
        constexpr int c = I - GI_SY, k = sy_row(c), j = c - k * (k - 1) / 2;
        static_assert(j >= 0 && j < k && k < ITER_P, "SY index");
#pragma unroll
        for (int e = 0; e < E; ++e) r = fma(in.S[k][e], in.Y[j][e], r);
    } else if constexpr (I < GI_END) {
        constexpr int c = I - GI_YHY, k = yhy_row(c), j = c - k * (k + 1) / 2;
        static_assert(j >= 0 && j <= k && k < ITER_P, "YHY index");
#pragma unroll
        for (int e = 0; e < E; ++e) r = fma(in.H0[e] * in.Y[k][e], in.Y[j][e], r);
    }
    return r;
}

// The batched wave reduction, register-lean (the evaluation kernel leaves ~128 VGPRs per wave): values i and i + 16 are
// produced as a pair and exchanged across row pairs at once (v_permlane16_swap: even rows keep i, odd rows i + 16), so
// only 16 running values exist; four halving steps inside the rows of 16 lanes (DPP row_ror:8, ds_swizzle xor 4,
// quad_perm) and one xor-32 butterfly leave the wave total of quantity l in lane l (l < 32).  32 exchange-adds; the
// 64-value form of kernels.hpp needs 63 and 128 VGPRs of running values.
template <int I, int E>
__device__ __forceinline__ double gram_pair16(const GramIn<E> &in)
{
    const double a = gram_val<I, E>(in), b = gram_val<I + 16, E>(in);
    const int l0 = __double2loint(a), h0 = __double2hiint(a);
    const int l1 = __double2loint(b), h1 = __double2hiint(b);
    const auto lo = __builtin_amdgcn_permlane16_swap(l0, l1, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(h0, h1, false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
template <int E, int... I>
__device__ __forceinline__ void gram_pairs16(const GramIn<E> &in, double (&v)[16], std::integer_sequence<int, I...>)
{
    ((v[I] = gram_pair16<I, E>(in)), ...);
}
// halving steps 4, 2, 1 inside a row of 16 lanes on v[0..7]: lane l ends with the row total of quantity (l & 7) [+ 8 for b8 lanes]
__device__ __forceinline__ double row_halve4(double (&v)[8], int lane)
{
    const bool b4 = lane & 4, b2 = lane & 2, b1 = lane & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double keep = b4 ? v[i + 4] : v[i], send = b4 ? v[i] : v[i + 4];
        v[i] = keep + swz_xor4(send);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double keep = b2 ? v[i + 2] : v[i], send = b2 ? v[i] : v[i + 2];
        v[i] = keep + dpp_f64<0x4E>(send);
    }
    const double keep = b1 ? v[1] : v[0], send = b1 ? v[0] : v[1];
    return keep + dpp_f64<0xB1>(send);
}
// returns in lane l < 32 the wave total of quantity l (lanes 32..63 hold a copy)
template <int E>
__device__ __forceinline__ double gram_reduce32(const GramIn<E> &in, int lane)
{
    const bool b8 = lane & 8;
    double v[16];
    gram_pairs16<E>(in, v, std::make_integer_sequence<int, 16>{});
    double w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double keep = b8 ? v[i + 8] : v[i], send = b8 ? v[i] : v[i + 8];
        w[i] = keep + dppd_ror8(send);
    }
    const double r = row_halve4(w, lane);
    double x, y;
    swap32_f64(r, x, y);
    return x + y;
}

// self-test of gram_reduce32 (cfmm_selftest): small-integer inputs make every product and sum exact, the reference
// is one plain butterfly per quantity (wave_allsum, itself checked by selftest_kernel); comparison is bitwise
template <int... I>
__device__ __forceinline__ int gram_selfcheck(const GramIn<1> &in, int lane, double q, std::integer_sequence<int, I...>)
{
    int bad = 0;
    (([&] {
        const double want = wave_allsum(gram_val<I, 1>(in));
        if ((lane & 31) == I && q != want) ++bad;
    }()), ...);
    return bad;
}
__global__ void __launch_bounds__(64)
selftest_gram_kernel(int *out)
{
    const int lane = threadIdx.x;
    GramIn<1> in;
#pragma unroll
    for (int i = 0; i < 8; ++i) in.g1[i] = (double)(((lane + 2) * (i + 3)) % 17) - 8.0;
#pragma unroll
    for (int k = 0; k < ITER_P; ++k) {
        in.S[k][0] = (double)(((lane + 5) * (k + 2)) % 11) - 5.0;
        in.Y[k][0] = (double)(((lane + 1) * (k + 7)) % 13) - 6.0;
    }
    in.q0[0] = (double)((lane * 7) % 9) - 4.0;
    in.H0[0] = (double)((lane % 3) + 1);
    in.hq[0] = in.H0[0] * in.q0[0];
    const double q = gram_reduce32<1>(in, lane);
    atomicAdd(out, gram_selfcheck(in, lane, q, std::make_integer_sequence<int, 32>{}));
}

// a wave-uniform value held in SGPRs instead of one VGPR pair per lane (the update is register-bound)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// extra LDS of iter_kernel behind eval_kernel's carve: the bounds (every workgroup) and the trial point's prices and net
// trade (workgroup 0, which stores them when the point is accepted), stashed between the two halves of the update
//   staged tile walk (DMA, kernels.hpp): the bounds, and (in front of everything) one 4 KB slot per wave; the accepted-point
//   stash of the two workgroups that store it borrows the LAST slots (their waves issue their first DMA behind the update)
__host__ __device__ inline int iter_extra_lds_doubles(int n, bool dma = false)
{
    return (dma ? 2 * iter_xvs(n) + (STAGE_BYTES / 8) * (EVAL_THREADS / 64) : 4 * iter_xvs(n)) + 32;      // (+ the safeguards' partials)
}

// PLAIN: the utility has h == 0, every token CFMM_GE and no upper bounds (linear-utility arbitrage, arbitrage.py:57,77):
// three of the update's vectors are never read and their registers do not exist (the other instantiation spills a few)
//
// Round 3: the update as a LATENCY CHAIN.  Measured (phase timers, C3): its ~8 us were not arithmetic but (i) two
// dependent global round trips (state record -> history ring slots / rho, addressed through the ring head), (ii) a scalar
// section of ~650 instructions issued by EVERY wave with variables (2 per SIMD at 1000 tokens, 4 at 2000: 14 us there),
// (iii) waves without variables executing the direction / reduction code on zeros beside the ones that had work, (iv)
// five barriers.  Now: the history window travels in the rotating state set in WINDOW ORDER and its rho's in the state
// record, so every load address depends on the rotation phase alone and all loads go out at once; ONE wave runs the scalar
// section and hands ten numbers over through LDS; waves without variables only meet the others at the barriers (one of
// them lays out the tile-range table meanwhile); the next trial point is computed speculatively at step 1 under the
// direction's reduction; every thread recycles its OWN stash entries (trial point -> price, trial gradient -> zeroed psi
// entry), which merges the last two barriers into one.
// DMA: the evaluation's tiles come through the staged walk (kernels.hpp), and every wave's FIRST tile is requested as soon
// as the scalar section has decided that this launch evaluates at all -- its columns arrive while the direction and the
// trial point are still being formed
// NT: the pool columns through non-temporal loads (kernels.hpp: ld_off) -- pool sets several times the Infinity Cache
template <int E, bool DET = false, bool PLAIN = false, bool DMA = false, int NT = 0>
__global__ void __launch_bounds__(EVAL_THREADS, EVAL_WAVES_PER_SIMD)
iter_kernel(IterArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    constexpr int SLOT = STAGE_BYTES / 8, NSLOT = EVAL_THREADS / 64;
    // DMA: [NSLOT][SLOT] staged tiles, one slot per wave, FIRST in the carve (their LDS addresses travel through M0 and stay
    // below 64 KB that way); the accepted-point stash of the workgroups that store it borrows the last slots
    double *const stage0 = lds_raw;
    double *const lds = lds_raw + (DMA ? NSLOT * SLOT : 0);
    constexpr int MM = ITER_MM, P = ITER_P;
    static_assert(MM == 3, "DevState::rhow0..2 hold a window of 3");
    const int n = a.n, M = a.M;
    const int tid = threadIdx.x, lane = tid & 63, wave = uni((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
    // LDS carve of eval_kernel<false, .>; the update borrows the waves' exchange strips (16 KB, free until the tile loop)
    // and the psi tile (a stash for the trial gradient)
    const int tile = eval_tile_doubles(n, DET);          // (reproducible mode: the psi tile holds 3 n integer limbs)
    double *psi_s = lds;
    double *nu_s = lds + tile;                           // [n + 1]
    double *fpart = nu_s + n + 2;                        // [16]
    int *next_tile = reinterpret_cast<int *>(fpart + 16);
    double *strips = lds + eval_lds_doubles(n, false, DET);
    constexpr bool LNU = eval_has_lnu(false, DET) && !DMA;
    double *lnu_s = lds + eval_lnu_offset(n, false, DET); // [n] log-prices for the K-asset tiles (kernels.hpp: tilen<LNU>)
    double *xw = strips;                                 // [16][32] per-wave sums
    double *xm = xw + 16 * 32;                           // [16][2] maxima
    BlockRed red(xm + 32);                               // [2][12][16]
    double *ctl = xm + 32 + 2 * BlockRed::NRED * 16;     // [2 P + 2] what the scalar section hands to the other waves: alpha | gamma | step | sum |pg|
    int *ctli = reinterpret_cast<int *>(ctl + 2 * ITER_P + 2);       // [4] accept | new direction | status | pair accepted
    double *gst_s = psi_s;                               // [n] trial gradient, stashed between the two halves of the update
    double *sts_s = nu_s;                                // [n] trial point, likewise (the prices are written at the very end)
    double *glo_s = strips + 2 * 64 * (EVAL_THREADS / 64);   // [xvs] lower bounds | [xvs] upper bounds
    double *ghi_s = glo_s + a.xvs;
    double *psi_k = DMA ? stage0 + NSLOT * SLOT - 2 * a.xvs : ghi_s + a.xvs, *nu_k = psi_k + a.xvs;    // the workgroups that store the accepted point only
    double *fsafe = (DMA ? ghi_s : ghi_s + 2 * a.xvs) + a.xvs;        // [2][16] per-wave partials of the direction's safeguards (d.G | max |d|)
    const int first_late = (NSLOT * SLOT - 2 * a.xvs) / SLOT;    // DMA: the slots from here on hold that stash during the update

#ifdef CFMM_PHASE_TIMERS
    const long long ts_c16 = clock64(), ts_w16 = wall_clock64();     // (stored only by launches that go on to evaluate: the idle ones behind the end of a solve must not overwrite them)
#endif
    const int p = a.phase, pr = (p + 2) % 3, pz = (p + 1) % 3;
    const int stride = acc_stride(n), xvs = a.xvs;
    const double *Xr = a.xs + (size_t)pr * a.xs_set;
    double *Xw = a.xs + (size_t)p * a.xs_set;
    const double *Ar = a.acc3 + (size_t)pr * a.acc_set;
    // Every workgroup computes the same new state; WHO stores which piece of it is spread over the first workgroups, one
    // vector each (workgroup 0 alone storing all sixteen -- state set, history window, accepted point -- ran 1-1.6 us behind
    // the others, and a launch ends with its slowest workgroup).  Role r is taken by workgroup r % gridDim.
    enum { R_STATE = 0, R_S = 0, R_ST = 1, R_GS = 2, R_D = 3, R_NU = 4, R_SW = 5, R_YW = 5 + ITER_MM, R_PSI_ACC = 5 + 2 * ITER_MM, R_NU_ACC, R_DS };
    const bool wide = gridDim.x > R_DS;                  // (the usual case: no division on the way)
    const bool has_role = !wide || blockIdx.x <= R_DS;   // (all other workgroups pass every store site on ONE test)
    auto mine = [&](int role) { return wide ? blockIdx.x == (unsigned)role : blockIdx.x == (unsigned)role % gridDim.x; };
    const bool wr = mine(R_STATE);                       // the workgroup that stores the state record, the prices and the progress word
    const int r0 = tid * E;
    const int ld0 = (r0 < n) ? r0 : 0;                   // threads past the end load element 0 and are masked out
    bool tin[E];
#pragma unroll
    for (int e = 0; e < E; ++e) tin[e] = r0 + e < n;
    // Waves that own no variable take no part in the update's vector work (wave-uniform branches): they only meet the
    // others at the barriers (E = 2: 8 of 16 waves carry variables at 1000 tokens)
    const bool wave_active = wave * 64 * E < n;
    const int nwa = (n + 64 * E - 1) / (64 * E);         // waves with variables: the first nwa
    red.nw = nwa;
    // EVERY load of the update goes out here, before the solver state is waited for: all addresses follow from the rotation
    // phase (the history window is stored in window order with the state set; which of its pairs count -- st.hist -- is
    // needed only when they are used)
    GramIn<E> in;
    double s[E], s_t[E], Gs[E], nuj[E], Ds[E], glo[E], ghi[E], hj[E], cj[E], psi[E];
    int ct[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        s[e] = s_t[e] = Gs[e] = nuj[e] = Ds[e] = glo[e] = cj[e] = 0.0; psi[e] = 0.0; ghi[e] = __builtin_inf(); hj[e] = 0.0; ct[e] = 0;
        in.q0[e] = in.H0[e] = in.hq[e] = 0.0;
#pragma unroll
        for (int k = 0; k < P; ++k) { in.S[k][e] = 0.0; in.Y[k][e] = 0.0; }
    }
    if (wave_active) {
        ldE<E>(Xr, ld0, s); ldE<E>(Xr + xvs, ld0, s_t); ldE<E>(Xr + 2 * xvs, ld0, Gs); ldE<E>(Xr + 4 * xvs, ld0, nuj);
        ldE<E>(a.Ds, ld0, Ds); ldE<E>(a.glo, ld0, glo); ldE<E>(a.c, ld0, cj);
        if (!PLAIN) { ldE<E>(a.ghi, ld0, ghi); ldE<E>(a.h, ld0, hj); ldEi<E>(a.ctype, ld0, ct); }
        for (int sl = 0; sl < a.nread; ++sl) {
            double t1[E];
            ldE<E>(Ar + (size_t)sl * stride, ld0, t1);
#pragma unroll
            for (int e = 0; e < E; ++e) psi[e] += t1[e];
        }
#pragma unroll
        for (int k = 0; k < MM; ++k) {                   // stored pairs, newest first, at in.S[k + 1]
            ldE<E>(Xr + (size_t)(XS_HIST + k) * xvs, ld0, in.S[k + 1]);
            ldE<E>(Xr + (size_t)(XS_HIST + MM + k) * xvs, ld0, in.Y[k + 1]);
        }
    }
    if (DET) for (int j = n + tid; j < tile; j += blockDim.x) psi_s[j] = 0.0;      // (limbs behind the stash: entries [0, n) are recycled by their owners below)
    DevState st = a.st3[pr];
    if (st.status != 0) {                                // the solve has ended: every workgroup of every later launch leaves here;
        if (blockIdx.x == 0 && tid == 0) {
            a.st3[p] = st;                               // the final state is handed on, or the launch after next would read a set
            if (a.hring) __hip_atomic_store(a.hring + (a.launch & (ITER_HRING - 1)), iter_hring_word(st.evals, st.status, a.launch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;                                          // from before the end (status 0) and resume from stale state
    }
    st.evals = uni(st.evals); st.iters = uni(st.iters); st.first = uni(st.first); st.hist = uni(st.hist);
    st.nrej = uni(st.nrej); st.f = uni(st.f); st.t_step = uni(st.t_step);
    // (tuning builds: only launches that go on to evaluate leave stamps, so that all of them come from ONE launch -- the
    //  budget-limited solves of tools/microbench.py end on the evaluation count)
    long long *const tsb = (st.evals + 1 < a.max_evals) ? a.ev.ts : nullptr;
    (void)tsb;
    PHASE_STAMP(tsb, 17);

    // the set the NEXT launch flushes into was last read one launch ago: one workgroup clears it now
    if (blockIdx.x == gridDim.x - 1) {
        double *Z = a.acc3 + (size_t)pz * a.acc_set;
        const int len = a.ev.nslices * stride;
        for (int j = tid; j < len; j += blockDim.x) Z[j] = 0.0;
    }

    // ================= first half: everything that feeds the batched reduction.  Its inputs (state, bounds,
    // accumulators: ~20 registers per variable) die here; the second half takes the few it needs from an LDS stash ========
    bool act[E];
#pragma unroll
    for (int e = 0; e < E; ++e) act[e] = true;
    const bool keeps_acc = has_role && (mine(R_PSI_ACC) || mine(R_NU_ACC));
    if (wave_active) {
        double mx[2] = {0.0, 0.0};
        if (st.first) {                                  // first update of a solve: the diagonal metric rides along
            double dg[E];
#pragma unroll
            for (int e = 0; e < E; ++e) dg[e] = 0.0;
            for (int sl = 0; sl < a.nread; ++sl) {
                double t2[E];
                ldE<E>(Ar + (size_t)sl * stride + acc_diag(n), ld0, t2);
#pragma unroll
                for (int e = 0; e < E; ++e) dg[e] += t2[e];
            }
#pragma unroll
            for (int e = 0; e < E; ++e) Ds[e] = dg[e];
            if (has_role && r0 < n && mine(R_DS)) stE<E>(a.Ds, r0, n, Ds);      // (the first trial point is always accepted)
        }
#pragma unroll
        for (int k = 0; k < MM; ++k) {                   // (pairs beyond the window's fill, lanes beyond the end: zero)
            const bool have = k < st.hist;
#pragma unroll
            for (int e = 0; e < E; ++e) if (!have || !tin[e]) { in.S[k + 1][e] = 0.0; in.Y[k + 1][e] = 0.0; }
        }
        double fpools = 0.0;
        if (tid < a.nread) fpools = Ar[(size_t)tid * stride + acc_arb(n)];
#pragma unroll
        for (int i = 0; i < 8; ++i) in.g1[i] = 0.0;
        in.g1[0] = fpools;
        double Gs_t[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            Gs_t[e] = 0.0;
            if (tin[e]) {
                const double rj = PLAIN ? psi[e] : psi[e] + hj[e];
                Gs_t[e] = nuj[e] * rj;
                if (!PLAIN) in.g1[0] += (nuj[e] - cj[e]) * hj[e];
                in.g1[1] += (nuj[e] - cj[e]) * rj;
                mx[0] = fmax(mx[0], (PLAIN || ct[e] == 0) ? fmax(-rj, 0.0) : (ct[e] == 1 ? fabs(rj) : 0.0));
                mx[1] = fmax(mx[1], PLAIN ? fabs(psi[e]) : fmax(fabs(psi[e]), fabs(hj[e])));
                const double sv = s_t[e] - s[e], yv = Gs_t[e] - Gs[e];
                if (!st.first) {
                    in.S[0][e] = sv; in.Y[0][e] = yv;
                    in.g1[2] += Gs[e] * sv; in.g1[3] += Gs_t[e] * sv;
                    in.g1[4] += sv * yv; in.g1[5] += sv * sv; in.g1[6] += yv * yv;
                }
                const double G = Gs_t[e], sr = s_t[e];
                double v = G;
                if (!PLAIN && glo[e] == ghi[e]) v = 0.0;
                else if (sr <= glo[e] + 1e-14) v = fmin(G, 0.0);
                else if (!PLAIN && sr >= ghi[e] - 1e-14) v = fmax(G, 0.0);
                in.g1[7] += fabs(v);
                act[e] = PLAIN ? (sr <= glo[e] + 1e-14 && G > 0.0) : is_active(sr, glo[e], ghi[e], G);
                in.q0[e] = act[e] ? 0.0 : G;
                const double H = Ds[e] + fmax(G, 0.0);
                in.H0[e] = H > 0.0 ? rcp_nr(H) : 0.0;
                in.hq[e] = in.H0[e] * in.q0[e];
                gst_s[r0 + e] = Gs_t[e];
                sts_s[r0 + e] = s_t[e]; glo_s[r0 + e] = glo[e];
                if (!PLAIN) ghi_s[r0 + e] = ghi[e];
                if (keeps_acc) { psi_k[r0 + e] = psi[e]; nu_k[r0 + e] = nuj[e]; }
            }
        }
        PHASE_STAMP(tsb, 18);
        const double qa = gram_reduce32<E>(in, lane);
        PHASE_STAMP(tsb, 19);
        mx[0] = wave_allmax(mx[0]); mx[1] = wave_allmax(mx[1]);
        if (lane < 32) xw[wave * 32 + lane] = qa;
        if (lane == 0) { xm[wave * 2] = mx[0]; xm[wave * 2 + 1] = mx[1]; }
    }
    // (the last wave -- one without variables up to 960 tokens -- lays out the workgroup's tile ranges meanwhile: the tile
    //  loop starts without a table-building prologue and its two barriers)
    if (wave == nw - 1) build_tile_table(a.ev, next_tile, lane);
    __syncthreads();
    st.evals += 1;
    // DMA, workgroups with waves that carry no variables (up to 960 tokens: half of them at 1000): THOSE waves request every
    // wave's first tile, now -- they idle until the tile phase anyway, while on the waves that run the update's chain the issue
    // of four 1 KB DMA pieces each (and the wait for the vector-memory queue they fill) was measured to cost 0.5-1 us of
    // chain.  Slots are dealt round-robin; the two workgroups that stash the accepted point in the last slots leave those to
    // their owners (behind the update).  Should the scalar section end the solve, the pieces are waited for before the exit.
    const bool keeps_acc0 = has_role && (mine(R_PSI_ACC) || mine(R_NU_ACC));
    const bool helper_mode = DMA && nwa < nw;
    if constexpr (DMA) {
        if (helper_mode && !wave_active && st.evals < a.max_evals) {
            __builtin_amdgcn_s_waitcnt(0x0F70);           // (nothing of the compiler's own is pending: see below)
            const int nhelp = nw - nwa, h = wave - nwa;
            for (int sl = h; sl < nw; sl += nhelp) {     // slots h, h + nhelp, ...: every slot exactly once over the helpers
                if (keeps_acc0 && sl >= first_late) continue;
                tiles_dma_first(a.ev, next_tile, lds_addr(stage0 + SLOT * sl), lane, sl);
            }
        }
    }

    // ================= the scalar section: accept test, curvature pair, stopping rule, two-loop recursion.  ONE wave runs
    // it; the others sleep at the next barrier and pick the results up from LDS =============================================
    bool accept = false, new_dir = false, pair_ok = false;
    double al[P], ga[P];
#pragma unroll
    for (int k = 0; k < P; ++k) { al[k] = 0.0; ga[k] = 0.0; }
    double gp_sq = 0.0;
    if (wave == 0) {
        // lane l < 32 ends with total l; a total is fetched with two v_readlane (into SGPRs: no LDS broadcast round trips on
        // the dependent chain)
        double tl = 0.0;
        if (lane < 32) for (int w = 0; w < nwa; ++w) tl += xw[w * 32 + lane];
        const int tl_lo = __double2loint(tl), tl_hi = __double2hiint(tl);
        auto T = [&](int i) { return __hiloint2double(__builtin_amdgcn_readlane(tl_hi, i), __builtin_amdgcn_readlane(tl_lo, i)); };
        PHASE_STAMP(tsb, 20);
        double viol = 0.0, scale = 0.0;
        for (int w = 0; w < nwa; ++w) { viol = fmax(viol, xm[w * 2]); scale = fmax(scale, xm[w * 2 + 1]); }
        double rho[P];
        rho[0] = 0.0;
#pragma unroll
        for (int k = 0; k < MM; ++k) rho[k + 1] = (k < st.hist) ? uni(k == 0 ? st.rhow0 : (k == 1 ? st.rhow1 : st.rhow2)) : 0.0;
        const double f_t = T(0), gapv = T(1);

        // ---- accept test ----------------------------------------------------------------------------------------
        accept = st.first != 0;
        if (!st.first) accept = lbfgs::accept(f_t, st.f, a.armijo, T(2), T(3));
        if (!accept) {
            lbfgs::reject(st);
        } else {
            // ---- curvature pair, move the accepted point -----------------------------------------------------
            const int old_hist0 = st.hist;
            if (!st.first) {
                const double sy = T(4);
                if (lbfgs::pair_ok(sy, T(5), T(6))) {
                    pair_ok = true;
                    rho[0] = rcp_nr(sy);
                    if (st.hist < M) st.hist += 1;
                }
                st.iters += 1;
            }
            gp_sq = T(7);                              // (sum |projected gradient|: positive iff some free variable has a gradient)
            lbfgs::certify(st, f_t, gapv, viol, scale, gp_sq);
            const bool was_first = st.first != 0;
            st.first = 0;
            if (lbfgs::converged(st, a.pg_rule, a.tol_gap, a.tol_infeas)) {
                st.status = 1;
            } else {
                // ---- the two-loop recursion on scalars -------------------------------------------------------
                // which pairs are in the window: the new one if it passed, then the newest stored ones
                new_dir = true;
                const int keep_old = lbfgs::keep_old(was_first, pair_ok, old_hist0, M);
#pragma unroll
                for (int k = 1; k < P; ++k) if (k - 1 >= keep_old) rho[k] = 0.0;
                lbfgs::gram_two_loop<P>(rho, [&](int k) { return T(GI_U + k); }, [&](int k) { return T(GI_V + k); },
                                        [&](int k, int j) { return T(GI_SY + k * (k - 1) / 2 + j); },
                                        [&](int k, int j) { return T(GI_YHY + k * (k + 1) / 2 + j); }, al, ga);
            }
            if (pair_ok) { st.rhow2 = st.rhow1; st.rhow1 = st.rhow0; st.rhow0 = rho[0]; }      // the window moves on
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < P; ++k) { ctl[k] = al[k]; ctl[P + k] = ga[k]; }
            ctl[2 * P] = st.t_step; ctl[2 * P + 1] = gp_sq;
            ctli[0] = accept ? 1 : 0; ctli[1] = new_dir ? 1 : 0; ctli[2] = st.status; ctli[3] = pair_ok ? 1 : 0;
        }
    }
    PHASE_STAMP(tsb, 21);
    if constexpr (DMA) lds_barrier(); else               // (LDS only: the helpers' DMA pieces may be in flight)
    __syncthreads();
    // every wave (wave 0 too: its vector copies of alpha / gamma die at the barrier) takes the uniform results into SGPRs
    accept = uni(ctli[0]) != 0; new_dir = uni(ctli[1]) != 0; pair_ok = uni(ctli[3]) != 0;
    gp_sq = uni(ctl[2 * P + 1]);
    if (wave != 0) { st.status = uni(ctli[2]); st.t_step = uni(ctl[2 * P]); }      // (wave 0 holds the complete record, which it stores at the end)
    // the columns of this wave's first tile: requested now (unless the solve has just ended or runs out of budget here: such
    // a launch does not evaluate), they land while the direction and the trial point are formed
    const bool dma_late = keeps_acc && wave >= first_late;
    if constexpr (DMA) {
        // (every load of the update has long been consumed -- but the compiler cannot see that the waves that issued them are
        //  the waves that consumed them (two `if (wave_active)` regions with barriers in between), so it re-waits for them at
        //  their next use, and at run time such a vmcnt(N) waits for the DMA pieces issued here.  This wait costs nothing
        //  and tells it that nothing of its own is pending any more.)
        __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0), nothing else
        if (!helper_mode && st.status == 0 && st.evals < a.max_evals && !dma_late) tiles_dma_first(a.ev, next_tile, lds_addr(stage0 + SLOT * wave), lane, wave);
    }
    if (new_dir && wave_active) {
#pragma unroll
        for (int k = 0; k < P; ++k) { al[k] = uni(ctl[k]); ga[k] = uni(ctl[P + k]); }
    }
    // the history window of the next launch, in window order, with the state set this launch writes
    if (has_role && wave_active && r0 < n) {
        const int sh = (accept && pair_ok) ? 0 : 1;      // (the new pair enters at the front, or the window stays)
#pragma unroll
        for (int k = 0; k < MM; ++k) {
            if (mine(R_SW + k)) { if (sh) stX<E>(Xw + (size_t)(XS_HIST + k) * xvs, r0, in.S[k + 1]); else stX<E>(Xw + (size_t)(XS_HIST + k) * xvs, r0, in.S[k]); }
            if (mine(R_YW + k)) { if (sh) stX<E>(Xw + (size_t)(XS_HIST + MM + k) * xvs, r0, in.Y[k + 1]); else stX<E>(Xw + (size_t)(XS_HIST + MM + k) * xvs, r0, in.Y[k]); }
        }
    }

    // ================= second half: the direction, the next trial point =================================================
    // its inputs come back from the LDS stash (keeping them in registers across the reduction spills; global reloads cost
    // an L2 round trip on the chain): the accepted point (s moves to the trial point, or stays), its gradient (the stashed
    // trial gradient, or the old one), the bounds; a rejected trial point reloads the old point, gradient and direction
    // from the state set.  Every thread reads and then recycles its OWN stash entries: no barrier in between.
    double d[E], v[E], nn[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { s[e] = Gs[e] = d[e] = glo[e] = 0.0; ghi[e] = __builtin_inf(); v[e] = nn[e] = 0.0; }
    if (wave_active) {
#pragma unroll
        for (int e = 0; e < E; ++e) if (tin[e]) { glo[e] = glo_s[r0 + e]; if (!PLAIN) ghi[e] = ghi_s[r0 + e]; }
        if (accept) {                                    // the trial point and its gradient: from the LDS stash
#pragma unroll
            for (int e = 0; e < E; ++e) if (tin[e]) { s[e] = sts_s[r0 + e]; Gs[e] = gst_s[r0 + e]; }
        } else {
            ldE<E>(Xr, ld0, s); ldE<E>(Xr + 2 * xvs, ld0, Gs); ldE<E>(Xr + 3 * xvs, ld0, d);
            // (waited for HERE: left pending, the compiler places its vmcnt(0) at their first use on the common path behind the
            //  join, where at run time it waits for the first tiles' LDS-DMA instead -- the accepted branch stood still for ~1 us)
            if constexpr (DMA) __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0), nothing else
        }
#pragma unroll
        for (int e = 0; e < E; ++e) if (tin[e]) psi_s[r0 + e] = 0.0;       // (the stash entry becomes this thread's share of the zeroed psi tile)
        if (keeps_acc && accept && r0 < n) {             // the accepted prices and their net trade, for the read-back
            double pk[E], nk[E];                         // (from the stash: a reload through L2 would sit on the chain)
#pragma unroll
            for (int e = 0; e < E; ++e) { pk[e] = tin[e] ? psi_k[r0 + e] : 0.0; nk[e] = tin[e] ? nu_k[r0 + e] : 0.0; }
            if (mine(R_PSI_ACC)) { stE<E>(a.psi_acc, r0, n, pk); if (a.h_psi_acc) stE<E>(a.h_psi_acc, r0, n, pk); }
            if (mine(R_NU_ACC)) { stE<E>(a.nu_acc, r0, n, nk); if (a.h_nu_acc) stE<E>(a.h_nu_acc, r0, n, nk); }
        }
    }

    auto trial = [&](double step) {                      // next trial point of this thread's variables
#pragma unroll
        for (int e = 0; e < E; ++e) {
            v[e] = fmax(s[e] + step * d[e], glo[e]);
            if (!PLAIN) v[e] = fmin(v[e], ghi[e]);
            nn[e] = tin[e] ? exp(v[e]) : 0.0;
        }
    };
    double F[2] = {0.0, 0.0};                  // d.G | max |d|
    bool F_pending = false;
    // the direction's safeguards on the two totals; true when the trial point had to be taken again (the speculated full step does not stand)
    auto safeguard = [&](double (&F)[2]) {
        bool redo = false;
        if (!(F[0] < 0.0) && gp_sq > 0.0) {       // not a descent direction: restart from the metric
            st.hist = 0;
            double m1[1] = {0.0};
#pragma unroll
            for (int e = 0; e < E; ++e) {
                d[e] = (!tin[e] || act[e]) ? 0.0 : -Gs[e] * in.H0[e];
                m1[0] = fmax(m1[0], fabs(d[e]));
            }
            red.put<0, 1>(m1, wave_active);
            red.template get<0, 1, DMA>(m1);
            F[1] = m1[0];
            redo = true;
        }
        st.t_step = lbfgs::step_cap(F[1], a.max_step);
        const bool again = redo || st.t_step != 1.0;
        if (again && wave_active) trial(st.t_step);
        return again;
    };
    if (new_dir) {
        if (wave_active) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                double qm = in.q0[e], rs = 0.0;
#pragma unroll
                for (int k = 0; k < P; ++k) { qm -= al[k] * in.Y[k][e]; rs += ga[k] * in.S[k][e]; }
                d[e] = (tin[e] && !act[e]) ? -(in.H0[e] * qm + rs) : 0.0;
                F[0] += d[e] * Gs[e]; F[1] = fmax(F[1], fabs(d[e]));
            }
        }
#ifndef CFMM_EAGER_SAFEGUARD
        F_pending = st.evals < a.max_evals;    // (a launch that runs out of budget here stores its state and leaves: the plain order, once per solve at most)
#endif
        if (F_pending) {
            // the totals are taken BEHIND the barrier in front of the tiles (below); the per-wave partials wait in a strip of LDS that
            // the tile phase does not touch (BlockRed's scratch lies in the waves' exchange strips: a fast wave's first tile would overwrite it)
            if (wave_active) {
                F[0] = wave_allsum(F[0]); F[1] = wave_allmax(F[1]);
                if (lane == 0) { fsafe[wave] = F[0]; fsafe[16 + wave] = F[1]; }
                trial(1.0);                    // (speculation: the full step is the common case)
            }
        } else {
            red.put<1, 1>(F, wave_active);
            if (wave_active) trial(1.0);
            red.template get<1, 1, DMA>(F);
            (void)safeguard(F);
        }
    } else if (st.status == 0 && wave_active) trial(st.t_step);

    // ---- workgroup 0 stores the state; the trial prices go into this workgroup's LDS table ---------------------------------
    // (a solve that has ended above took no trial point: its price entries stay zero)
    if (st.status == 0 && st.evals >= a.max_evals) st.status = 3;
    auto store_state = [&]() {
        if (has_role && wave_active && r0 < n) {
            if (mine(R_S)) stX<E>(Xw, r0, s);
            if (mine(R_ST)) stX<E>(Xw + xvs, r0, v);
            if (mine(R_GS)) stX<E>(Xw + 2 * xvs, r0, Gs);
            if (mine(R_D)) stX<E>(Xw + 3 * xvs, r0, d);
            if (mine(R_NU)) stX<E>(Xw + 4 * xvs, r0, nn);
            if (wr) stE<E>(a.nu, r0, n, nn);
        }
        if (wr) {
            if (tid == 0) {
                a.st3[p] = st; a.nu[n] = st.status != 0 ? 1.0 : 0.0;
                if (st.status != 0 && a.h_final) *a.h_final = st;
                // progress word for the host (system-scope store into pinned host memory: no copy, no API call on the host side)
                if (a.hstat) __hip_atomic_store(a.hstat, (unsigned long long)(unsigned)st.evals | ((unsigned long long)(unsigned)st.status << 32),
                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (a.hring) __hip_atomic_store(a.hring + (a.launch & (ITER_HRING - 1)), iter_hring_word(st.evals, st.status, a.launch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    if (st.status != 0) {                                // ended (converged / stalled / out of budget): nothing to evaluate
        store_state();
        if constexpr (DMA) dma_wait();                   // (a helper's pieces must have landed before its LDS goes back)
        return;
    }
    PHASE_STAMP(a.ev.ts, 22);
    auto publish = [&]() {                               // (over this thread's own stash entry; v = log nn: the trial point itself)
#pragma unroll
        for (int e = 0; e < E; ++e) if (tin[e]) { nu_s[r0 + e] = nn[e]; if constexpr (LNU) lnu_s[r0 + e] = v[e]; }
    };
    publish();
    if constexpr (DMA) {
        if (helper_mode && !wave_active) dma_wait();     // (what a helper requested for the other waves is in LDS before they pass the barrier)
        lds_barrier();                                   // (LDS only: a fence would wait for the first tiles' DMA -- and for nothing else that matters here)
    } else
    __syncthreads();                                     // (the scratch in the exchange strips is free from here on; the tile table
    PHASE_STAMP(a.ev.ts, 23);                            //  and the ticket counter have been ready since the first barrier)
    // Round 6: the direction's two safeguards (is it a descent direction?  does the full step exceed the cap in log-price?) used to
    // cost a barrier of their own between the direction and the trial point.  Both almost never fire, and the trial point at the full
    // step was already computed speculatively under that barrier -- so the speculation now runs all the way: the per-wave partials
    // wait in LDS (BlockRed::put), the trial prices are published, and the totals are formed BEHIND the barrier the tile phase needs
    // anyway.  Only a launch whose safeguard fires republishes its prices and pays the second barrier.  The state goes out after the
    // check (it carries the step): every workgroup sums the same partials in the same order, so all of them decide alike.
    if (F_pending) {
        F[0] = 0.0; F[1] = 0.0;
        for (int w = 0; w < nwa; ++w) { F[0] += fsafe[w]; F[1] = fmax(F[1], fsafe[16 + w]); }
        if (safeguard(F)) {
            publish();
            if constexpr (DMA) lds_barrier(); else __syncthreads();
        }
    }
    store_state();
#ifdef CFMM_PHASE_TIMERS
    if (a.ev.ts && tid == 0) {
        if (blockIdx.x == 0) { a.ev.ts[2 * 16] = ts_c16; a.ev.ts[2 * 16 + 1] = ts_w16; }
        if (blockIdx.x < 256) { long long *tb = a.ev.ts + 64 + 8 * 4096; tb[2 * blockIdx.x] = ts_w16; tb[2 * (blockIdx.x + 256)] = wall_clock64(); }     // block start | update done
    }
#endif
    double2 *xs = reinterpret_cast<double2 *>(strips) + 64 * wave;
    if constexpr (DMA)
        eval_tiles_and_flush<false, false, DET, false, true, true>(a.ev, a.acc3 + (size_t)p * a.acc_set, nu_s, psi_s, nullptr, fpart, next_tile, xs,
                                                                   BatchCtl{1u, 0, 0}, nullptr, stage0 + SLOT * wave, !dma_late);
    else
    eval_tiles_and_flush<false, false, DET, false, true, false, NT, LNU>(a.ev, a.acc3 + (size_t)p * a.acc_set, nu_s, psi_s, nullptr, fpart, next_tile, xs,
                                                                         BatchCtl{1u, 0, 0}, nullptr, nullptr, false, lnu_s);
#ifdef CFMM_PHASE_TIMERS
    __syncthreads();
    if (a.ev.ts && tid == 0 && blockIdx.x < 256) a.ev.ts[64 + 8 * 4096 + 2 * blockIdx.x + 1] = wall_clock64();    // block end
#endif
}

}  // namespace cfmm
