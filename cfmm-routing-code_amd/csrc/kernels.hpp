// HIP kernels of the dual-decomposition hot path (gfx950 / MI355X, wave64, fp64, no MFMA).
//
//   eval_kernel<WITH_D>      ONE launch = one dual evaluation of every pool bucket: stream the SoA
//        columns (coalesced, once), gather nu from an LDS copy, solve the pool (pool_math.hpp; K-asset
//        pools leg-per-lane), scatter-add A_i(L_i - D_i) into an LDS tile of psi (ds_add_f64), flush
//        the tile to one of `nslices` global accumulators (global_atomic_add_f64).
//                                                                        reference: arbitrage.py:54
//   update_gram_kernel / update_reg_kernel / update_kernel     consume the accumulators (after the
//        all-reduce when pool-sharded) and take one step of the projected L-BFGS iteration on
//        log-prices -- the on-device "nu update": Gram form (<= 1024 tokens, memory <= 4),
//        register-resident sequential form (<= 2048 tokens), generic form.
//                                                                        reference: arbitrage.py:82
//   trades2_kernel / tradesn_kernel       materialise Delta_i, Lambda_i at the accepted prices
//        (once per solve).                                              reference: two-asset.py:94,98
//   fold_kernel, start_kernel             slice fold before the RCCL all-reduce; start of a solve.
// The second-order outer iteration lives in smooth.hpp (barrier-smoothed evaluation, Hessian) and chol.hpp (dense solve).
#pragma once
#include "pool_math.hpp"
#include "phi2.hpp"
#include "lbfgs_rules.hpp"

namespace cfmm {

// phase timers for kernel tuning (variant builds with -DCFMM_PHASE_TIMERS): thread 0 of block 0
// stamps {shader cycles, 100 MHz wall clock} pairs into a debug buffer read by cfmm_debug_timers()
#ifdef CFMM_PHASE_TIMERS
#define PHASE_STAMP(buf, i) do { if ((buf) && blockIdx.x == 0 && threadIdx.x == 0) { (buf)[2 * (i)] = clock64(); (buf)[2 * (i) + 1] = wall_clock64(); } } while (0)
#else
#define PHASE_STAMP(buf, i) do { } while (0)
#endif

#ifndef EVAL_THREADS_DEF
#define EVAL_THREADS_DEF 1024
#endif
constexpr int EVAL_THREADS = EVAL_THREADS_DEF;     // fused evaluation kernel: waves per workgroup x 64
#ifndef EVAL_WAVES_PER_SIMD
#define EVAL_WAVES_PER_SIMD 4             // min waves per SIMD the register allocator must leave room for (<= 128 VGPRs)
#endif
constexpr int UPD_THREADS = 1024;
constexpr int MAX_MEMORY = 8;         // L-BFGS pairs kept (register-resident in update_kernel)

// the fused evaluation kernel walks "wave-tiles": WT_LIGHT / WT_HEAVY consecutive pools of a two-asset bucket
// (two / one per lane) or 64 / K consecutive pools of a K-asset bucket (one LEG per lane); buckets are laid out in
// the tile space heaviest first, so the light tiles fill the tail of the launch
#ifndef WT_LIGHT_DEF
#define WT_LIGHT_DEF 128
#endif
constexpr int WT_LIGHT = WT_LIGHT_DEF;  // cp2, sum2: two pools per lane
constexpr int WT_HEAVY = 64;          // w2, curve2: one pool per lane
constexpr int WT_WIDE = 256;          // cp2 of a very large bucket (EvalArgs::wide): four pools per lane
// The staged tile walk (LDS-DMA one tile ahead, below) is compiled in with -DCFMM_STAGED_WALK=1 (`make variant TAG=staged
// DEFS=-DCFMM_STAGED_WALK=1`) and then taken unless CFMM_TILE_DMA=0.  Round 4 built it, validated it (the whole -m gpu suite
// passes through it) and measured it SLOWER than the direct walk on every BASELINE config (tried and rejected):
// it is kept as an A/B variant, not as the default.
#ifndef CFMM_STAGED_WALK
#define CFMM_STAGED_WALK 0
#endif
// pools of a K-asset wave-tile: 64 / K with one leg per lane.  Staged-walk builds take K = 3 as 20 (not 21) and K = 7 as 8
// (not 9) pools, so that every tile starts on a multiple of 4 legs and of 2 pools: its idx / R / w / fee segments are then
// 16-byte aligned, which is what the 16-byte LDS-DMA pieces (tile_dma_issue) move
__host__ __device__ constexpr int ktile_pools(int k) { return CFMM_STAGED_WALK ? (k == 3 ? 20 : (k == 7 ? 8 : 64 / k)) : 64 / k; }
__host__ __device__ constexpr int wave_tile_pools(int code)     // code: CFMM_POOL_* kind, or -k
{
    return code < 0 ? ktile_pools(-code)               // k-asset geo-mean: one LEG per lane
                    : ((code == 0 || code == 2) ? WT_LIGHT : WT_HEAVY);
}
constexpr int N_BUCKETS = 11;         // gn8 gn7 gn6 gn5 gn4 gn3 curve2 pow2 w2 cp2 sum2 (processing order)
constexpr int N_KINDS2 = 5;           // CFMM_POOL_KINDS2: cp2, w2, sum2, curve2, and the generic bucket's first tenant pow2 (phi2.hpp)
// "heavy" two-asset kinds: evaluated by an iteration per pool (stableswap's own Newton loop; every kind that goes through
// pool_generic2).  They live in the tile space of eval_kernel<., STABLE = true>, a launch of its own: their loops need
// ~20 more VGPRs than the closed forms, which the main instantiation cannot spare.
__host__ __device__ constexpr bool heavy_kind(int kind) { return kind == 3 || kind == 4; }

struct DevState {
    int status, evals, iters, first, hist, head, nrej, pad;
    double f, t_step, gap, infeas, primal, pg;
    // iter_kernel only: 1 / s'y of its history window, newest first (iterate.hpp).  Three scalars, not an array: the kernels that only
    // CARRY the window (`DevState st = load_state(a.st); ... *a.st = st`) copied an array member through memory the compiler would not promote
    // to registers -- 32 B of scratch in every stand-alone update kernel and in solve_tiny_kernel
    double rhow0, rhow1, rhow2;
};
// The record as the stand-alone update kernels and the one-wave solves use it: field by field, WITHOUT the window scalars (which they
// neither read nor change).  A whole-struct copy in and out (`DevState st = load_state(a.st); ... store_state(a.st, st);`) carried that untouched 24-byte
// tail through two overlapping 16-byte pieces in scratch memory -- 32 B of private segment in eleven kernels.
__device__ __forceinline__ DevState load_state(const DevState *p)
{
    DevState s;
    s.status = p->status; s.evals = p->evals; s.iters = p->iters; s.first = p->first; s.hist = p->hist; s.head = p->head; s.nrej = p->nrej; s.pad = 0;
    s.f = p->f; s.t_step = p->t_step; s.gap = p->gap; s.infeas = p->infeas; s.primal = p->primal; s.pg = p->pg;
    s.rhow0 = s.rhow1 = s.rhow2 = 0.0;
    return s;
}
__device__ __forceinline__ void store_state(DevState *p, const DevState &s)
{
    p->status = s.status; p->evals = s.evals; p->iters = s.iters; p->first = s.first; p->hist = s.hist; p->head = s.head; p->nrej = s.nrej;
    p->f = s.f; p->t_step = s.t_step; p->gap = s.gap; p->infeas = s.infeas; p->primal = s.primal; p->pg = s.pg;
}

struct Bucket2 {
    long long m;
    const double *Ra, *Rb, *fee, *param;
    const int *ia, *ib, *flags;
    const int *perm;                  // position -> the caller's pool index (reorder.hpp); null = the caller's order
    // the COMPACT MIRROR of the ids and the fee (round 4; built on the device behind the upload for buckets of >= 1e6 pools,
    // compact_build_kernel): token ids as one 32-bit word (ia | ib << 16: fewer than 65536 tokens fit the LDS tiles anyway) and the
    // fee as a one-byte index into a table of the bucket's distinct fees (<= 256 of them, else no mirror) -- 21 bytes per
    // constant-product pool instead of 32, bit for bit the same numbers.  The evaluation tiles take it when present; a pool set
    // of 160 MB .. 1.28 GB is evaluated at ~6.5 TB/s of bytes MOVED whatever its size, so the bytes are the time.
    const unsigned *cid;
    const unsigned char *cfee;
    const double *ctab;
};

struct BucketN {
    long long m;
    const int *idx;
    const double *R, *w, *fee;
    const double *lfee;               // log(fee), computed once at upload (+8 B/pool instead of one log per wave-tile)
    const double *lrw;                // log(R / w) per leg, computed once at upload: with the workgroup's table of log-prices a leg's
                                      // a = log(R p / w) is ONE add where it was a 36-instruction log per leg and evaluation (round 4)
    const int *perm;                  // position -> the caller's pool index (reorder.hpp); null = the caller's order
};

// builds a bucket's compact mirror; tab: 256 slots of fee BITS, zero = empty (a fee is > 0), open addressing
__global__ void __launch_bounds__(256) compact_build_kernel(Bucket2 b, unsigned *cid, unsigned char *cfee, unsigned long long *tab, int *overflow)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < b.m; i += stride) {
        cid[i] = (unsigned)b.ia[i] | ((unsigned)b.ib[i] << 16);
        const unsigned long long bits = (unsigned long long)__double_as_longlong(b.fee[i]);
        unsigned h = (unsigned)((bits * 0x9E3779B97F4A7C15ull) >> 56);
        int slot = -1;
        for (int probe = 0; probe < 256; ++probe, h = (h + 1) & 255u) {
            unsigned long long cur = tab[h];
            if (cur == 0ull) cur = atomicCAS(&tab[h], 0ull, bits), cur = cur == 0ull ? bits : cur;
            if (cur == bits) { slot = (int)h; break; }
        }
        if (slot < 0) { *overflow = 1; slot = 0; }
        cfee[i] = (unsigned char)slot;
    }
}

// the derived column of a K-asset bucket: lrw = log(R / w) per leg (BucketN::lrw), filled behind the upload's copies
__global__ void __launch_bounds__(256) lrw_fill_kernel(const double *__restrict__ R, const double *__restrict__ w, double *__restrict__ lrw, long long legs)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < legs; i += stride) lrw[i] = log(R[i] / w[i]);
}

struct EvalArgs {
    Bucket2 b2[N_KINDS2];             // indexed by CFMM_POOL_* kind
    BucketN bn[6];                    // bn[k - 3], k = 3..8
    int tile_end[N_BUCKETS];          // cumulative wave-tile counts in processing order
    int ntiles, n, nslices;
    int rev;                          // 1: this launch walks every workgroup's tile range BACKWARDS (see eval_tiles_and_flush)
    int wide;                         // 1: the constant-product bucket is walked in tiles of WT_WIDE pools (4 per lane) instead of WT_LIGHT:
                                      //    twice the bytes in flight per wave for a bucket whose evaluation is bound by memory latency (>= 8e6 pools)
    const double *nu;                 // [n + 1]: prices, then the stop flag (non-zero = solve has ended)
    double *acc;
    long long *ts;                    // phase timers (tuning builds only)
    unsigned long long *acc_l;        // reproducible mode: [2][3][n] fixed-point limbs of psi | diag (see Scatter<true>)
    double det_scale, det_scale_d;    // reproducible mode: powers of two that scale psi / diag contributions to integers
};

// LDS carve of eval_kernel (doubles), 16-byte aligned pieces; then 64 double2 per wave
__host__ __device__ inline int eval_tile_doubles(int n, bool det) { return (det ? 3 : 1) * n; }     // one scatter tile (psi or diag)
// (+ ticket + the tile-range table: 2 x N_BUCKETS ints; + the table of log-prices for the K-asset tiles of the evaluations
//  that do not build the metric -- every launch of a solve but its first -- outside the reproducible mode)
__host__ __device__ constexpr bool eval_has_lnu(bool with_d, bool det) { return !with_d && !det; }
__host__ __device__ inline int eval_lnu_offset(int n, bool with_d, bool det = false)
{
    return (((with_d ? 2 : 1) * eval_tile_doubles(n, det) + n + 2 + 16 + 2 + N_BUCKETS) + 1) & ~1;
}
__host__ __device__ inline int eval_lds_doubles(int n, bool with_d, bool det = false)
{
    return eval_lnu_offset(n, with_d, det) + (eval_has_lnu(with_d, det) ? ((n + 1) & ~1) : 0);
}

// accumulator slice layout (np = n rounded up to even, so that every piece is 16-byte aligned):
//   [0,n) psi | [np] sum arb | [np+8, np+8+n) diag
__host__ __device__ inline int acc_arb(int n) { return (n + 1) & ~1; }
__host__ __device__ inline int acc_diag(int n) { return acc_arb(n) + 8; }
__host__ __device__ inline int acc_stride(int n) { return acc_diag(n) + acc_arb(n); }
// row stride of the L-BFGS history (rows 16-byte aligned)
__host__ __device__ inline int hist_stride(int n) { return (n + 1) & ~1; }

// wave64 butterflies on the VALU cross-lane paths (DPP within a row of 16 lanes, then gfx950's
// v_permlane16_swap / v_permlane32_swap across rows): every lane ends up with the result, no LDS
// traffic (the __shfl_xor forms compile to ds_bpermute_b32 pairs, ~6x the latency)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// the two halves of an xor-16 / xor-32 exchange: x = own-or-partner, y = partner-or-own
__device__ __forceinline__ void swap16_f64(double v, double &x, double &y)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    x = __hiloint2double(b[0], a[0]); y = __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ void swap32_f64(double v, double &x, double &y)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    x = __hiloint2double(b[0], a[0]); y = __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double wave_allsum(double v)
{
    double x, y;
    v += dpp_f64<0xB1>(v);      // quad_perm [1,0,3,2]   lane ^ 1
    v += dpp_f64<0x4E>(v);      // quad_perm [2,3,0,1]   lane ^ 2
    v += dpp_f64<0x141>(v);     // row_half_mirror       lane ^ 4 (quads are uniform by now)
    v += dpp_f64<0x140>(v);     // row_mirror            lane ^ 8
    swap16_f64(v, x, y); v = x + y;
    swap32_f64(v, x, y); v = x + y;
    return v;
}
__device__ __forceinline__ double wave_allmax(double v)
{
    double x, y;
    v = fmax(v, dpp_f64<0xB1>(v));
    v = fmax(v, dpp_f64<0x4E>(v));
    v = fmax(v, dpp_f64<0x141>(v));
    v = fmax(v, dpp_f64<0x140>(v));
    swap16_f64(v, x, y); v = fmax(x, y);
    swap32_f64(v, x, y); v = fmax(x, y);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) { return wave_allsum(v); }
__device__ __forceinline__ double wave_max(double v) { return wave_allmax(v); }

// ------------------------------------------------------------------------------------------
// Where a pool's A_i (Lambda_i - Delta_i) lands in the workgroup's LDS tile of psi (arbitrage.py:54).
//   Scatter<false>: ds_add_f64 -- fast, but fp64 addition is not associative: the sum depends on the order in which lanes,
//     waves and workgroups arrive, so psi differs in its last bits from run to run (and with it the path of the solve).
//   Scatter<true> (reproducible mode, cfmm_set_deterministic): every contribution is converted EXACTLY to a 96-bit
//     fixed-point integer  y * 2^F = a2 2^64 + a1 2^32 + a0  and its three limbs are added to three 64-bit integer
//     accumulators (ds_add_u64; 32 bits of carry head-room each: no carry propagation while adding).  Integer addition is
//     associative and commutative, so the result is bit-identical whatever the order -- lanes, waves, ticket scheduling,
//     workgroups, accumulator flushes, pool shards on other GPUs (the limbs are all-reduced as integers) -- and the one
//     conversion back to fp64 happens in a fixed order (det_fold_kernel).  F is chosen per problem from the largest reserve.
// ------------------------------------------------------------------------------------------
// element `i` of a column whose base is wave-uniform: byte offset formed in 32 bits (columns stay below 4 GB)
// NT: non-temporal.  A pool set several times the size of the Infinity Cache (256 MiB) gets nothing out of the cache levels on
// its way to the CU -- every byte is used once per launch and evicted before the next -- and letting it allocate there costs
// bandwidth: at 1.28 GB per evaluation (4e7 constant-product pools) the `nt` loads run the evaluation at 6.59 TB/s instead of
// 6.16 (194 against 208 us).  At 320 MB, where the ping-pong walk finds most of a launch still cached, they LOSE (52.4 against
// 47.8 us): the host takes the NT instantiations above twice the cache size only (cfmm_hip.hip: stream_nt).
// The kernels' NT parameter is the host's large_set_mode(): 0 the plain instantiation (no mirror, no wide tiles compiled in),
// 1 large sets with non-temporal loads, 2 large sets with cached loads.
template <bool NT = false, class T>
__device__ __forceinline__ T ld_off(const T *base, unsigned i)
{
    const T *p = reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (size_t)(i * (unsigned)sizeof(T)));
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

template <bool DET> struct Scatter;
template <> struct Scatter<false> {
    double *t; int n; double sc;
    __device__ __forceinline__ void add(int tok, double y) const { unsafeAtomicAdd(&t[tok], y); }
};
template <> struct Scatter<true> {
    double *t; int n; double sc;                       // t: [3][n] unsigned 64-bit limbs, limb-major
    __device__ __forceinline__ void add(int tok, double y) const
    {
        unsigned long long *L = reinterpret_cast<unsigned long long *>(t);
        const double yf = y * sc;                                  // exact (power of two)
        const double a2 = floor(yf * 0x1p-64);
        const double r = fma(-a2, 0x1p64, yf);                     // exact, in [0, 2^64)
        const double a1 = floor(r * 0x1p-32);
        const double a0 = floor(fma(-a1, 0x1p32, r));              // in [0, 2^32); the fraction below 2^-F is dropped
        atomicAdd(&L[tok], (unsigned long long)(unsigned)__double2uint_rz(a0));
        atomicAdd(&L[n + tok], (unsigned long long)(unsigned)__double2uint_rz(a1));
        atomicAdd(&L[2 * n + tok], (unsigned long long)(long long)__double2int_rz(a2));
    }
};

// ------------------------------------------------------------------------------------------
// Batched evaluation (eval_batch_kernel: B price vectors per pool read -- the parametric sweep of two-asset.py:34-100,
// independent baskets over one pool set): the B price vectors sit `nu_stride` doubles apart in LDS, their psi tiles
// `tile_stride` apart; `alive` has one bit per vector whose solve is still running (wave-uniform).  A tile loads its
// pool columns ONCE and solves the pools at every live price vector.
// ------------------------------------------------------------------------------------------
constexpr int BATCH_MAX = 8;
struct BatchCtl { unsigned alive; int nu_stride, tile_stride; };

// ------------------------------------------------------------------------------------------
// The STAGED tile walk (round 4): every wave owns a 4 KB LDS slot, and the pool columns of its NEXT wave-tile are brought
// into it by LDS-DMA (global_load_lds_dwordx4: 16 B per lane, 1 KB per wave-instruction, no VGPRs, asynchronous) while it
// computes the current tile -- and, in iter_kernel, the FIRST tile of every wave while the in-launch nu update's latency
// chain still runs (the memory system idles there: 7 us of every 21 at C3, 13 of 59 at C4).  Per tile:
//     s_waitcnt vmcnt(0)          this tile's columns have landed (the only thing outstanding)
//     tile_stage_read             LDS slot -> 6 doubles + 4 ints per lane (lane-linear ds_reads)
//     s_waitcnt lgkmcnt(0)        ... and are in registers: the slot is free
//     tile_dma_issue(next)        2-4 DMA instructions for the next ticket's tile (any bucket)
//     tile2 / tilen <..., PRE>    the pool arithmetic on the registers
// so no tile of the launch waits for HBM / L2 latency, head and tail of the launch included (a wave-tile loaded at its
// own start exposes one full memory latency: at C3 a wave runs only ~4 tiles per launch).  The DMA instructions are
// inline asm: hipcc neither orders ds_reads behind the builtin form nor lets ordinary loads pass it (it drains vmcnt(0)),
// so the two waits above are placed by hand and the walk contains no compiler-visible global load.  Slot layouts (bytes):
//   two pools per lane (cp2, sum2; 128 pools): Ra 0 | Rb 1024 | fee 2048 | ia 3072 | ib 3584
//   one pool per lane (w2, curve2, pow2; 64):  Ra 0 | Rb 512 | fee 1024 | param 1536 | ia 2048 | ib 2304
//   K-asset, leg per lane:                     R 0 | w 512 | idx 1024 | fee 1280 (32 pools) | log fee 1536
// Every piece is 16-byte aligned in global memory (columns are 256-byte aligned, tiles start on multiples of 64 / 128 pools,
// K-asset tiles on multiples of 4 legs and 2 pools: ktile_pools); chunks past a column's end are clamped to its last chunk
// (the lanes that would use them are not live; a column is followed by >= 16 readable bytes: cfmm_hip.hip, upload_arena).
// ------------------------------------------------------------------------------------------
constexpr int STAGE_BYTES = 4096;                       // per wave
struct TileRegs { double d[6]; int i[4]; };            // a staged wave-tile's columns, as one lane holds them

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst)      // lane l: 16 B at gsrc -> LDS lds_dst + 16 l
{
    unsigned keep;                                      // (M0 is compiler-reserved: saved and restored inside the statement)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// a workgroup barrier that orders LDS traffic only.  __syncthreads() carries a fence, and wherever ANY path into it has a
// global store or load pending the compiler places s_waitcnt vmcnt(0) in front of the barrier -- which at run time also waits
// for the LDS-DMA pieces in flight (they count on vmcnt): the update chain of iter_kernel stood still until its first tiles
// had landed (+1.8 us per launch at C4).  Nothing that crosses these barriers goes through global memory.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void *p) { return __builtin_amdgcn_readfirstlane((unsigned)(size_t)p); }   // (flat -> LDS offset: the low 32 bits)
// a column base out of the kernel arguments, pinned to SGPRs: without this the compiler turns `lane < 32 ? b.R : b.w` into a
// per-lane VECTOR load of the pointer from the argument block (a load the walk would have to wait for)
__device__ __forceinline__ const char *uni_ptr(const void *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const char *>(((unsigned long long)hi << 32) | lo);
}

// ------------------------------------------------------------------------------------------
// one wave-tile of a two-asset bucket: lane l solves pools i0 + l + 64 u, u < U.  All 5U column
// loads are issued before the first use (each 512 B coalesced per wave).
// 32 B (CP2, SUM2) or 40 B (W2, CURVE2, POW2: + the parameter column) of HBM per pool, read once.
// ------------------------------------------------------------------------------------------
// PRE: the columns come from `pre` (the staged walk: tile_stage_read) instead of global memory
// CARRY: sum arb is carried per lane (fsum); without it the caller forms it at the flush as nu' psi (sum_i arb_i = sum_i nu' y_i)
// UU: pools per lane (0 = the kind's own: wave_tile_pools / 64); the wide constant-product tiles of large buckets pass 4
template <int KIND, bool WITH_D, bool DET, bool BATCH = false, bool PRE = false, int NT = 0, bool CARRY = true, int UU = 0>
__device__ __forceinline__ void tile2(const Bucket2 &b, long long i0, int lane, const double *nu_s,
                                      const Scatter<DET> &psi_s, const Scatter<DET> &diag_s, double &fsum, const BatchCtl &bc,
                                      const TileRegs *pre = nullptr)
{
    static_assert(!(BATCH && WITH_D), "the batched evaluation does not build the metric");
    constexpr int U = UU ? UU : wave_tile_pools(KIND) / 64;
    asm volatile("" : "+v"(lane));              // (opaque: keeps per-kind lane arithmetic from being hoisted out of the tile loop)
    double Ra[U], Rb[U], g[U], prm[U];
    int ia[U], ib[U], fl[U];
    unsigned idx[U];
    bool live[U];
    // The compact mirror is compiled into the large-set instantiations only (NT >= 1), with ONE branch around their whole load
    // clause; everything else keeps the loads inside the per-pool loop below, untouched.  Both mattered: with the test between
    // the loads of a pool the compiler split the clause (C3 +0.8 us per launch), with the loads moved into a loop of their own
    // behind the index loop it issued the two id loads, waited for them and only then issued the reserve loads (C3 +0.3, C2 +0.25).
    constexpr bool BIG = !PRE && KIND <= 2 && NT >= 1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        // (32-bit element offsets against the uniform column bases -- a bucket holds < 2^29 pools -- instead of seven 64-bit
        //  address computations per lane)
        unsigned i = (unsigned)i0 + u * 64 + lane;
        live[u] = i < (unsigned long long)b.m;
        i = live[u] ? i : (unsigned)b.m - 1u;
        idx[u] = i;
        if constexpr (PRE) {
            Ra[u] = pre->d[u]; Rb[u] = pre->d[U + u]; g[u] = pre->d[2 * U + u];
            ia[u] = live[u] ? pre->i[u] : 0; ib[u] = live[u] ? pre->i[U + u] : 0;     // (a dead lane holds a clamped chunk's tail: not a token id)
            prm[u] = (KIND == 1 || KIND >= 3) ? pre->d[U == 1 ? 3 : 0] : 0.0;
        } else if constexpr (!BIG) {
            Ra[u] = ld_off<NT == 1>(b.Ra, i); Rb[u] = ld_off<NT == 1>(b.Rb, i); g[u] = ld_off<NT == 1>(b.fee, i);
            ia[u] = ld_off<NT == 1>(b.ia, i); ib[u] = ld_off<NT == 1>(b.ib, i);
            prm[u] = (KIND == 1 || KIND >= 3) ? ld_off<NT == 1>(b.param, i) : 0.0;
        }
        if constexpr (!BIG) fl[u] = (KIND == 2 && b.flags) ? ld_off(b.flags, i) : 0;
    }
    if constexpr (BIG) {
        if (b.cid != nullptr) {                 // (wave-uniform: a property of the bucket)
            unsigned pk[U];
            unsigned char fi[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                Ra[u] = ld_off<NT == 1>(b.Ra, idx[u]); Rb[u] = ld_off<NT == 1>(b.Rb, idx[u]);
                pk[u] = ld_off<NT == 1>(b.cid, idx[u]); fi[u] = ld_off<NT == 1>(b.cfee, idx[u]);
                prm[u] = KIND == 1 ? ld_off<NT == 1>(b.param, idx[u]) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { g[u] = b.ctab[fi[u]]; ia[u] = (int)(pk[u] & 0xffffu); ib[u] = (int)(pk[u] >> 16); }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                Ra[u] = ld_off<NT == 1>(b.Ra, idx[u]); Rb[u] = ld_off<NT == 1>(b.Rb, idx[u]); g[u] = ld_off<NT == 1>(b.fee, idx[u]);
                ia[u] = ld_off<NT == 1>(b.ia, idx[u]); ib[u] = ld_off<NT == 1>(b.ib, idx[u]);
                prm[u] = KIND == 1 ? ld_off<NT == 1>(b.param, idx[u]) : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) fl[u] = (KIND == 2 && b.flags) ? ld_off(b.flags, idx[u]) : 0;
    }
#pragma unroll 1
    for (unsigned mask = BATCH ? bc.alive : 1u; mask; mask &= mask - 1) {
        const int bb = BATCH ? __builtin_ctz(mask) : 0;
        const double *nub = BATCH ? nu_s + bb * bc.nu_stride : nu_s;
        const Scatter<DET> ps{BATCH ? psi_s.t + bb * bc.tile_stride : psi_s.t, psi_s.n, psi_s.sc};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double pa = nub[ia[u]], pb = nub[ib[u]];
            if constexpr (KIND == 0 && !WITH_D) {          // constant product, directed form (pool_math.hpp)
                const Y2dir y = pool_cp2_dir(Ra[u], Rb[u], g[u], pa, pb);
                if (live[u] && y.active) {
                    ps.add(y.ab ? ia[u] : ib[u], y.yin);
                    ps.add(y.ab ? ib[u] : ia[u], y.yout);
                    if (CARRY && !DET && !BATCH) fsum += y.arb;
                }
                continue;
            }
            Y2 y;
            if (KIND == 0) y = pool_cp2(Ra[u], Rb[u], g[u], pa, pb);
            else if (KIND == 1) y = pool_w2<DET>(Ra[u], Rb[u], g[u], prm[u], pa, pb);
            else if (KIND == 2) { y = pool_sum2(Ra[u], Rb[u], g[u], pa, pb); if (fl[u]) { y.ya = 0.0; y.yb = 0.0; } }
            else if (KIND == 3) y = pool_curve2(Ra[u], Rb[u], g[u], prm[u], pa, pb);
            else y = pool_generic2<(KIND >= 4 ? KIND : 4)>(Ra[u], Rb[u], g[u], prm[u], pa, pb);      // every kind without a hand-tuned closed form
            if (live[u] && (y.ya != 0.0 || y.yb != 0.0)) {
                ps.add(ia[u], y.ya);
                ps.add(ib[u], y.yb);
                if (CARRY && !DET && !BATCH) fsum += pa * y.ya + pb * y.yb;
            }
            if (WITH_D && live[u] && KIND != 2) {
                double da = 0.0, db = 0.0;
                if (KIND == 0) { da = 0.5 * pa * Ra[u]; db = 0.5 * pb * Rb[u]; }
                else if (KIND == 1) { da = (1.0 - prm[u]) * pa * Ra[u]; db = prm[u] * pb * Rb[u]; }
                else if (KIND == 3) curve_diag(Ra[u], Rb[u], prm[u], pa, pb, da, db);
                else generic_diag<(KIND >= 4 ? KIND : 4)>(Ra[u], Rb[u], prm[u], pa, pb, da, db);
                diag_s.add(ia[u], da);
                diag_s.add(ib[u], db);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// one wave-tile of a K-asset geo-mean bucket, LEG PER LANE: the K legs of a pool sit on K
// consecutive lanes (64 / K pools per wave) and are stored pool-major ("CSR with constant row
// length": leg j of pool i at [i*K + j]), so a wave's idx / R / w loads are three fully coalesced
// 256-512 B transactions for ANY K; 20 + 20 K bytes per pool (fee, log fee, K x {id, R, w}), read once.
//
// KKT (pool_math.hpp): x_j(t) = R_j e^{f(t - a_j)}, a_j = log(R_j p_j / w_j), and the residual
// F(t) = sum_j w_j f(t - a_j) is non-decreasing, so leg j is WITHDRAWN at the root t* iff
// F(a_j) > 0 and DEPOSITED iff F(a_j - lg) < 0: every lane evaluates F at its own two kinks
// (K terms each, the (a, w) pairs of its pool exchanged through a wave-private LDS strip), learns
// its own membership, and the root follows from one K-lane sum:
//     t* = [sum_W w_j a_j + sum_D w_j (a_j - lg)] / [sum_W w_j + sum_D w_j].
// No sort, no bracket scan, no data-dependent loop, ~40 VGPRs; the serial chain per wave is
// ~350 instructions instead of ~2000 for one-pool-per-lane at K = 8.
// ------------------------------------------------------------------------------------------
template <int K>
__host__ __device__ constexpr int pools_per_wave() { return ktile_pools(K); }

// LNU: a = lrw + lnu_s[token] (the bucket's log(R / w) column and the workgroup's table of log-prices) instead of log(R p / w)
template <int K, bool WITH_D, bool DET, bool BATCH = false, bool PRE = false, int NT = 0, bool CARRY = true, bool LNU = false>
__device__ __forceinline__ void tilen(const BucketN &b, long long tb, int lane, const double *nu_s,
                                      const Scatter<DET> &psi_s, const Scatter<DET> &diag_s, double2 *xs, double &fsum, const BatchCtl &bc,
                                      const TileRegs *pre = nullptr, const double *lnu_s = nullptr)
{
    static_assert(!(BATCH && WITH_D), "the batched evaluation does not build the metric");
    static_assert(!LNU || (!BATCH && !PRE && !WITH_D && !CARRY), "the log-price table serves the plain evaluation tiles");
    constexpr int P = pools_per_wave<K>();
    // (opaque copy: otherwise the lane / K, lane % K and strip addresses of all six instantiations are hoisted out of
    //  the tile loop and stay live across it -- ~20 VGPRs on a kernel that sits on a register cliff)
    asm volatile("" : "+v"(lane));
    const int g = lane / K, j = lane - g * K;
    // (32-bit element offsets against the uniform column bases: a bucket holds < 2^28 legs, and the loads then take the
    //  scalar-base + 32-bit-offset form instead of five 64-bit address computations per lane)
    const unsigned pool = (unsigned)tb * P + g;
    const bool live = (g < P) && (pool < (unsigned long long)b.m);
    const unsigned leg = live ? pool * K + j : 0u, pl = live ? pool : 0u;
    int tok;
    double R, w, fee, lg;
    if constexpr (PRE) { tok = live ? pre->i[0] : 0; R = pre->d[0]; w = pre->d[1]; fee = pre->d[2]; lg = pre->d[3]; (void)leg; (void)pl; }     // (dead lanes: a neighbouring tile's legs or a clamped chunk's tail)
    else { tok = ld_off<NT == 1>(b.idx, leg); R = ld_off<NT == 1>(b.R, leg); w = ld_off<NT == 1>(b.w, leg); fee = ld_off<NT == 1>(b.fee, pl); lg = ld_off<NT == 1>(b.lfee, pl); }
    double lrw = 0.0;
    if constexpr (LNU) lrw = ld_off<NT == 1>(b.lrw, leg);
    const int gb = (g < P ? g : 0) * K;
#pragma unroll 1
    for (unsigned mask = BATCH ? bc.alive : 1u; mask; mask &= mask - 1) {
        const int bb = BATCH ? __builtin_ctz(mask) : 0;
        const Scatter<DET> ps{BATCH ? psi_s.t + bb * bc.tile_stride : psi_s.t, psi_s.n, psi_s.sc};
        const double p = LNU ? 0.0 : (BATCH ? nu_s + bb * bc.nu_stride : nu_s)[tok];
        SCHED_FENCE();
        double a;
        if constexpr (LNU) a = lrw + lnu_s[tok]; else a = log_pos(R * p * rcp_nr(w));
        SCHED_FENCE();
        xs[lane] = make_double2(a, w);                     // ds_write_b128; same-wave LDS ops stay in order
        __builtin_amdgcn_wave_barrier();
        const double t1 = a, t2 = a - lg;
        // (NOT rewritten as min(u,0) + max(u,L) - L etc., two operations fewer per term: inside a pool's no-trade band F is
        //  EXACTLY zero in this form -- a sum of exact zeros -- and the strict sign tests below rely on it; the rewritten sums
        //  come out as +-1e-19 there and flag legs of pools that must not trade)
        // (written as u - clamp(u, 0, -lg): the same three cases -- u, 0, u + lg -- bit for bit [x - (-y) IS x + y], one operation
        //  fewer per term, and still an exact zero inside the band)
        double f1 = 0.0, f2 = 0.0;
        const double nlg = -lg;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double2 v = xs[gb + k];
            const double u1 = t1 - v.x, u2 = t2 - v.x;
            f1 += v.y * (u1 - fmin(fmax(u1, 0.0), nlg));
            f2 += v.y * (u2 - fmin(fmax(u2, 0.0), nlg));
        }
        SCHED_FENCE();
        const bool wd = f1 > 0.0;                           // withdrawn at the root: t* < a_j
        const bool dp = f2 < 0.0;                           // deposited at the root: t* > a_j - lg
        const double den_j = (wd || dp) ? w : 0.0;
        const double num_j = wd ? w * t1 : (dp ? w * t2 : 0.0);
        __builtin_amdgcn_wave_barrier();
        xs[lane] = make_double2(num_j, den_j);
        __builtin_amdgcn_wave_barrier();
        double num = 0.0, den = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) { const double2 v = xs[gb + k]; num += v.x; den += v.y; }
        __builtin_amdgcn_wave_barrier();
        SCHED_FENCE();
        double y = 0.0;
        if (live && den > 0.0 && (wd || dp)) {             // (live: a dead lane's numbers must not push the wave off expm1_wave's short path)
            const double t = num * rcp_nr(den);
            const double rx = -R * expm1_wave<DET>(wd ? t - t1 : t - t2);    // R - x,  x = R e^{f(t - a_j)}
            y = wd ? rx : rx * rcp_nr(fee);
        }
        SCHED_FENCE();
        if (live) {
            if (y != 0.0) { ps.add(tok, y); if (CARRY && !DET && !BATCH) fsum += p * y; }
            if (WITH_D) diag_s.add(tok, (1.0 - w) * p * R);
        }
    }
}

// ------------------------------------------------------------------------------------------
// The staged walk's two halves around a tile (layouts: above).  bk = the tile's bucket in processing order
// (0..5: K-asset, K = 8 - bk; 6 curve2, 7 pow2, 8 w2: one pool per lane; 9 cp2, 10 sum2: two per lane), tb = the tile's index
// inside its bucket.  Both are wave-uniform; the bucket's column pointers come out of the kernel arguments by index.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int bucket_kind2(int bk) { return bk == 6 ? 3 : (bk == 7 ? 4 : (bk == 8 ? 1 : (bk == 9 ? 0 : 2))); }

__device__ __forceinline__ void tile_dma_issue(const EvalArgs &a, int bk, int tb, unsigned stage, int lane)
{
    // (every column base and every source address is formed BEFORE the first DMA statement: the asm statements are memory
    //  barriers to the compiler, a scalar load of a base placed between two of them would be waited for in between)
    const int l32 = lane & 31, l16 = lane & 15;
    const bool hi = lane >= 32;
    if (bk <= 5) {
        const int K = 8 - bk;
        const BucketN &b = a.bn[5 - bk];
        const char *pR = uni_ptr(b.R), *pW = uni_ptr(b.w), *pI = uni_ptr(b.idx), *pF = uni_ptr(b.fee), *pL = uni_ptr(b.lfee);
        const unsigned P = (unsigned)((0x14100c0a0808ull >> (8 * bk)) & 0xff);          // ktile_pools(K)
        const unsigned m = (unsigned)b.m, nleg = m * (unsigned)K;
        const unsigned pool0 = (unsigned)tb * P, leg0 = pool0 * (unsigned)K;
        // lanes 0-31: R[leg0 + 2 l ..], lanes 32-63: w[...]  ->  slot bytes [0, 512) | [512, 1024)
        unsigned e = leg0 + 2u * l32;
        const unsigned lim = (nleg - 1u) & ~1u;
        e = e < lim ? e : lim;
        const char *s0 = (hi ? pW : pR) + (size_t)e * 8u;
        // lanes 0-15: idx[leg0 + 4 l ..] -> [1024, 1280); 16-31: fee[pool0 + 2 (l - 16) ..] -> [1280, 1536); 32-47: log fee -> [1536, 1792)
        unsigned e4 = leg0 + 4u * l16, e2 = pool0 + 2u * l16;
        const unsigned lim4 = (nleg - 1u) & ~3u, lim2 = (m - 1u) & ~1u;
        e4 = e4 < lim4 ? e4 : lim4; e2 = e2 < lim2 ? e2 : lim2;
        const char *s1 = lane < 16 ? pI + (size_t)e4 * 4u : (hi ? pL : pF) + (size_t)e2 * 8u;
        glds16(s0, stage);
        if (lane < 48) glds16(s1, stage + 1024u);
    } else if (bk <= 8) {
        const Bucket2 &b = a.b2[bucket_kind2(bk)];
        const char *pA = uni_ptr(b.Ra), *pB = uni_ptr(b.Rb), *pF = uni_ptr(b.fee), *pP = uni_ptr(b.param), *pIa = uni_ptr(b.ia), *pIb = uni_ptr(b.ib);
        const unsigned m = (unsigned)b.m, e0 = (unsigned)tb * WT_HEAVY;
        unsigned e = e0 + 2u * l32, e4 = e0 + 4u * l16;
        const unsigned lim = (m - 1u) & ~1u, lim4 = (m - 1u) & ~3u;
        e = e < lim ? e : lim; e4 = e4 < lim4 ? e4 : lim4;
        const char *s0 = (hi ? pB : pA) + (size_t)e * 8u, *s1 = (hi ? pP : pF) + (size_t)e * 8u, *s2 = (lane >= 16 ? pIb : pIa) + (size_t)e4 * 4u;
        glds16(s0, stage);                                   // Ra | Rb
        glds16(s1, stage + 1024u);                           // fee | param
        if (lane < 32) glds16(s2, stage + 2048u);            // ia | ib
    } else {
        const Bucket2 &b = a.b2[bk == 9 ? 0 : 2];
        const char *pA = uni_ptr(b.Ra), *pB = uni_ptr(b.Rb), *pF = uni_ptr(b.fee), *pIa = uni_ptr(b.ia), *pIb = uni_ptr(b.ib);
        const unsigned m = (unsigned)b.m, e0 = (unsigned)tb * WT_LIGHT;
        unsigned e = e0 + 2u * lane, e4 = e0 + 4u * l32;
        const unsigned lim = (m - 1u) & ~1u, lim4 = (m - 1u) & ~3u;
        e = e < lim ? e : lim; e4 = e4 < lim4 ? e4 : lim4;
        const size_t off = (size_t)e * 8u;
        static_assert(!CFMM_STAGED_WALK || WT_LIGHT == 128, "slot layout of the two-pools-per-lane tiles");
        const char *s0 = pA + off, *s1 = pB + off, *s2 = pF + off, *s3 = (hi ? pIb : pIa) + (size_t)e4 * 4u;
        glds16(s0, stage);
        glds16(s1, stage + 1024u);
        glds16(s2, stage + 2048u);
        glds16(s3, stage + 3072u);                           // ia | ib
    }
}

__device__ __forceinline__ void tile_stage_read(int bk, const double *sd, int lane, TileRegs &r)
{
    const int *si = reinterpret_cast<const int *>(sd);
    if (bk <= 5) {
        const int g = (lane * (int)((0x55900334ab24c80ull >> (10 * bk)) & 0x3ff)) >> 10;      // lane / K  (ceil(1024 / K), exact for lane < 64)
        r.d[0] = sd[lane]; r.d[1] = sd[64 + lane]; r.d[2] = sd[160 + g]; r.d[3] = sd[192 + g];
        r.i[0] = si[256 + lane];
        r.d[4] = r.d[5] = 0.0; r.i[1] = r.i[2] = r.i[3] = 0;
    } else if (bk <= 8) {
        r.d[0] = sd[lane]; r.d[1] = sd[64 + lane]; r.d[2] = sd[128 + lane]; r.d[3] = sd[192 + lane];
        r.i[0] = si[512 + lane]; r.i[1] = si[576 + lane];
        r.d[4] = r.d[5] = 0.0; r.i[2] = r.i[3] = 0;
    } else {
        r.d[0] = sd[lane]; r.d[1] = sd[64 + lane]; r.d[2] = sd[128 + lane]; r.d[3] = sd[192 + lane];
        r.d[4] = sd[256 + lane]; r.d[5] = sd[320 + lane];
        r.i[0] = si[768 + lane]; r.i[1] = si[832 + lane]; r.i[2] = si[896 + lane]; r.i[3] = si[960 + lane];
    }
}

// where a walk index lies: the bucket pointer of a wave only ever moves one way (its tickets grow)
struct TileCursor {
    int bk, cstart, cend, sfirst;
    __device__ __forceinline__ void init(const int *tab, bool rev)
    {
        bk = rev ? N_BUCKETS - 1 : 0;
        cstart = rev ? __builtin_amdgcn_readfirstlane(tab[N_BUCKETS - 2]) : 0;
        cend = __builtin_amdgcn_readfirstlane(tab[bk]); sfirst = __builtin_amdgcn_readfirstlane(tab[N_BUCKETS + bk]);
    }
    // the same for ANY walk index in two LDS round trips (lanes 0..N_BUCKETS-1 compare one table entry each): the first
    // ticket of a wave, which the serial walk reaches only after up to N_BUCKETS - 1 dependent steps
    __device__ __forceinline__ int find(const int *tab, int i, int lane)
    {
        const int c = lane < N_BUCKETS ? tab[lane] : 0x7fffffff;
        const unsigned long long below = __ballot(c <= i) & ((1ull << N_BUCKETS) - 1);
        bk = __builtin_popcountll(below);                // buckets that end at or before i (the counts are cumulative: a prefix)
        cend = __builtin_amdgcn_readlane(c, bk);
        cstart = bk ? __builtin_amdgcn_readlane(c, bk - 1) : 0;
        sfirst = __builtin_amdgcn_readfirstlane(tab[N_BUCKETS + bk]);
        return sfirst + (i - cstart);
    }
    __device__ __forceinline__ int seek(const int *tab, int i)      // -> the tile's index inside bucket `bk`
    {
        while (i >= cend) {
            ++bk; cstart = cend;
            cend = __builtin_amdgcn_readfirstlane(tab[bk]);
            sfirst = __builtin_amdgcn_readfirstlane(tab[N_BUCKETS + bk]);
        }
        while (i < cstart) {
            --bk; cend = cstart;
            cstart = bk ? __builtin_amdgcn_readfirstlane(tab[bk - 1]) : 0;
            sfirst = __builtin_amdgcn_readfirstlane(tab[N_BUCKETS + bk]);
        }
        return sfirst + (i - cstart);
    }
};

// the first DMA of a wave's staged walk (its first ticket is its own index): iter_kernel issues it under the update
__device__ __forceinline__ void tiles_dma_first(const EvalArgs &a, const int *next_tile, unsigned stage, int lane, int wib)
{
    const int *tab = next_tile + 2;
    const int nlocal = __builtin_amdgcn_readfirstlane(tab[N_BUCKETS - 1]);
    if (wib >= nlocal) return;
    const bool rev = a.rev != 0;
    TileCursor c;
    const int tb = c.find(tab, rev ? nlocal - 1 - wib : wib, lane);
    tile_dma_issue(a, c.bk, tb, stage, lane);
}

// ------------------------------------------------------------------------------------------
// The workgroup's tile-range table (see eval_tiles_and_flush), built by the first N_BUCKETS lanes of ONE wave: lane q takes
// bucket q's range [b n_q / G, (b + 1) n_q / G), an inclusive scan over the lanes (DPP row_shr inside the row of 16: no
// LDS round trips, no barrier of its own) gives the cumulative counts.  Also arms the ticket counter: the first ticket of
// wave w is w.  The caller passes a barrier between this and the tile loop.
// ------------------------------------------------------------------------------------------
// `whole`: this workgroup walks EVERY tile (the workgroups of a sweep launch are independent solves: tiny.hpp, BATCH)
__device__ __forceinline__ void build_tile_table(const EvalArgs &a, int *next_tile, int lane, bool whole = false)
{
    int *tab = next_tile + 2;
    int cnt = 0, s = 0;
    if (lane < N_BUCKETS) {
        const int nq = a.tile_end[lane] - (lane ? a.tile_end[lane - 1] : 0);
        const double G = whole ? 1.0 : (double)gridDim.x;           // (products < 2^47: exact in fp64, and the floors below cannot be off by one)
        const double b = whole ? 0.0 : (double)blockIdx.x;
        s = (int)((b * nq) / G);
        cnt = (int)(((b + 1.0) * nq) / G) - s;
    }
    int c = cnt;
    c += __builtin_amdgcn_update_dpp(0, c, 0x111, 0xf, 0xf, true);     // row_shr:1 (lanes without a source read 0)
    c += __builtin_amdgcn_update_dpp(0, c, 0x112, 0xf, 0xf, true);     // row_shr:2
    c += __builtin_amdgcn_update_dpp(0, c, 0x114, 0xf, 0xf, true);     // row_shr:4
    c += __builtin_amdgcn_update_dpp(0, c, 0x118, 0xf, 0xf, true);     // row_shr:8
    if (lane < N_BUCKETS) { tab[lane] = c; tab[N_BUCKETS + lane] = s; }
    if (lane == 0) *next_tile = (int)(blockDim.x >> 6);
}

// ------------------------------------------------------------------------------------------
// The dual evaluation, ONE launch for every bucket:  psi(nu) = sum_i A_i (L_i - D_i),
// sum_i arb_i(A_i' nu), optionally the diagonal metric.                reference: arbitrage.py:54
//
// LDS: psi_s[n] | (diag_s[n]) | nu_s[n + 1] | fpart[16] | ticket | wave-private exchange strips 64 x 16 B.  Every wave walks its own wave-tiles
// (tile = pass * W + wave_in_block * gridDim + block, W = waves in the grid): the waves of one
// workgroup take tiles W/8 apart, so each CU holds the same mix of ALU-heavy geo-mean tiles and
// streaming constant-product tiles; no barrier between the prologue and the epilogue.  The
// epilogue flushes the workgroup's psi tile into accumulator slice blockIdx % nslices
// (global_atomic_add_f64).
// ------------------------------------------------------------------------------------------
// STABLE = false: every bucket but the stableswap one; STABLE = true: the stableswap bucket alone (its Newton loops need
// ~20 more VGPRs than anything else: kept out of the main instantiation, it lets that one run at 6 waves per SIMD)
// the tile loop and the flush, shared by eval_kernel (below) and iter_kernel (iterate.hpp).  On entry nu_s holds the
// prices, psi_s (diag_s) are zero, build_tile_table has run and a barrier has been passed; `acc` is the accumulator set to flush into.
// BATCH: `bc` describes the B price vectors / psi tiles in LDS, `acc_b[b]` is where vector b's tile is flushed; sum arb
// is formed at the flush as nu' psi per vector (sum_i arb_i = sum_i nu' y_i) instead of being carried per lane.
// FLUSH = false: the tiles stay in LDS (psi_t, diag_t, fpart[wave] = per-wave partial of sum arb) for a consumer in the same
// workgroup (tiny.hpp)
// DMA: the staged walk (above) -- `stage` is this wave's 4 KB LDS slot, `first_issued` says that the caller has already
// issued the DMA of the wave's first tile (tiles_dma_first)
// LNU: `lnu_s` = the workgroup's table of log-prices (filled by the caller next to nu_s): the K-asset tiles take a = log(R p / w) from it
template <bool WITH_D, bool STABLE, bool DET = false, bool BATCH = false, bool FLUSH = true, bool DMA = false, int NT = 0, bool LNU = false>
__device__ __forceinline__ void eval_tiles_and_flush(const EvalArgs &a, double *acc, const double *nu_s, double *psi_t, double *diag_t,
                                                     double *fpart, int *next_tile, double2 *xs, const BatchCtl &bc = BatchCtl{1u, 0, 0},
                                                     double *const *acc_b = nullptr, const double *stage = nullptr, bool first_issued = false,
                                                     const double *lnu_s = nullptr)
{
    static_assert(!LNU || (!WITH_D && !DET && !BATCH && !DMA && FLUSH), "the log-price table: plain evaluations that flush");
    const int n = a.n;
    const Scatter<DET> psi_s{psi_t, n, a.det_scale}, diag_s{diag_t, n, a.det_scale_d};
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double fsum = 0.0;
    if constexpr (DMA) {
        static_assert(!STABLE && !BATCH && FLUSH, "the staged walk serves the main tile space only");
        const int *tab = next_tile + 2;
        const int nlocal = __builtin_amdgcn_readfirstlane(tab[N_BUCKETS - 1]);
        const bool rev = a.rev != 0;
        const unsigned slot = lds_addr(stage);
        TileCursor cur; cur.init(tab, rev);
        int i0 = wib, bk = 0, tb = 0, ticket = 0;
        if (i0 < nlocal) {
            tb = cur.find(tab, rev ? nlocal - 1 - i0 : i0, lane); bk = cur.bk;
            if (!first_issued) tile_dma_issue(a, bk, tb, slot, lane);
        }
        while (i0 < nlocal) {
            if (lane == 0) ticket = atomicAdd(next_tile, 1);           // the next ticket: its LDS round trip rides under the read-out
            TileRegs r;
            dma_wait();                                                // this tile's columns have landed in the slot
            tile_stage_read(bk, stage, lane, r);
            lds_wait();                                                // ... and sit in registers: the slot is free for the next tile
            const int n0 = __builtin_amdgcn_readfirstlane(ticket);
            int nbk = 0, ntb = 0;
            if (n0 < nlocal) {
                ntb = cur.seek(tab, rev ? nlocal - 1 - n0 : n0); nbk = cur.bk;
                tile_dma_issue(a, nbk, ntb, slot, lane);
            }
            switch (bk) {
            case 0: tilen<8, WITH_D, DET, false, true, false, !FLUSH>(a.bn[5], tb, lane, nu_s, psi_s, diag_s, xs, fsum, bc, &r); break;
            case 1: tilen<7, WITH_D, DET, false, true, false, !FLUSH>(a.bn[4], tb, lane, nu_s, psi_s, diag_s, xs, fsum, bc, &r); break;
            case 2: tilen<6, WITH_D, DET, false, true, false, !FLUSH>(a.bn[3], tb, lane, nu_s, psi_s, diag_s, xs, fsum, bc, &r); break;
            case 3: tilen<5, WITH_D, DET, false, true, false, !FLUSH>(a.bn[2], tb, lane, nu_s, psi_s, diag_s, xs, fsum, bc, &r); break;
            case 4: tilen<4, WITH_D, DET, false, true, false, !FLUSH>(a.bn[1], tb, lane, nu_s, psi_s, diag_s, xs, fsum, bc, &r); break;
            case 5: tilen<3, WITH_D, DET, false, true, false, !FLUSH>(a.bn[0], tb, lane, nu_s, psi_s, diag_s, xs, fsum, bc, &r); break;
            case 8: tile2<1, WITH_D, DET, false, true, false, !FLUSH>(a.b2[1], (long long)tb * WT_HEAVY, lane, nu_s, psi_s, diag_s, fsum, bc, &r); break;
            case 9: tile2<0, WITH_D, DET, false, true, false, !FLUSH>(a.b2[0], (long long)tb * WT_LIGHT, lane, nu_s, psi_s, diag_s, fsum, bc, &r); break;
            case 10: tile2<2, WITH_D, DET, false, true, false, !FLUSH>(a.b2[2], (long long)tb * WT_LIGHT, lane, nu_s, psi_s, diag_s, fsum, bc, &r); break;
            default: break;                                            // (6, 7: the heavy kinds live in the other tile space)
            }
            i0 = n0; bk = nbk; tb = ntb;
        }
    } else {
#ifdef CFMM_PHASE_TIMERS
    int nlog = 0;
    long long t_prev = clock64(), t_out = 0;         // time spent between tiles (ticket, bucket search, dispatch)
#endif
    // Workgroup b owns a CONTIGUOUS range of every bucket's wave-tiles, [b n_q / G, (b + 1) n_q / G) of bucket q's n_q:
    // every workgroup sees the same mix of buckets, heaviest first, and -- the point -- with the pools of a bucket ordered by
    // token blocks at upload (reorder.hpp) the pools of ONE workgroup touch few tokens: its psi tile is sparse, the flush
    // sends a few hundred atomics instead of n, the LDS gathers / scatters stay in a small window (C4 shard: evaluation
    // 16.1 -> 10.5 us).  Its waves draw the index i into the concatenation of its ranges from an LDS ticket counter
    // (ds_add_rtn, ~100 cycles against >= 1 us per tile), so they all finish within one tile of each other however
    // uneven the tile costs are.  Range table (LDS, behind the ticket; build_tile_table, by ONE wave in the caller's
    // prologue, in front of a barrier the caller has anyway): tab[q] = tiles of buckets <= q in this workgroup,
    // tab[N_BUCKETS + q] = first tile of its range inside bucket q.  The first ticket of wave w is w (the counter starts
    // at the number of waves).
    // a.rev: the launch walks the concatenation BACKWARDS.  The host alternates the direction from launch to launch
    // (ping-pong): what a launch read LAST -- and what therefore still sits in the XCD's L2 (4 MB for its 32 workgroups)
    // and in the Infinity Cache (256 MB) -- is what the next launch reads FIRST.  A forward-only walk of a pool set larger
    // than a cache level gets nothing out of an LRU-like level (cyclic access); the ping-pong walk finds up to the
    // level's capacity of it still there.
    const int *tab = next_tile + 2;
    const int nlocal = __builtin_amdgcn_readfirstlane(tab[N_BUCKETS - 1]);
    const bool rev = a.rev != 0;
    int bk = rev ? N_BUCKETS - 1 : 0;
    int cstart = rev ? __builtin_amdgcn_readfirstlane(tab[N_BUCKETS - 2]) : 0;
    int cend = __builtin_amdgcn_readfirstlane(tab[bk]), sfirst = __builtin_amdgcn_readfirstlane(tab[N_BUCKETS + bk]);
    int ticket = wib;
    for (;;) {
        const int i0 = __builtin_amdgcn_readfirstlane(ticket);
        if (i0 >= nlocal) break;
        // the NEXT ticket is drawn before this tile's work, so its LDS round trip (behind the
        // previous tile's scatter atomics) overlaps the tile instead of separating two tiles
        if (lane == 0) ticket = atomicAdd(next_tile, 1);
        const int i = rev ? nlocal - 1 - i0 : i0;
        // (a wave's tickets only grow: its bucket pointer only moves one way)
        while (i >= cend) {
            ++bk; cstart = cend;
            cend = __builtin_amdgcn_readfirstlane(tab[bk]);
            sfirst = __builtin_amdgcn_readfirstlane(tab[N_BUCKETS + bk]);
        }
        while (i < cstart) {
            --bk; cend = cstart;
            cstart = bk ? __builtin_amdgcn_readfirstlane(tab[bk - 1]) : 0;
            sfirst = __builtin_amdgcn_readfirstlane(tab[N_BUCKETS + bk]);
        }
        const int tb = sfirst + (i - cstart);
#ifdef CFMM_PHASE_TIMERS
        const long long tc0 = clock64();
        t_out += tc0 - t_prev;
#endif
        switch (bk) {
        case 0: if constexpr (!STABLE) { tilen<8, WITH_D, DET, BATCH, false, NT, !FLUSH, LNU>(a.bn[5], tb, lane, nu_s, psi_s, diag_s, xs, fsum, bc, nullptr, lnu_s); } break;
        case 1: if constexpr (!STABLE) { tilen<7, WITH_D, DET, BATCH, false, NT, !FLUSH, LNU>(a.bn[4], tb, lane, nu_s, psi_s, diag_s, xs, fsum, bc, nullptr, lnu_s); } break;
        case 2: if constexpr (!STABLE) { tilen<6, WITH_D, DET, BATCH, false, NT, !FLUSH, LNU>(a.bn[3], tb, lane, nu_s, psi_s, diag_s, xs, fsum, bc, nullptr, lnu_s); } break;
        case 3: if constexpr (!STABLE) { tilen<5, WITH_D, DET, BATCH, false, NT, !FLUSH, LNU>(a.bn[2], tb, lane, nu_s, psi_s, diag_s, xs, fsum, bc, nullptr, lnu_s); } break;
        case 4: if constexpr (!STABLE) { tilen<4, WITH_D, DET, BATCH, false, NT, !FLUSH, LNU>(a.bn[1], tb, lane, nu_s, psi_s, diag_s, xs, fsum, bc, nullptr, lnu_s); } break;
        case 5: if constexpr (!STABLE) { tilen<3, WITH_D, DET, BATCH, false, NT, !FLUSH, LNU>(a.bn[0], tb, lane, nu_s, psi_s, diag_s, xs, fsum, bc, nullptr, lnu_s); } break;
        case 6: if constexpr (STABLE) { tile2<3, WITH_D, DET, BATCH, false, NT, !FLUSH>(a.b2[3], (long long)tb * WT_HEAVY, lane, nu_s, psi_s, diag_s, fsum, bc); } break;
        case 7: if constexpr (STABLE) { tile2<4, WITH_D, DET, BATCH, false, NT, !FLUSH>(a.b2[4], (long long)tb * WT_HEAVY, lane, nu_s, psi_s, diag_s, fsum, bc); } break;
        case 8: if constexpr (!STABLE) { tile2<1, WITH_D, DET, BATCH, false, NT, !FLUSH>(a.b2[1], (long long)tb * WT_HEAVY, lane, nu_s, psi_s, diag_s, fsum, bc); } break;
        case 9: if constexpr (!STABLE) {
                    if (NT >= 1 && a.wide) { if constexpr (NT >= 1) tile2<0, WITH_D, DET, BATCH, false, NT, !FLUSH, WT_WIDE / 64>(a.b2[0], (long long)tb * WT_WIDE, lane, nu_s, psi_s, diag_s, fsum, bc); }
                    else tile2<0, WITH_D, DET, BATCH, false, NT, !FLUSH>(a.b2[0], (long long)tb * WT_LIGHT, lane, nu_s, psi_s, diag_s, fsum, bc);
                } break;
        default: if constexpr (!STABLE) { tile2<2, WITH_D, DET, BATCH, false, NT, !FLUSH>(a.b2[2], (long long)tb * WT_LIGHT, lane, nu_s, psi_s, diag_s, fsum, bc); } break;
        }
#ifdef CFMM_PHASE_TIMERS
        if (a.ts && lane == 0) {                       // per-wave tile log: ts[64 + 8 gw + i] = bucket << 48 | cycles
            const int gw = blockIdx.x * (blockDim.x >> 6) + wib;
            t_prev = clock64();
            if (gw < 4096 && nlog < 7) a.ts[64 + 8 * gw + nlog] = ((long long)(bk + 1) << 48) | (t_prev - tc0);
            ++nlog;
        }
#endif
    }
#ifdef CFMM_PHASE_TIMERS
    if (a.ts && lane == 0) {
        const int gw = blockIdx.x * (blockDim.x >> 6) + wib;
        if (gw < 4096) a.ts[64 + 8 * gw + 7] = (15ll << 48) | (t_out + (clock64() - t_prev));
    }
#endif
    }
    PHASE_STAMP(a.ts, 2);
    if constexpr (BATCH) {
        // every live vector's tile into its own accumulator; fpart: [BATCH_MAX][16] wave partials of nu' psi
        __syncthreads();
        const int nwv = blockDim.x >> 6;
        for (unsigned mask = bc.alive; mask; mask &= mask - 1) {
            const int bb = __builtin_ctz(mask);
            double *base = acc_b[bb] + (size_t)(blockIdx.x % a.nslices) * acc_stride(n);
            const double *pt = psi_t + bb * bc.tile_stride, *nub = nu_s + bb * bc.nu_stride;
            double f = 0.0;
            for (int j = threadIdx.x; j < n; j += blockDim.x) {
                const double v = pt[j];
                if (v != 0.0) { unsafeAtomicAdd(&base[j], v); f += nub[j] * v; }
            }
            f = wave_sum(f);
            if (lane == 0) fpart[bb * 16 + wib] = f;
        }
        __syncthreads();
        if ((int)threadIdx.x < BATCH_MAX && ((bc.alive >> threadIdx.x) & 1u)) {
            double f = 0.0;
            for (int w = 0; w < nwv; ++w) f += fpart[threadIdx.x * 16 + w];
            if (f != 0.0) unsafeAtomicAdd(&acc_b[threadIdx.x][(size_t)(blockIdx.x % a.nslices) * acc_stride(n) + acc_arb(n)], f);
        }
        return;
    }
    if constexpr (!FLUSH || DET) {                // (kept in LDS for a consumer in the workgroup: per-wave partials of the carried sum)
        fsum = wave_sum(fsum);
        if (lane == 0) fpart[wib] = fsum;
    }
    __syncthreads();
    PHASE_STAMP(a.ts, 3);
    if constexpr (!FLUSH) return;

    if constexpr (DET) {
        // integer limbs: ONE global accumulator (no slices: the order of these atomics cannot change the sum)
        const unsigned long long *pl = reinterpret_cast<const unsigned long long *>(psi_t), *dl = reinterpret_cast<const unsigned long long *>(diag_t);
        for (int j = threadIdx.x; j < 3 * n; j += blockDim.x) {
            const unsigned long long v = pl[j];
            if (v) atomicAdd(&a.acc_l[j], v);
            if (WITH_D) { const unsigned long long dv = dl[j]; if (dv) atomicAdd(&a.acc_l[3 * n + j], dv); }
        }
        PHASE_STAMP(a.ts, 4);
        return;
    }
    // sum arb = nu' psi, formed here from the workgroup's tile (one fma per token) instead of one per pool and lane in the tiles
    double *base = acc + (size_t)(blockIdx.x % a.nslices) * acc_stride(n);
    double fw = 0.0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const double v = psi_t[j];
        if (v != 0.0) { unsafeAtomicAdd(&base[j], v); fw = fma(nu_s[j], v, fw); }
        if (WITH_D) {
            const double dv = diag_t[j];
            if (dv != 0.0) unsafeAtomicAdd(&base[acc_diag(n) + j], dv);
        }
    }
    fw = wave_sum(fw);
    if (lane == 0) fpart[wib] = fw;
    __syncthreads();
    if (threadIdx.x == 0) {
        double f = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) f += fpart[w];
        if (f != 0.0) unsafeAtomicAdd(&base[acc_arb(n)], f);
    }
    PHASE_STAMP(a.ts, 4);
}

// DMA: the staged tile walk; the launch then carries one 4 KB slot per wave behind the exchange strips
template <bool WITH_D, bool STABLE, bool DET = false, bool DMA = false, int NT = 0>
__global__ void __launch_bounds__(EVAL_THREADS, EVAL_WAVES_PER_SIMD)
eval_kernel(EvalArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    // (the staged walk's slots come FIRST: their LDS addresses travel through M0 and stay below 64 KB that way)
    double *const lds = lds_raw + (DMA ? (STAGE_BYTES / 8) * (EVAL_THREADS / 64) : 0);
    PHASE_STAMP(a.ts, 0);
#ifdef CFMM_PHASE_TIMERS
    if (a.ts && threadIdx.x == 0 && blockIdx.x < 1024) a.ts[64 + 8 * 4096 + 2 * blockIdx.x] = wall_clock64();    // block start
#endif
    const int n = a.n, tile = eval_tile_doubles(n, DET);
    double *psi_s = lds, *diag_s = lds + tile;
    double *nu_s = lds + (WITH_D ? 2 : 1) * tile;       // [n + 1]
    double *fpart = nu_s + n + 2;                       // [16]
    int *next_tile = reinterpret_cast<int *>(fpart + 16);   // the workgroup's tile ticket counter, the tile-range table behind it
    if (threadIdx.x < 64) build_tile_table(a, next_tile, threadIdx.x);
    double2 *xs = reinterpret_cast<double2 *>(lds + eval_lds_doubles(n, WITH_D, DET)) + 64 * (threadIdx.x >> 6);   // wave-private [64]
    // prices and the stop flag arrive in ONE round trip (the flag rides behind the prices)
    constexpr bool LNU = eval_has_lnu(WITH_D, DET) && !STABLE && !DMA;
    double *lnu_s = lds + eval_lnu_offset(n, WITH_D, DET);
    // (the table of log-prices serves the K-asset tiles alone: a network without such pools does not pay for it)
    const bool want_lnu = LNU && (a.bn[0].m | a.bn[1].m | a.bn[2].m | a.bn[3].m | a.bn[4].m | a.bn[5].m) != 0;
    for (int j = threadIdx.x; j <= n; j += blockDim.x) {
        const double v = a.nu[j];
        nu_s[j] = v;
        if (want_lnu && j < n) lnu_s[j] = log_pos(v);
    }
    for (int j = threadIdx.x; j < (WITH_D ? 2 : 1) * tile; j += blockDim.x) lds[j] = 0.0;      // (+0.0 is the all-zero bit pattern: limbs too)
    __syncthreads();
    if (nu_s[n] != 0.0) return;
    PHASE_STAMP(a.ts, 1);
    if constexpr (DMA) {
        const double *stage = lds_raw + (STAGE_BYTES / 8) * (threadIdx.x >> 6);
        eval_tiles_and_flush<WITH_D, STABLE, DET, false, true, true>(a, a.acc, nu_s, psi_s, diag_s, fpart, next_tile, xs, BatchCtl{1u, 0, 0}, nullptr, stage, false);
    } else
    eval_tiles_and_flush<WITH_D, STABLE, DET, false, true, false, NT, LNU>(a, a.acc, nu_s, psi_s, diag_s, fpart, next_tile, xs, BatchCtl{1u, 0, 0}, nullptr, nullptr, false, lnu_s);
#ifdef CFMM_PHASE_TIMERS
    __syncthreads();
    if (a.ts && threadIdx.x == 0 && blockIdx.x < 1024) a.ts[64 + 8 * 4096 + 2 * blockIdx.x + 1] = wall_clock64();    // block end
#endif
}

// ------------------------------------------------------------------------------------------
// The dual evaluation at B price vectors in ONE pass over the pools ("B price vectors per pool read").
// The B solves share the pool set (cfmm_clone) and differ in utility, prices and solver state: vector b's prices come
// from nu[b] (its own stop flag behind them), its psi tile is flushed into acc[b].  Per tile the pool columns are loaded
// once; arithmetic, LDS gathers and scatters are per vector.  LDS: psi[B][n] | nu[B][n + 2] | fpart[B_MAX][16] | ticket
// | strips: 144 KB at B = 8 and 1000 tokens.  No metric (WITH_D) and no stableswap bucket here: the first evaluation
// of every solve (which builds the metric) runs through eval_kernel.
// ------------------------------------------------------------------------------------------
struct BatchArgs {
    const double *nu[BATCH_MAX];
    double *acc[BATCH_MAX];
    int nb, pad;
};
__host__ __device__ inline int batch_nu_stride(int n) { return (n + 3) & ~1; }
// (rounded up to even: the wave-private exchange strips behind it are double2 -- at an odd offset every ds_read/write_b128 of
//  the k-asset tiles is a misaligned access and the batched evaluation takes twice as long: 139 against 68 us at C3, B = 8)
// (-DCFMM_TEST_MISALIGN_STRIPS=1 re-creates that bug on purpose: the variant tools/kernel_budget.py must flag -- the proof that the
//  kernel-time guard of tests/test_gpu_perf.py sees what the parity tests cannot)
#ifndef CFMM_TEST_MISALIGN_STRIPS
#define CFMM_TEST_MISALIGN_STRIPS 0
#endif
__host__ __device__ inline int batch_lds_doubles(int n, int nb)
{
    const int d = (nb * n + nb * batch_nu_stride(n) + BATCH_MAX * 16 + 2 + N_BUCKETS + 1) & ~1;
    return CFMM_TEST_MISALIGN_STRIPS ? d + 1 : d;
}
__host__ __device__ inline size_t batch_lds_bytes(int n, int nb) { return (size_t)(batch_lds_doubles(n, nb) + 2 * 64 * (EVAL_THREADS / 64)) * sizeof(double); }
__host__ __device__ inline int batch_capacity(int n)          // price vectors per launch that fit 160 KB of LDS
{
    int nb = BATCH_MAX;
    while (nb > 1 && batch_lds_bytes(n, nb) > 160 * 1024) --nb;
    return nb;
}

__global__ void __launch_bounds__(EVAL_THREADS, EVAL_WAVES_PER_SIMD)
eval_batch_kernel(EvalArgs a, BatchArgs bt)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = a.n, nb = bt.nb, nus = batch_nu_stride(n);
    double *psi_s = lds;                                // [nb][n]
    double *nu_s = lds + nb * n;                        // [nb][nus]: prices, then the stop flag
    double *fpart = nu_s + nb * nus;                    // [BATCH_MAX][16]
    int *next_tile = reinterpret_cast<int *>(fpart + BATCH_MAX * 16);
    if (threadIdx.x < 64) build_tile_table(a, next_tile, threadIdx.x);
    double2 *xs = reinterpret_cast<double2 *>(lds + batch_lds_doubles(n, nb)) + 64 * (threadIdx.x >> 6);
    for (int b = 0; b < nb; ++b) {
        const double *src = bt.nu[b];
        for (int j = threadIdx.x; j <= n; j += blockDim.x) nu_s[b * nus + j] = src[j];
    }
    for (int j = threadIdx.x; j < nb * n; j += blockDim.x) psi_s[j] = 0.0;
    __syncthreads();
    unsigned alive = 0;
    for (int b = 0; b < nb; ++b) alive |= (nu_s[b * nus + n] == 0.0 ? 1u : 0u) << b;
    alive = __builtin_amdgcn_readfirstlane(alive);
    if (!alive) return;
    const BatchCtl bc{alive, nus, n};
    eval_tiles_and_flush<false, false, false, true>(a, nullptr, nu_s, psi_s, nullptr, fpart, next_tile, xs, bc, bt.acc);
}

// ------------------------------------------------------------------------------------------
// trade materialisation (once per solve): Delta = max(-y,0), Lambda = max(y,0), slot-major
// ------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ void trades2_body(const Bucket2 &b, long long i, const double *__restrict__ nu, double *__restrict__ delta, double *__restrict__ lambda)
{
    const double Ra = b.Ra[i], Rb = b.Rb[i], g = b.fee[i];
    const double pa = nu[b.ia[i]], pb = nu[b.ib[i]];
    Y2 y;
    if (KIND == 0) y = pool_cp2(Ra, Rb, g, pa, pb);
    else if (KIND == 1) y = pool_w2(Ra, Rb, g, b.param[i], pa, pb);
    else if (KIND == 2) { y = pool_sum2(Ra, Rb, g, pa, pb); if (b.flags && b.flags[i]) { y.ya = 0.0; y.yb = 0.0; } }
    else if (KIND == 3) y = pool_curve2(Ra, Rb, g, b.param[i], pa, pb);
    else y = pool_generic2<(KIND >= 4 ? KIND : 4)>(Ra, Rb, g, b.param[i], pa, pb);
    const long long o = b.perm ? b.perm[i] : i;          // (the tenders go out in the caller's pool order)
    delta[o] = fmax(-y.ya, 0.0);  delta[b.m + o] = fmax(-y.yb, 0.0);
    lambda[o] = fmax(y.ya, 0.0);  lambda[b.m + o] = fmax(y.yb, 0.0);
}
template <int KIND>
__global__ void __launch_bounds__(256)
trades2_kernel(Bucket2 b, const double *__restrict__ nu, double *__restrict__ delta, double *__restrict__ lambda)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < b.m) trades2_body<KIND>(b, i, nu, delta, lambda);
}
// the tenders of EVERY point of a sweep (cfmm_solve_sweep) in one launch per bucket: blockIdx.y = the point; its accepted prices sit
// nu_stride doubles apart, its tenders out_stride doubles apart, its tied-pool flags (constant-sum bucket) flags_stride ints apart
template <int KIND>
__global__ void __launch_bounds__(256)
trades2_sweep_kernel(Bucket2 b, const double *__restrict__ nu, int nu_stride, double *__restrict__ delta, double *__restrict__ lambda, long long out_stride,
                     const int *flags_b, int flags_stride)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.m) return;
    const size_t p = blockIdx.y;
    if (KIND == 2) b.flags = flags_b ? flags_b + p * flags_stride : nullptr;
    trades2_body<KIND>(b, i, nu + p * nu_stride, delta + p * out_stride, lambda + p * out_stride);
}

template <int K>
__device__ __forceinline__ void tradesn_body(const BucketN &b, long long i, const double *__restrict__ nu, const double *__restrict__ slo, double *__restrict__ delta, double *__restrict__ lambda)
{
    double R[K], w[K], p[K], y[K];
    int tok[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {                      // device layout is pool-major (leg j of pool i at i*K + j)
        R[j] = b.R[i * K + j];
        w[j] = b.w[i * K + j];
        tok[j] = b.idx[i * K + j];
        p[j] = nu[tok[j]];
    }
    const double g = b.fee[i];
    pool_geomean_n<K>(R, w, g, [&](int j) { return p[j]; }, y);
    if (slo) {
        // low-order log-prices of a second-order solve (smooth.hpp: gn_newton_kernel): the legs that trade move by
        // the pool's exact first-order response,  dy_j = -(w_j e^t / p_j)(dt - s_lo_j),  dt = sum_A w s_lo / sum_A w;
        // w_j e^t / p_j is the traded leg's post-trade reserve (over the fee on the deposit side)
        double wa = 0.0, dt = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) if (y[j] != 0.0) { wa += w[j]; dt += w[j] * slo[tok[j]]; }
        if (wa > 0.0) {
            dt /= wa;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (y[j] == 0.0) continue;
                const double cx = y[j] > 0.0 ? (R[j] - y[j]) : (R[j] - g * y[j]) / g;  // c_j x_j = w_j e^t / p_j: x_j (withdrawn) or x_j / gamma (deposited)
                y[j] -= cx * (dt - slo[tok[j]]);
            }
        }
    }
    const long long o = b.perm ? b.perm[i] : i;
#pragma unroll
    for (int j = 0; j < K; ++j) {                      // results slot-major and in the caller's pool order, as the C-ABI hands them out
        delta[(size_t)j * b.m + o] = fmax(-y[j], 0.0);
        lambda[(size_t)j * b.m + o] = fmax(y[j], 0.0);
    }
}
template <int K>
__global__ void __launch_bounds__(256)
tradesn_kernel(BucketN b, const double *__restrict__ nu, const double *__restrict__ slo, double *__restrict__ delta, double *__restrict__ lambda)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < b.m) tradesn_body<K>(b, i, nu, slo, delta, lambda);
}
template <int K>
__global__ void __launch_bounds__(256)
tradesn_sweep_kernel(BucketN b, const double *__restrict__ nu, int nu_stride, double *__restrict__ delta, double *__restrict__ lambda, long long out_stride)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.m) return;
    const size_t p = blockIdx.y;
    tradesn_body<K>(b, i, nu + p * nu_stride, nullptr, delta + p * out_stride, lambda + p * out_stride);
}

// ------------------------------------------------------------------------------------------
// fold the accumulator slices into slice 0 (used before the RCCL all-reduce and by eval_dual)
// ------------------------------------------------------------------------------------------
__global__ void fold_kernel(double *__restrict__ acc, int n, int nslices, int with_d, const DevState *st)
{
    if (st && st->status != 0) return;
    const int len = with_d ? acc_stride(n) : acc_arb(n) + 1;
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
// reproducible mode: the integer limbs (after the integer all-reduce when pool-sharded) -> fp64 accumulator slice 0, in
// a FIXED order: psi_j and diag_j from their three limbs, sum_i arb_i = nu'psi by a fixed reduction tree (the pools' own
// running sums of arb_i would depend on the tile schedule).  Clears the limbs for the next evaluation.  One workgroup.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double limbs_to_double(unsigned long long l0, unsigned long long l1, unsigned long long l2, double inv_scale)
{
    // value = l2 2^64 + l1 2^32 + l0 with every limb a wrapped SIGNED 64-bit sum: exact in 128-bit integer arithmetic
    const __int128 t = ((__int128)(long long)l2 << 64) + ((__int128)(long long)l1 << 32) + (__int128)(long long)l0;
    const long long hi = (long long)(t >> 64);
    const unsigned long long lo = (unsigned long long)t;
    return ((double)hi * 0x1p64 + (double)lo) * inv_scale;     // two roundings, always the same two
}
__global__ void __launch_bounds__(1024)
det_fold_kernel(unsigned long long *__restrict__ L, const double *__restrict__ nu, double *__restrict__ out, int n, double inv_scale,
                double inv_scale_d, int with_d)
{
    __shared__ double part[1024];
    double f = 0.0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const double psi = limbs_to_double(L[j], L[n + j], L[2 * n + j], inv_scale);
        L[j] = 0; L[n + j] = 0; L[2 * n + j] = 0;
        out[j] = psi;
        f = fma(nu[j], psi, f);
        if (with_d) {
            out[acc_diag(n) + j] = limbs_to_double(L[3 * n + j], L[4 * n + j], L[5 * n + j], inv_scale_d);
            L[3 * n + j] = 0; L[4 * n + j] = 0; L[5 * n + j] = 0;
        }
    }
    part[threadIdx.x] = f;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {                 // fixed tree
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[acc_arb(n)] = part[0];
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
    long long *ts;                    // phase timers (tuning builds only)
    // batched solves (cfmm_solve_batch): the update kernels are launched with one workgroup per solve, workgroup b taking
    // its arguments from batch[b] (device memory); hstat: pinned host word {status << 32 | evals} the host polls
    const UpdArgs *batch;
    unsigned long long *hstat;
    const int *pool_flags;            // sweep launches (tiny.hpp, BATCH): this solve's tied-pool flags of the constant-sum bucket, or null
    // Price ties, ORDERED group sums (round 6): gptr[ng + 1] / gmem[n] = the tokens of every group in ascending order (cfmm_set_ties),
    // tie_tmp[2 n] = scratch for the per-token terms.  The group gradient used to be summed by LDS atomics, i.e. in whatever order the
    // waves arrived: for a group of three or more tokens the sum then differs in its last bits from run to run -- and between the RANKS of
    // a pool-sharded solve, whose replicated update must produce the same bits everywhere (found by the two-rank test of the K-asset
    // constant-sum kinks, which tie longer chains of prices than the two-asset pools did).  Null: the atomics (the one-wave solves).
    const int *gptr, *gmem;
    double *tie_tmp;
};
// the group's sum of the per-token terms the workgroup has just stored (behind a barrier), members in ascending token order
__device__ __forceinline__ void tie_group_sum(const UpdArgs &a, int g, bool with_d, double &s1, double &s2)
{
    s1 = 0.0; s2 = 0.0;
    const int m1 = a.gptr[g + 1];
    for (int m = a.gptr[g]; m < m1; ++m) {
        const int t = a.gmem[m];
        s1 += a.tie_tmp[t];
        if (with_d) s2 += a.tie_tmp[a.n + t];
    }
}
template <bool BATCH>
__device__ __forceinline__ const UpdArgs &upd_args(const UpdArgs &a0)
{
    if constexpr (BATCH) return a0.batch[blockIdx.x]; else return a0;
}
// the same record BY VALUE through scalar loads (constant address space: the host wrote the array before the launch, nothing writes
// it during one).  Through the reference above every field is a vector load from a uniform address -- a VGPR pair per pointer; the
// register-resident update with 4 variables per thread sat at its 256-VGPR cap and spilled 20 B in its batched instantiation
template <bool BATCH>
__device__ __forceinline__ UpdArgs upd_args_scalar(const UpdArgs &a0)
{
    if constexpr (BATCH) {
        typedef const unsigned long long __attribute__((address_space(4))) *ConstWords;
        static_assert(sizeof(UpdArgs) % 8 == 0, "UpdArgs: whole 8-byte words");
        const ConstWords src = (ConstWords)(unsigned long long)(a0.batch + blockIdx.x);
        union { UpdArgs a; unsigned long long w[sizeof(UpdArgs) / 8]; } r;
#pragma unroll
        for (unsigned i = 0; i < sizeof(UpdArgs) / 8; ++i) r.w[i] = src[i];
        return r.a;
    } else return a0;
}
__device__ __forceinline__ void report_progress(const UpdArgs &a, const DevState &st)
{
    if (a.hstat) __hip_atomic_store(a.hstat, ((unsigned long long)(unsigned)st.status << 32) | (unsigned)st.evals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

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

// the body of the generic update: any token count, price ties, any memory.  `lds`: upd_lds_bytes(ng) of scratch.
// Returns the solve's status after the step (the same in every thread).
__device__ __forceinline__ int update_generic_body(const UpdArgs &a, double *lds)
{
    double *q = lds;                         // [ng]
    double *q2 = lds + a.ng;                 // [ng]
    double *scratch = lds + 2 * a.ng;        // [16*8]
    DevState st = load_state(a.st);
    if (st.status != 0) return st.status;
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
            if (st.first) { dg += base[acc_diag(n) + j]; base[acc_diag(n) + j] = 0.0; }
        }
        a.psi_t[j] = psi;
        const double nuj = a.nu[j], hj = a.h[j], cj = a.c[j];
        const int ct = a.ctype[j];
        double rj = psi + hj;
        if (lbfgs::smooth_utility(ct)) {       // a token of the utility table (lbfgs_rules.hpp): conjugate, its maximiser, the Fenchel-Young gap term
            const lbfgs::UtilityTerm ut = lbfgs::utility_term(ct, cj, hj, nuj, psi);
            rj = psi - ut.pstar;
            sums[0] += ut.ubar;
            sums[1] += ut.ubar + nuj * psi - ut.uval;
            maxs[0] = fmax(maxs[0], ut.viol);
            maxs[1] = fmax(maxs[1], fmax(fabs(psi), fabs(ut.pstar)));
            if (st.first) dg += fmax(ut.curv, 0.0);
        } else {
            sums[0] += (nuj - cj) * hj;
            sums[1] += (nuj - cj) * rj;
            maxs[0] = fmax(maxs[0], ct == 0 ? fmax(-rj, 0.0) : (ct == 1 ? fabs(rj) : 0.0));
            maxs[1] = fmax(maxs[1], fmax(fabs(psi), fabs(hj)));
        }
        if (ties && a.gptr) {                // group sums in a fixed order: the terms out, summed per group behind the barrier below
            a.tie_tmp[j] = nuj * rj;
            if (st.first) a.tie_tmp[n + j] = dg;
        } else if (ties) {                   // group sums through LDS (ds_add_f64)
            unsafeAtomicAdd(&q[a.grp[j]], nuj * rj);
            if (st.first) unsafeAtomicAdd(&q2[a.grp[j]], dg);
        } else {
            a.Gs_t[j] = nuj * rj;
            if (st.first) a.Ds[j] = dg;
        }
    }
    double fpools = 0.0;
    if (tid == 0) {
        for (int sl = 0; sl < a.nslices; ++sl) { double *base = a.acc + (size_t)sl * stride; fpools += base[acc_arb(n)]; base[acc_arb(n)] = 0.0; }
    }
    sums[0] += fpools;                       // f_t = sum arb + (nu - c)'h
    block_reduce<2, 2>(sums, maxs, scratch);
    if (ties && a.gptr) {                    // (block_reduce's barriers stand between the terms' stores and these loads)
        for (int r = tid; r < ng; r += nt) {
            double g1, g2;
            tie_group_sum(a, r, st.first != 0, g1, g2);
            a.Gs_t[r] = g1; if (st.first) a.Ds[r] = g2;
        }
        __syncthreads();
    } else if (ties) {
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
        accept = lbfgs::accept(f_t, st.f, a.armijo, dd[0], dd[1]);
    }

    if (!accept) {
        lbfgs::reject(st);
    } else {
        // ---- C. curvature pair, move the accepted point ----------------------------------
        if (!st.first) {
            double *sv = a.S + (size_t)st.head * hist_stride(n), *yv = a.Y + (size_t)st.head * hist_stride(n);
            double t3[3] = {0.0, 0.0, 0.0};
            double dummy[1] = {0.0};
            for (int r = tid; r < ng; r += nt) {
                const double s1 = a.s_t[r] - a.s[r], y1 = a.Gs_t[r] - a.Gs[r];
                sv[r] = s1; yv[r] = y1;
                t3[0] += s1 * y1; t3[1] += s1 * s1; t3[2] += y1 * y1;
            }
            block_reduce<3, 0>(t3, dummy, scratch);
            if (lbfgs::pair_ok(t3[0], t3[1], t3[2])) {
                if (tid == 0) a.rho[st.head] = rcp_nr(t3[0]);
                st.head = (st.head + 1) % M;
                if (st.hist < M) st.hist += 1;
            }
            st.iters += 1;
        }
        for (int r = tid; r < ng; r += nt) { a.s[r] = a.s_t[r]; a.Gs[r] = a.Gs_t[r]; }
        for (int j = tid; j < n; j += nt) { a.psi_acc[j] = a.psi_t[j]; a.nu_acc[j] = a.nu[j]; }
        st.first = 0;
        // value of the projected reduced gradient: sum_r |P(Gs)_r| / max(1,|f|)  (>= gap)
        {
            double pgs[1] = {0.0};
            double dummy[1] = {0.0};
            for (int r = tid; r < ng; r += nt) pgs[0] += lbfgs::pg_entry(a.Gs[r], a.s[r], a.glo[r], a.ghi[r]);
            __syncthreads();
            block_reduce<1, 0>(pgs, dummy, scratch);
            lbfgs::certify(st, f_t, gapv, viol, scale, pgs[0]);
        }
        if (lbfgs::converged(st, a.pg_rule, a.tol_gap, a.tol_infeas)) {
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
                    const double *sv = a.S + (size_t)i * hist_stride(n), *yv = a.Y + (size_t)i * hist_stride(n);
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
                    const double *sv = a.S + (size_t)i * hist_stride(n), *yv = a.Y + (size_t)i * hist_stride(n);
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
            st.t_step = lbfgs::step_cap(dmx[0], a.max_step);
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
    if (tid == 0) { store_state(a.st, st); a.nu[n] = st.status != 0 ? 1.0 : 0.0; report_progress(a, st); }
    return st.status;
}

template <bool BATCH = false>
__global__ void __launch_bounds__(UPD_THREADS)
update_kernel(UpdArgs a0)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    (void)update_generic_body(upd_args<BATCH>(a0), lds);
}

// ------------------------------------------------------------------------------------------
// The nu update, register-resident form (n <= 2048 tokens): the same iteration as update_kernel
// above (and oracle/cfmm_oracle.c:oracle_step).  Every thread owns TWO adjacent variables and
// their whole L-BFGS history in registers; every vector is read with one 16-byte load per thread
// (a wave has at most 63 loads in flight, so 8-byte loads of ~25 vectors x 4 elements stalled the
// old 4-wave form twice); all loads are issued before the first use; every scalar the accept
// test, the stopping rule and the curvature pair need comes out of ONE batched block reduction;
// then 2m + 1 sequential dot products (wave butterflies on DPP / v_permlane*_swap, one LDS
// exchange, one barrier each).  blockDim = 64 * ceil(ceil(n / E) / 64); the instantiations are
// chosen by the register budget (cfmm_hip.hip: launch_update).
// ------------------------------------------------------------------------------------------
constexpr int UPD_E = 2;

struct BlockRed {
    double *scratch;            // [2][NRED][16]
    int parity, wave, lane, nw;
    static constexpr int NRED = 12;
    __device__ __forceinline__ BlockRed(double *s) : scratch(s), parity(0)
    {
        wave = threadIdx.x >> 6; lane = threadIdx.x & 63; nw = blockDim.x >> 6;
    }
    // NS sums followed by NM maxima; result in every thread
    template <int NS, int NM>
    __device__ __forceinline__ void run(double (&v)[NS + NM])
    {
        put<NS, NM>(v, true);
        if (nw == 1) return;
        get<NS, NM>(v);
    }
    // the same in two halves, for callers that have work to put between the wave-level part and the barrier, or whose
    // waves do not all contribute (`mine` false: the wave only keeps the parity; set nw to the number of contributing
    // waves -- the first nw -- before the first use): put = wave butterflies + one LDS row per wave, get = the barrier and the
    // sum over the rows, in every thread
    template <int NS, int NM>
    __device__ __forceinline__ void put(double (&v)[NS + NM], bool mine)
    {
        static_assert(NS + NM <= NRED, "BlockRed: too many values");
        double *sl = scratch + parity * (NRED * 16);
        if (!mine) return;
#pragma unroll
        for (int k = 0; k < NS; ++k) v[k] = wave_allsum(v[k]);
#pragma unroll
        for (int k = NS; k < NS + NM; ++k) v[k] = wave_allmax(v[k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NS + NM; ++k) sl[k * 16 + wave] = v[k];
        }
    }
    // LDS_ONLY: lds_barrier() instead of __syncthreads() (callers with LDS-DMA in flight)
    template <int NS, int NM, bool LDS_ONLY = false>
    __device__ __forceinline__ void get(double (&v)[NS + NM])
    {
        const double *sl = scratch + parity * (NRED * 16);
        parity ^= 1;
        if constexpr (LDS_ONLY) lds_barrier(); else __syncthreads();
#pragma unroll
        for (int k = 0; k < NS; ++k) { double r = 0.0; for (int w = 0; w < nw; ++w) r += sl[k * 16 + w]; v[k] = r; }
#pragma unroll
        for (int k = NS; k < NS + NM; ++k) { double r = sl[k * 16]; for (int w = 1; w < nw; ++w) r = fmax(r, sl[k * 16 + w]); v[k] = r; }
    }
    __device__ __forceinline__ double sum(double x) { double v[1] = {x}; run<1, 0>(v); return v[0]; }
};

// E adjacent doubles / ints starting at element `first` (a multiple of E, E even): 16-byte loads
template <int E>
__device__ __forceinline__ double dotE(const double (&a)[E], const double (&b)[E])
{
    double r = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) r += a[e] * b[e];
    return r;
}
template <int E>
__device__ __forceinline__ void ldv(const double *p, int first, double (&v)[E])
{
#pragma unroll
    for (int k = 0; k < E / 2; ++k) {
        const double2 t = reinterpret_cast<const double2 *>(p + first)[k];
        v[2 * k] = t.x; v[2 * k + 1] = t.y;
    }
}
template <int E>
__device__ __forceinline__ void ldvi(const int *p, int first, int (&v)[E])
{
#pragma unroll
    for (int k = 0; k < E / 2; ++k) {
        const int2 t = reinterpret_cast<const int2 *>(p + first)[k];
        v[2 * k] = t.x; v[2 * k + 1] = t.y;
    }
}
// store E adjacent doubles starting at element `first`, never touching index >= len
template <int E>
__device__ __forceinline__ void stv(double *p, int first, int len, const double (&v)[E])
{
#pragma unroll
    for (int k = 0; k < E / 2; ++k) {
        const int i = first + 2 * k;
        if (i + 1 < len) reinterpret_cast<double2 *>(p + i)[0] = make_double2(v[2 * k], v[2 * k + 1]);
        else if (i < len) p[i] = v[2 * k];
    }
}

// GEN: the utility may hold entries of the utility table (lbfgs_rules.hpp: CFMM_ULOG / CFMM_UQUAD) -- their conjugate, maximiser and
// Fenchel-Young gap term replace the linear-box token's in part A, nothing else changes (round 5: such utilities ran through the generic
// update_kernel alone, ~25 dependent L2 round trips per step: 45 us per evaluation on 5e4 pools where this form takes ~25)
template <int MAXT, int MM, int E, bool BATCH = false, bool GEN = false>          // <= MAXT threads (register budget), <= MM history pairs, E variables per thread
__global__ void __launch_bounds__(MAXT)
update_reg_kernel(UpdArgs a0)
{
    const UpdArgs a = upd_args_scalar<BATCH>(a0);
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const int n = a.n, ng = a.ng, M = a.M;
    const int hs = hist_stride(n);
    // (tuning probe) the kernel may be launched with several identical workgroups: only workgroup 0 stores
    const bool wr = BATCH || blockIdx.x == 0;
    const int nS = wr ? n : 0, ngS = wr ? ng : 0;
    double *q = lds;                         // [ng]
    double *q2 = lds + ng;                   // [ng]
    BlockRed red(lds + 2 * ng);              // [2][12][16]
    const int stride = acc_stride(n);
    const bool ties = (ng != n);
    // this thread's E adjacent variables; threads past the end load from 0 and are masked out
    const int r0 = tid * E;                  // first element (stores)
    const int pr = (r0 < n) ? r0 : 0;        // first element (loads)
    bool gin[E], tin[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { gin[e] = r0 + e < ng; tin[e] = r0 + e < n; }

#ifdef CFMM_PHASE_TIMERS
    const long long c8 = clock64(), w8 = wall_clock64();
#endif
    const DevState *stp = a.st;
    DevState st;
    st = *stp;                                        // (consumed below, after the first batch of loads is in flight)
    // ---- loads, all issued up front (one 16-byte load per vector and thread) ---------------------
    double s[E], s_t[E], Gs[E], d[E], Ds[E], glo[E], ghi[E];
    ldv<E>(a.s, pr, s); ldv<E>(a.s_t, pr, s_t); ldv<E>(a.Gs, pr, Gs); ldv<E>(a.d, pr, d); ldv<E>(a.Ds, pr, Ds);
    ldv<E>(a.glo, pr, glo); ldv<E>(a.ghi, pr, ghi);
    double nuj[E], hj[E], cj[E], offj[E];
    int ct[E], grp[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { offj[e] = 0.0; grp[e] = 0; }
    ldv<E>(a.nu, pr, nuj); ldv<E>(a.h, pr, hj); ldv<E>(a.c, pr, cj); ldvi<E>(a.ctype, pr, ct);
    if (ties) { ldv<E>(a.off, pr, offj); ldvi<E>(a.grp, pr, grp); }
    double psi[E], dg[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { psi[e] = 0.0; dg[e] = 0.0; }
    for (int sl0 = 0; sl0 < a.nslices; sl0 += 4) {     // 4 slices per trip
        double pp[4][E];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int e = 0; e < E; ++e) pp[u][e] = 0.0;
            if (sl0 + u < a.nslices) ldv<E>(a.acc + (size_t)(sl0 + u) * stride, pr, pp[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < E; ++e) psi[e] += pp[u][e];
    }
    if (st.status != 0) return;
#ifdef CFMM_PHASE_TIMERS
    if (a.ts && threadIdx.x == 0) { a.ts[16] = c8; a.ts[17] = w8; }
#endif
    PHASE_STAMP(a.ts, 9);
    if (st.first) {                                    // first evaluation of a solve: the diagonal metric
        for (int sl = 0; sl < a.nslices; ++sl) {
            double t2[E];
            ldv<E>(a.acc + (size_t)sl * stride + acc_diag(n), pr, t2);
#pragma unroll
            for (int e = 0; e < E; ++e) dg[e] += t2[e];
        }
    }
    // history last (it is needed last): ages are resolved to physical slots here, after st has arrived
    double Sx[MM][E], Yx[MM][E], rho_old[MM];
#pragma unroll
    for (int k = 0; k < MM; ++k) {
        const bool have = k < st.hist;
        const int slot = have ? (st.head - 1 - k + 2 * M) % M : 0;
        rho_old[k] = have ? a.rho[slot] : 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) { Sx[k][e] = 0.0; Yx[k][e] = 0.0; }
        if (have) { ldv<E>(a.S + (size_t)slot * hs, pr, Sx[k]); ldv<E>(a.Y + (size_t)slot * hs, pr, Yx[k]); }
#pragma unroll
        for (int e = 0; e < E; ++e) if (!gin[e]) { Sx[k][e] = 0.0; Yx[k][e] = 0.0; }     // the dots run over all E
    }
    double fpools = 0.0;
    if (tid < a.nslices) fpools = a.acc[(size_t)tid * stride + acc_arb(n)];
    // the accumulators are consumed: clear them for the next evaluation
    {
        double zero[E];
#pragma unroll
        for (int e = 0; e < E; ++e) zero[e] = 0.0;
        for (int sl = 0; sl < a.nslices; ++sl) {
            double *base = a.acc + (size_t)sl * stride;
            if (r0 < n) { stv<E>(base, r0, nS, zero); if (st.first) stv<E>(base + acc_diag(n), r0, nS, zero); }
        }
        if (wr && tid < a.nslices) a.acc[(size_t)tid * stride + acc_arb(n)] = 0.0;
    }
    PHASE_STAMP(a.ts, 10);

    // ---- A. residuals, group gradient at the trial point -----------------------------------------
    if (ties) {
        for (int r = tid; r < ng; r += blockDim.x) { q[r] = 0.0; q2[r] = 0.0; }
        __syncthreads();
    }
    double Gs_t[E];
    lbfgs::UtilityTerm ut[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        Gs_t[e] = 0.0;
        ut[e] = lbfgs::UtilityTerm{0.0, 0.0, 0.0, 0.0, 0.0};
        if (tin[e]) {
            double rj = psi[e] + hj[e];
            if (GEN && lbfgs::smooth_utility(ct[e])) {
                ut[e] = lbfgs::utility_term(ct[e], cj[e], hj[e], nuj[e], psi[e]);
                rj = psi[e] - ut[e].pstar;
                if (st.first) dg[e] += fmax(ut[e].curv, 0.0);
            }
            if (!BATCH && ties && a.gptr) {  // (ordered group sums: UpdArgs::gptr; the batched solves take no ties)
                a.tie_tmp[r0 + e] = nuj[e] * rj;
                if (st.first) a.tie_tmp[n + r0 + e] = dg[e];
            } else if (ties) {
                unsafeAtomicAdd(&q[grp[e]], nuj[e] * rj);
                if (st.first) unsafeAtomicAdd(&q2[grp[e]], dg[e]);
            } else {
                Gs_t[e] = nuj[e] * rj;
                if (st.first) Ds[e] = dg[e];
            }
        }
    }
    if (!BATCH && ties && a.gptr) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) if (gin[e]) {
            double g1, g2;
            tie_group_sum(a, r0 + e, st.first != 0, g1, g2);
            Gs_t[e] = g1; if (st.first) Ds[e] = g2;
        }
        __syncthreads();
    } else if (ties) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) if (gin[e]) { Gs_t[e] = q[r0 + e]; if (st.first) Ds[e] = q2[r0 + e]; }
        __syncthreads();
    }
    // ONE batched reduction; the quantities of the accepted branch (pg, |q|^2, act) are computed
    // speculatively at the trial point:
    // 0 f_lin  1 gapv  2 Gs.ds  3 Gs_t.ds  4 s.y  5 s.s  6 y.y  7 pg  8 |q|^2  |  9 viol  10 scale
    double A[11] = {fpools, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double sv[E], yv[E], qv[E];
    bool act[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (tin[e]) {
            if (GEN && lbfgs::smooth_utility(ct[e])) {
                A[0] += ut[e].ubar;
                A[1] += ut[e].ubar + nuj[e] * psi[e] - ut[e].uval;
                A[9] = fmax(A[9], ut[e].viol);
                A[10] = fmax(A[10], fmax(fabs(psi[e]), fabs(ut[e].pstar)));
            } else {
            const double rj = psi[e] + hj[e];
            A[0] += (nuj[e] - cj[e]) * hj[e];
            A[1] += (nuj[e] - cj[e]) * rj;
            A[9] = fmax(A[9], ct[e] == 0 ? fmax(-rj, 0.0) : (ct[e] == 1 ? fabs(rj) : 0.0));
            A[10] = fmax(A[10], fmax(fabs(psi[e]), fabs(hj[e])));
            }
        }
        sv[e] = 0.0; yv[e] = 0.0; qv[e] = 0.0; act[e] = true;
        if (gin[e]) {
            sv[e] = s_t[e] - s[e]; yv[e] = Gs_t[e] - Gs[e];
            A[2] += Gs[e] * sv[e]; A[3] += Gs_t[e] * sv[e];
            A[4] += sv[e] * yv[e]; A[5] += sv[e] * sv[e]; A[6] += yv[e] * yv[e];
            const double G = Gs_t[e], sr = s_t[e];
            double v = G;
            if (glo[e] == ghi[e]) v = 0.0;
            else if (sr <= glo[e] + 1e-14) v = fmin(G, 0.0);
            else if (sr >= ghi[e] - 1e-14) v = fmax(G, 0.0);
            A[7] += fabs(v);
            act[e] = is_active(sr, glo[e], ghi[e], G);
            qv[e] = act[e] ? 0.0 : G;
            A[8] += qv[e] * qv[e];
        }
    }
    red.run<9, 2>(A);
    const double f_t = A[0], gapv = A[1], viol = A[9], scale = A[10];
    st.evals += 1;
    PHASE_STAMP(a.ts, 11);

    // ---- B. accept test ------------------------------------------------------------------------
    bool accept = st.first != 0;
    if (!st.first)
        accept = lbfgs::accept(f_t, st.f, a.armijo, A[2], A[3]);
    PHASE_STAMP(a.ts, 12);

    bool new_dir = false;
    if (!accept) {
        lbfgs::reject(st);
    } else {
        // ---- C. curvature pair, move the accepted point ---------------------------------------
        bool pair_ok = false;
        double rho_new = 0.0;
        if (!st.first) {
            if (r0 < ng) { stv<E>(a.S + (size_t)st.head * hs, r0, ngS, sv); stv<E>(a.Y + (size_t)st.head * hs, r0, ngS, yv); }
            if (lbfgs::pair_ok(A[4], A[5], A[6])) {
                pair_ok = true;
                rho_new = rcp_nr(A[4]);
                if (wr && tid == 0) a.rho[st.head] = rho_new;
                st.head = (st.head + 1) % M;
                if (st.hist < M) st.hist += 1;
            }
            st.iters += 1;
        }
#pragma unroll
        for (int e = 0; e < E; ++e) if (gin[e]) { s[e] = s_t[e]; Gs[e] = Gs_t[e]; }
        if (r0 < ng) { stv<E>(a.s, r0, ngS, s); stv<E>(a.Gs, r0, ngS, Gs); if (st.first) stv<E>(a.Ds, r0, ngS, Ds); }
        if (r0 < n) { stv<E>(a.psi_acc, r0, nS, psi); stv<E>(a.nu_acc, r0, nS, nuj); }
        lbfgs::certify(st, f_t, gapv, viol, scale, A[7]);
        const double gp_sq = A[8];
        const bool was_first = st.first != 0;
        st.first = 0;
        if (lbfgs::converged(st, a.pg_rule, a.tol_gap, a.tol_infeas)) {
            st.status = 1;
        } else {
            PHASE_STAMP(a.ts, 13);
            // ---- D. two-loop recursion with the diagonal metric; pair 0 = the new pair ------------
            new_dir = true;
            // how many of the prefetched (old) pairs are still in the window
            const int old_hist = was_first ? 0 : (pair_ok ? (st.hist - 1) : st.hist);
            double alpha_new = 0.0, alpha[MM];
            if (pair_ok) {
                alpha_new = rho_new * red.sum(dotE<E>(sv, qv));
#pragma unroll
                for (int e = 0; e < E; ++e) qv[e] -= alpha_new * yv[e];
            }
#pragma unroll
            for (int k = 0; k < MM; ++k) {
                alpha[k] = 0.0;
                if (k < old_hist) {
                    alpha[k] = rho_old[k] * red.sum(dotE<E>(Sx[k], qv));
#pragma unroll
                    for (int e = 0; e < E; ++e) qv[e] -= alpha[k] * Yx[k][e];
                }
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double H = Ds[e] + fmax(Gs[e], 0.0);
                qv[e] = (gin[e] && H > 0.0) ? qv[e] / H : 0.0;
            }
#pragma unroll
            for (int k = MM - 1; k >= 0; --k) {
                if (k < old_hist) {
                    const double beta = rho_old[k] * red.sum(dotE<E>(Yx[k], qv));
#pragma unroll
                    for (int e = 0; e < E; ++e) qv[e] += Sx[k][e] * (alpha[k] - beta);
                }
            }
            if (pair_ok) {
                const double beta = rho_new * red.sum(dotE<E>(yv, qv));
#pragma unroll
                for (int e = 0; e < E; ++e) qv[e] += sv[e] * (alpha_new - beta);
            }
            double F[2] = {0.0, 0.0};              // d.G | max |d|
#pragma unroll
            for (int e = 0; e < E; ++e) {
                d[e] = (gin[e] && !act[e]) ? -qv[e] : 0.0;
                F[0] += d[e] * Gs[e]; F[1] = fmax(F[1], fabs(d[e]));
            }
            red.run<1, 1>(F);
            if (!(F[0] < 0.0) && gp_sq > 0.0) {       // not a descent direction: restart from the metric
                st.hist = 0;
                double mx[1] = {0.0};
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const double H = Ds[e] + fmax(Gs[e], 0.0);
                    d[e] = (!gin[e] || act[e] || !(H > 0.0)) ? 0.0 : -Gs[e] / H;
                    mx[0] = fmax(mx[0], fabs(d[e]));
                }
                red.run<0, 1>(mx);
                F[1] = mx[0];
            }
            st.t_step = lbfgs::step_cap(F[1], a.max_step);
        }
    }

    // ---- E. next trial point -----------------------------------------------------------------
    PHASE_STAMP(a.ts, 14);
    if (st.status == 0) {
        double v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            v[e] = s[e] + st.t_step * d[e];
            v[e] = fmax(v[e], glo[e]);
            v[e] = fmin(v[e], ghi[e]);
        }
        if (r0 < ng) { stv<E>(a.s_t, r0, ngS, v); if (new_dir) stv<E>(a.d, r0, ngS, d); }
        double nn[E];
        if (ties) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) if (gin[e]) q[r0 + e] = v[e];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) nn[e] = tin[e] ? exp(q[grp[e]] + offj[e]) : 0.0;
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) nn[e] = exp(v[e]);
        }
        if (r0 < n) stv<E>(a.nu, r0, nS, nn);
        if (st.evals >= a.max_evals) st.status = 3;
    }
    if (wr && tid == 0) { store_state(a.st, st); a.nu[n] = st.status != 0 ? 1.0 : 0.0; report_progress(a, st); }
    PHASE_STAMP(a.ts, 15);
}

// ------------------------------------------------------------------------------------------
// The nu update, "Gram" form (memory <= 4, n <= 2048): identical iteration, but the two-loop
// recursion runs on scalars.  Every dot product it needs,
//     u_k = s_k.q0,  v_k = y_k.(H0 q0),  SY[k][j] = s_k.y_j (k > j),  YHY[k][j] = y_k.H0 y_j,
// is independent of the alphas, so all of them -- together with the scalars of the accept test, the
// stopping rule and the curvature pair -- come out of ONE batched reduction (46 values).  The batch is
// a 6-step reduce-scatter over the wave (v_permlane32/16_swap, DPP row_ror / quad_perm, one
// ds_swizzle step): 63 exchange-adds instead of 46 x 6 butterfly steps, lane l ends up with the
// wave total of value l; two LDS exchanges finish it across waves.  Then
//     alpha_k = rho_k (u_k - sum_{j<k} alpha_j SY[k][j])
//     beta_k  = rho_k (v_k - sum_j alpha_j YHY[k][j] + sum_{j>k} gamma_j SY[j][k]),  gamma = alpha - beta
//     d = -(H0 (q0 - sum_j alpha_j y_j) + sum_k gamma_k s_k)       (pairs newest first, invalid pairs rho = 0)
// replaces 2m + 1 sequential block reductions (0.45 us each) by scalar code.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double dppd_ror8(double v) { return dpp_f64<0x128>(v); }       // lane ^ 8 within a row of 16
__device__ __forceinline__ double swz_xor4(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_swizzle(lo, 0x101F);       // bit mode: and 0x1f, or 0, xor 4
    hi = __builtin_amdgcn_ds_swizzle(hi, 0x101F);
    return __hiloint2double(hi, lo);
}

// v[0..63]: per-lane partials of 64 quantities; returns the wave total of quantity `lane`
__device__ __forceinline__ double wave_reduce_scatter64(double (&v)[64], int lane)
{
    double x, y;
#pragma unroll
    for (int i = 0; i < 32; ++i) {                      // lanes 32..63 take over quantities 32..63
        const int l0 = __double2loint(v[i]), h0 = __double2hiint(v[i]);
        const int l1 = __double2loint(v[i + 32]), h1 = __double2hiint(v[i + 32]);
        const auto a = __builtin_amdgcn_permlane32_swap(l0, l1, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(h0, h1, false, false);
        x = __hiloint2double(b[0], a[0]); y = __hiloint2double(b[1], a[1]);
        v[i] = x + y;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {                      // odd rows take over quantities +16
        const int l0 = __double2loint(v[i]), h0 = __double2hiint(v[i]);
        const int l1 = __double2loint(v[i + 16]), h1 = __double2hiint(v[i + 16]);
        const auto a = __builtin_amdgcn_permlane16_swap(l0, l1, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(h0, h1, false, false);
        x = __hiloint2double(b[0], a[0]); y = __hiloint2double(b[1], a[1]);
        v[i] = x + y;
    }
    const bool b8 = lane & 8, b4 = lane & 4, b2 = lane & 2, b1 = lane & 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {                       // lanes with bit 3 take over +8
        const double keep = b8 ? v[i + 8] : v[i], send = b8 ? v[i] : v[i + 8];
        v[i] = keep + dppd_ror8(send);
    }
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
    {
        const double keep = b1 ? v[1] : v[0], send = b1 ? v[0] : v[1];
        v[0] = keep + dpp_f64<0xB1>(send);
    }
    return v[0];
}

constexpr int GRAM_MM = 4;                  // history pairs kept by the Gram form
constexpr int GRAM_P = GRAM_MM + 1;         // + the new pair

template <int MAXT, int E, bool BATCH = false>
__global__ void __launch_bounds__(MAXT)
update_gram_kernel(UpdArgs a0)
{
    const UpdArgs &a = upd_args<BATCH>(a0);
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int MM = GRAM_MM, P = GRAM_P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int n = a.n, ng = a.ng, M = a.M;
    const int hs = hist_stride(n);
    double *q = lds;                         // [ng]
    double *q2 = lds + ng;                   // [ng]
    double *xw = lds + 2 * ng;               // [16][64] per-wave sums | then [64] totals at xw + 16*64 | [16][2] maxima
    double *xt = xw + 16 * 64;
    double *xm = xt + 64;
    BlockRed red(xm + 32);                   // [2][12][16] for the small closing reduction
    const int stride = acc_stride(n);
    const bool ties = (ng != n);
    const bool wr = BATCH || blockIdx.x == 0;
    const int nS = wr ? n : 0, ngS = wr ? ng : 0;
    const int r0 = tid * E;
    const int pr = (r0 < n) ? r0 : 0;
    bool gin[E], tin[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { gin[e] = r0 + e < ng; tin[e] = r0 + e < n; }

#ifdef CFMM_PHASE_TIMERS
    const long long c8 = clock64(), w8 = wall_clock64();
#endif
    DevState st = load_state(a.st);
    // ---- loads, all issued up front (16-byte loads), first-needed first -------------------------------
    double s[E], s_t[E], Gs[E], d[E], Ds[E], glo[E], ghi[E];
    ldv<E>(a.s, pr, s); ldv<E>(a.s_t, pr, s_t); ldv<E>(a.Gs, pr, Gs); ldv<E>(a.d, pr, d); ldv<E>(a.Ds, pr, Ds);
    ldv<E>(a.glo, pr, glo); ldv<E>(a.ghi, pr, ghi);
    double nuj[E], hj[E], cj[E], offj[E];
    int ct[E], grp[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { offj[e] = 0.0; grp[e] = 0; }
    ldv<E>(a.nu, pr, nuj); ldv<E>(a.h, pr, hj); ldv<E>(a.c, pr, cj); ldvi<E>(a.ctype, pr, ct);
    if (ties) { ldv<E>(a.off, pr, offj); ldvi<E>(a.grp, pr, grp); }
    double psi[E], dg[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { psi[e] = 0.0; dg[e] = 0.0; }
    for (int sl0 = 0; sl0 < a.nslices; sl0 += 4) {
        double pp[4][E];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int e = 0; e < E; ++e) pp[u][e] = 0.0;
            if (sl0 + u < a.nslices) ldv<E>(a.acc + (size_t)(sl0 + u) * stride, pr, pp[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < E; ++e) psi[e] += pp[u][e];
    }
    if (st.status != 0) return;
#ifdef CFMM_PHASE_TIMERS
    if (a.ts && threadIdx.x == 0) { a.ts[16] = c8; a.ts[17] = w8; }
#endif
    PHASE_STAMP(a.ts, 9);
    if (st.first) {
        for (int sl = 0; sl < a.nslices; ++sl) {
            double t2[E];
            ldv<E>(a.acc + (size_t)sl * stride + acc_diag(n), pr, t2);
#pragma unroll
            for (int e = 0; e < E; ++e) dg[e] += t2[e];
        }
    }
    // pairs, newest first: index 0 is the pair this very step may create, 1..MM the stored ones
    double Sx[P][E], Yx[P][E], rho[P];
#pragma unroll
    for (int k = 0; k < MM; ++k) {
        const bool have = k < st.hist;
        const int slot = have ? (st.head - 1 - k + 2 * M) % M : 0;
        rho[k + 1] = have ? a.rho[slot] : 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) { Sx[k + 1][e] = 0.0; Yx[k + 1][e] = 0.0; }
        if (have) { ldv<E>(a.S + (size_t)slot * hs, pr, Sx[k + 1]); ldv<E>(a.Y + (size_t)slot * hs, pr, Yx[k + 1]); }
#pragma unroll
        for (int e = 0; e < E; ++e) if (!gin[e]) { Sx[k + 1][e] = 0.0; Yx[k + 1][e] = 0.0; }
    }
    double fpools = 0.0;
    if (tid < a.nslices) fpools = a.acc[(size_t)tid * stride + acc_arb(n)];
    {
        double zero[E];
#pragma unroll
        for (int e = 0; e < E; ++e) zero[e] = 0.0;
        for (int sl = 0; sl < a.nslices; ++sl) {
            double *base = a.acc + (size_t)sl * stride;
            if (r0 < n) { stv<E>(base, r0, nS, zero); if (st.first) stv<E>(base + acc_diag(n), r0, nS, zero); }
        }
        if (wr && tid < a.nslices) a.acc[(size_t)tid * stride + acc_arb(n)] = 0.0;
    }
    PHASE_STAMP(a.ts, 10);

    // ---- A. group gradient at the trial point ----------------------------------------------------------
    if (ties) {
        for (int r = tid; r < ng; r += blockDim.x) { q[r] = 0.0; q2[r] = 0.0; }
        __syncthreads();
    }
    double Gs_t[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        Gs_t[e] = 0.0;
        if (tin[e]) {
            const double rj = psi[e] + hj[e];
            if (!BATCH && ties && a.gptr) {  // (ordered group sums: UpdArgs::gptr; the batched solves take no ties)
                a.tie_tmp[r0 + e] = nuj[e] * rj;
                if (st.first) a.tie_tmp[n + r0 + e] = dg[e];
            } else if (ties) {
                unsafeAtomicAdd(&q[grp[e]], nuj[e] * rj);
                if (st.first) unsafeAtomicAdd(&q2[grp[e]], dg[e]);
            } else {
                Gs_t[e] = nuj[e] * rj;
                if (st.first) Ds[e] = dg[e];
            }
        }
    }
    if (!BATCH && ties && a.gptr) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) if (gin[e]) {
            double g1, g2;
            tie_group_sum(a, r0 + e, st.first != 0, g1, g2);
            Gs_t[e] = g1; if (st.first) Ds[e] = g2;
        }
        __syncthreads();
    } else if (ties) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) if (gin[e]) { Gs_t[e] = q[r0 + e]; if (st.first) Ds[e] = q2[r0 + e]; }
        __syncthreads();
    }
    // ---- the batch.  0 f_lin 1 gapv 2 Gs.ds 3 Gs_t.ds 4 s.y 5 s.s 6 y.y 7 pg 8 |q0|^2
    //      9..13 u_k | 14..18 v_k | 19..28 SY[k][j], k > j | 29..43 YHY[k][j], k >= j | maxima: viol, scale
    PHASE_STAMP(a.ts, 19);
    double V[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) V[i] = 0.0;
    V[0] = fpools;
    double mx[2] = {0.0, 0.0};
    double q0[E], H0[E];
    bool act[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (tin[e]) {
            const double rj = psi[e] + hj[e];
            V[0] += (nuj[e] - cj[e]) * hj[e];
            V[1] += (nuj[e] - cj[e]) * rj;
            mx[0] = fmax(mx[0], ct[e] == 0 ? fmax(-rj, 0.0) : (ct[e] == 1 ? fabs(rj) : 0.0));
            mx[1] = fmax(mx[1], fmax(fabs(psi[e]), fabs(hj[e])));
        }
        Sx[0][e] = 0.0; Yx[0][e] = 0.0; q0[e] = 0.0; H0[e] = 0.0; act[e] = true;
        if (gin[e]) {
            Sx[0][e] = s_t[e] - s[e]; Yx[0][e] = Gs_t[e] - Gs[e];
            V[2] += Gs[e] * Sx[0][e]; V[3] += Gs_t[e] * Sx[0][e];
            V[4] += Sx[0][e] * Yx[0][e]; V[5] += Sx[0][e] * Sx[0][e]; V[6] += Yx[0][e] * Yx[0][e];
            const double G = Gs_t[e], sr = s_t[e];
            double v = G;
            if (glo[e] == ghi[e]) v = 0.0;
            else if (sr <= glo[e] + 1e-14) v = fmin(G, 0.0);
            else if (sr >= ghi[e] - 1e-14) v = fmax(G, 0.0);
            V[7] += fabs(v);
            act[e] = is_active(sr, glo[e], ghi[e], G);
            q0[e] = act[e] ? 0.0 : G;
            V[8] += q0[e] * q0[e];
            const double H = Ds[e] + fmax(G, 0.0);
            H0[e] = H > 0.0 ? rcp_nr(H) : 0.0;
        }
    }
    if (st.first) {                               // no pair yet on the very first step
#pragma unroll
        for (int e = 0; e < E; ++e) { Sx[0][e] = 0.0; Yx[0][e] = 0.0; }
        V[2] = V[3] = V[4] = V[5] = V[6] = 0.0;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double hq = H0[e] * q0[e];
        int c = 19, c2 = 29;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            V[9 + k] += Sx[k][e] * q0[e];
            V[14 + k] += Yx[k][e] * hq;
            const double hy = H0[e] * Yx[k][e];
#pragma unroll
            for (int j = 0; j < k; ++j) V[c++] += Sx[k][e] * Yx[j][e];
#pragma unroll
            for (int j = 0; j <= k; ++j) V[c2++] += hy * Yx[j][e];
        }
    }
    PHASE_STAMP(a.ts, 20);
    const double mine = wave_reduce_scatter64(V, lane);
    PHASE_STAMP(a.ts, 21);
    mx[0] = wave_allmax(mx[0]); mx[1] = wave_allmax(mx[1]);
    xw[wave * 64 + lane] = mine;
    if (lane == 0) { xm[wave * 2] = mx[0]; xm[wave * 2 + 1] = mx[1]; }
    __syncthreads();
    if (wave == 0) {
        double t = 0.0;
        for (int w = 0; w < nw; ++w) t += xw[w * 64 + lane];
        xt[lane] = t;
    }
    __syncthreads();
    double T[44];
#pragma unroll
    for (int i = 0; i < 44; ++i) T[i] = xt[i];
    PHASE_STAMP(a.ts, 22);
    double viol = 0.0, scale = 0.0;
    for (int w = 0; w < nw; ++w) { viol = fmax(viol, xm[w * 2]); scale = fmax(scale, xm[w * 2 + 1]); }
    const double f_t = T[0], gapv = T[1];
    st.evals += 1;
    PHASE_STAMP(a.ts, 11);

    // ---- B. accept test ---------------------------------------------------------------------------------
    bool accept = st.first != 0;
    if (!st.first)
        accept = lbfgs::accept(f_t, st.f, a.armijo, T[2], T[3]);
    PHASE_STAMP(a.ts, 12);

    bool new_dir = false;
    if (!accept) {
        lbfgs::reject(st);
    } else {
        // ---- C. curvature pair, move the accepted point ------------------------------------------------
        bool pair_ok = false;
        const int old_hist0 = st.hist;
        if (!st.first) {
            if (r0 < ng) { stv<E>(a.S + (size_t)st.head * hs, r0, ngS, Sx[0]); stv<E>(a.Y + (size_t)st.head * hs, r0, ngS, Yx[0]); }
            if (lbfgs::pair_ok(T[4], T[5], T[6])) {
                pair_ok = true;
                if (wr && tid == 0) a.rho[st.head] = rcp_nr(T[4]);
                st.head = (st.head + 1) % M;
                if (st.hist < M) st.hist += 1;
            }
            st.iters += 1;
        }
#pragma unroll
        for (int e = 0; e < E; ++e) if (gin[e]) { s[e] = s_t[e]; Gs[e] = Gs_t[e]; }
        if (r0 < ng) { stv<E>(a.s, r0, ngS, s); stv<E>(a.Gs, r0, ngS, Gs); if (st.first) stv<E>(a.Ds, r0, ngS, Ds); }
        if (r0 < n) { stv<E>(a.psi_acc, r0, nS, psi); stv<E>(a.nu_acc, r0, nS, nuj); }
        lbfgs::certify(st, f_t, gapv, viol, scale, T[7]);
        const double gp_sq = T[8];
        const bool was_first = st.first != 0;
        st.first = 0;
        if (lbfgs::converged(st, a.pg_rule, a.tol_gap, a.tol_infeas)) {
            st.status = 1;
        } else {
            PHASE_STAMP(a.ts, 13);
            // ---- D. the two-loop recursion on scalars ---------------------------------------------------
            new_dir = true;
            // which pairs are in the window: the new one if it passed, then the newest stored ones
            const int keep_old = lbfgs::keep_old(was_first, pair_ok, old_hist0, M);
            rho[0] = pair_ok ? rcp_nr(T[4]) : 0.0;
#pragma unroll
            for (int k = 1; k < P; ++k) if (k - 1 >= keep_old) rho[k] = 0.0;
            double al[P], ga[P];
            lbfgs::gram_two_loop<P>(rho, [&](int k) { return T[9 + k]; }, [&](int k) { return T[14 + k]; },
                                    [&](int k, int j) { return T[19 + k * (k - 1) / 2 + j]; },
                                    [&](int k, int j) { return T[29 + k * (k + 1) / 2 + j]; }, al, ga);
            double F[2] = {0.0, 0.0};              // d.G | max |d|
#pragma unroll
            for (int e = 0; e < E; ++e) {
                double qm = q0[e], rs = 0.0;
#pragma unroll
                for (int k = 0; k < P; ++k) { qm -= al[k] * Yx[k][e]; rs += ga[k] * Sx[k][e]; }
                d[e] = (gin[e] && !act[e]) ? -(H0[e] * qm + rs) : 0.0;
                F[0] += d[e] * Gs[e]; F[1] = fmax(F[1], fabs(d[e]));
            }
            red.run<1, 1>(F);
            if (!(F[0] < 0.0) && gp_sq > 0.0) {       // not a descent direction: restart from the metric
                st.hist = 0;
                double m1[1] = {0.0};
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    d[e] = (!gin[e] || act[e]) ? 0.0 : -Gs[e] * H0[e];
                    m1[0] = fmax(m1[0], fabs(d[e]));
                }
                red.run<0, 1>(m1);
                F[1] = m1[0];
            }
            st.t_step = lbfgs::step_cap(F[1], a.max_step);
        }
    }

    // ---- E. next trial point ----------------------------------------------------------------------------
    PHASE_STAMP(a.ts, 14);
    if (st.status == 0) {
        double v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            v[e] = s[e] + st.t_step * d[e];
            v[e] = fmax(v[e], glo[e]);
            v[e] = fmin(v[e], ghi[e]);
        }
        if (r0 < ng) { stv<E>(a.s_t, r0, ngS, v); if (new_dir) stv<E>(a.d, r0, ngS, d); }
        double nn[E];
        if (ties) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) if (gin[e]) q[r0 + e] = v[e];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) nn[e] = tin[e] ? exp(q[grp[e]] + offj[e]) : 0.0;
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) nn[e] = exp(v[e]);
        }
        if (r0 < n) stv<E>(a.nu, r0, nS, nn);
        if (st.evals >= a.max_evals) st.status = 3;
    }
    if (wr && tid == 0) { store_state(a.st, st); a.nu[n] = st.status != 0 ? 1.0 : 0.0; report_progress(a, st); }
    PHASE_STAMP(a.ts, 15);
}

// ------------------------------------------------------------------------------------------
// self-test of the cross-lane primitives the update kernels rest on (cfmm_selftest): the butterflies
// must leave the wave total / maximum in every lane, the reduce-scatter the total of quantity l in
// lane l.  Inputs are small integers, so every sum is exact and the comparison is bitwise.
// out[0] = number of mismatching lanes (0 = pass).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
selftest_kernel(int *out)
{
    const int lane = threadIdx.x;
    int bad = 0;
    const double x = (double)((lane * 37 + 11) % 101) - 50.0;
    double ts = 0.0, tm = -1e300;
    for (int l = 0; l < 64; ++l) { const double v = (double)((l * 37 + 11) % 101) - 50.0; ts += v; tm = fmax(tm, v); }
    if (wave_allsum(x) != ts) ++bad;
    if (wave_allmax(x) != tm) ++bad;
    double V[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) V[i] = (double)(((lane + 3) * (i + 5)) % 23) - 7.0;
    double want = 0.0;
    for (int l = 0; l < 64; ++l) want += (double)(((l + 3) * (lane + 5)) % 23) - 7.0;      // total of quantity `lane`
    if (wave_reduce_scatter64(V, lane) != want) ++bad;
    atomicAdd(out, bad);
}

// self-test of the generic exact pool (phi2.hpp): pool_generic2 -- the root search on the table entry's L', started from
// D = 0, no closed form used -- against the hand-tuned closed forms of the constant-product, weighted and stableswap kinds
// on 64 x 48 random pools and prices from 30 % below to 30 % above the pools' own; and the power-sum entry against ITS
// closed form (which the exact path never uses).  Counts tenders more than 1e-11 of the reserve apart.
__global__ void __launch_bounds__(64)
selftest_generic_kernel(int *out)
{
    int bad = 0;
    unsigned long long st = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1);
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) * (1.0 / 9007199254740992.0); };
    for (int k = 0; k < 48; ++k) {
        const double Ra = exp(10.0 * rnd() - 2.0), Rb = exp(10.0 * rnd() - 2.0);
        const double g = 1.0 - 0.01 * rnd();
        const double pa = exp(0.6 * rnd() - 0.3), wa = 0.1 + 0.8 * rnd(), t = 0.05 + 0.9 * rnd();
        auto cmp = [&](const Y2 &a, const Y2 &b) {
            if (!(fabs(a.ya - b.ya) <= 1e-11 * Ra && fabs(a.yb - b.yb) <= 1e-11 * Rb)) ++bad;
        };
        {   // constant product: the pool's own price ratio is Rb / Ra ... (pa / pb = Rb / Ra at no trade)
            const double pb = pa * Ra / Rb * exp(0.6 * rnd() - 0.3);
            cmp(pool_generic2<0>(Ra, Rb, g, 0.0, pa, pb), pool_cp2(Ra, Rb, g, pa, pb));
        }
        {   // weighted: pa / pb = (wa / wb) Rb / Ra
            const double pb = pa * (1.0 - wa) / wa * Ra / Rb * exp(0.6 * rnd() - 0.3);
            cmp(pool_generic2<1>(Ra, Rb, g, wa, pa, pb), pool_w2(Ra, Rb, g, wa, pa, pb));
        }
        {   // stableswap near its peg
            const double Rb2 = Ra * exp(0.4 * rnd() - 0.2), al = Ra * Ra * Rb2 * (0.05 + rnd());
            const double pb = pa * exp(0.02 * rnd() - 0.01);
            const Y2 a = pool_generic2<3>(Ra, Rb2, g, al, pa, pb), b = pool_curve2(Ra, Rb2, g, al, pa, pb);
            if (!(fabs(a.ya - b.ya) <= 1e-10 * Ra && fabs(a.yb - b.yb) <= 1e-10 * Rb2)) ++bad;
        }
        {   // power sum: pa / pb = (Rb / Ra)^t; closed form: (y / x)^t = nu_in / (gamma nu_out) on the level set
            const double pb = pa * exp(t * log(Ra / Rb)) * exp(0.6 * rnd() - 0.3);
            const Y2 a = pool_generic2<4>(Ra, Rb, g, t, pa, pb);
            Y2 b; b.ya = 0.0; b.yb = 0.0;
            const double q = 1.0 - t, K = exp(q * log(Ra)) + exp(q * log(Rb));
            auto dir = [&](double Rin, double Rout, double ni, double no, double &yin, double &yout) {
                if (!(no * g * exp(t * log(Rout / Rin)) > ni)) return false;
                const double rho = ni / (g * no);
                const double x = exp(log(K / (1.0 + exp((q / t) * log(rho)))) / q), y = x * exp(log(rho) / t);
                yin = -(x - Rin) / g; yout = Rout - y;
                return true;
            };
            if (!dir(Ra, Rb, pa, pb, b.ya, b.yb)) dir(Rb, Ra, pb, pa, b.yb, b.ya);
            if (!(fabs(a.ya - b.ya) <= 1e-10 * Ra && fabs(a.yb - b.yb) <= 1e-10 * Rb)) ++bad;
        }
    }
    atomicAdd(out, bad);
}

// self-test of log_pos (pool_math.hpp) against the library logarithm: 64 x 256 arguments from 1e-300 to 1e300, dense around 1
// (where log changes sign and the argument reduction switches octave); counts results more than 2 ulp apart
__global__ void __launch_bounds__(64)
selftest_log_kernel(int *out)
{
    int bad = 0;
    for (int k = 0; k < 256; ++k) {
        const int i = k * 64 + threadIdx.x;                          // 0 .. 16383
        double x;
        if (i < 8192) x = exp((i - 4096) * (690.0 / 4096.0));         // e^-690 .. e^690
        else x = 1.0 + (i - 12288) * (1.0 / 8192.0) * ((i & 1) ? 1.0 : 1e-6);   // around 1, coarse and fine
        if (!(x > 0.0)) continue;
        const double a = log_pos(x), b = log(x);
        const double ulp = fmax(fabs(b), 1e-300) * 2.220446049250313e-16;
        if (!(fabs(a - b) <= 2.0 * ulp + 1e-320)) ++bad;
    }
    if (bad) atomicAdd(out, bad);
}

// start of a solve: group variable = mean over members of (log nu0_j - off_j), clamped
template <bool BATCH = false>
__global__ void __launch_bounds__(UPD_THREADS)
start_kernel(UpdArgs a0, const double *nu0_, double *zero, long long nzero, DevState *st_clear, int n_clear)
{
    // (batched: workgroup b starts solve b from ITS previous / given prices a.nu_acc, and clears its own accumulator)
    const UpdArgs &a = upd_args<BATCH>(a0);
    const double *nu0 = BATCH ? a.nu_acc : nu0_;
    if (BATCH) { zero = a.acc; }
    // (nu0 may be a.nu_acc itself: every element is read before the first one is written.  `zero` / `st_clear`: the
    //  accumulator sets and the spare solver-state records of the one-launch iteration, cleared here instead of by
    //  separate fill operations on the stream)
    for (long long j = threadIdx.x; j < nzero; j += blockDim.x) zero[j] = 0.0;
    if ((int)threadIdx.x < n_clear) { DevState z = {}; st_clear[threadIdx.x] = z; }
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *sum = lds, *cnt = lds + a.ng;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n = a.n, ng = a.ng;
    for (int r = tid; r < ng; r += nt) { sum[r] = 0.0; cnt[r] = 0.0; }
    __syncthreads();
    if (a.gptr && ng != n) {                 // (ordered: the mean log-price of a group the same bits on every rank, UpdArgs::gptr)
        for (int r = tid; r < ng; r += nt) {
            double v = 0.0;
            const int m1 = a.gptr[r + 1];
            for (int m = a.gptr[r]; m < m1; ++m) { const int t = a.gmem[m]; v += log(nu0[t]) - a.off[t]; }
            sum[r] = v; cnt[r] = (double)(m1 - a.gptr[r]);
        }
    } else
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
        DevState st = {};
        st.status = 0; st.evals = 0; st.iters = 0; st.first = 1; st.hist = 0; st.head = 0; st.nrej = 0; st.pad = 0;
        st.f = 0.0; st.t_step = 1.0; st.gap = 0.0; st.infeas = 0.0; st.primal = 0.0; st.pg = 0.0;
        *a.st = st;
        a.nu[n] = 0.0;
        report_progress(a, st);
    }
}

}  // namespace cfmm
