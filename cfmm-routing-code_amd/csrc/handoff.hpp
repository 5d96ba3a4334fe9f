// Host <-> device hand-off of the second-order loop's small vectors (round 4).
//
// The loop is host-driven (second-order iteration): per Newton step a few vectors of n doubles go down
// (prices, the low-order log-prices, [diagonal | right-hand side], the pin mask), a few buffers are zeroed (the smoothed
// evaluation's output, the 8.6 MB Hessian) and a vector comes back.  Done with hipMemcpyAsync / hipMemsetAsync every one of
// these is its own blit or fill dispatch: the rocprofv3 trace of a config-5 solve (profiles/r04_*C5newton*) shows ~70 copy and
// ~45 fill kernels per solve with 5-7 us of idle device between any two of them, and ~19 us from the end of a device-to-host
// copy to the start of the next host-to-device one (stream synchronisation, the host's part, two API calls).
//
// Here ONE launch runs a whole list of such jobs: the source of a copy may be pinned host memory (read over the bus by the
// kernel: a few KB), the destination too -- and a launch that ends a hand-off writes a sequence number into pinned memory
// behind its data (system-scope release), which the host polls instead of synchronising the stream.
#pragma once
#include <hip/hip_runtime.h>

namespace cfmm {

constexpr int IO_MAX_JOBS = 6;
struct IoJob { void *dst; const void *src; unsigned long long bytes; };        // src == nullptr: zero-fill.  bytes: a multiple of 8
struct IoArgs {
    IoJob j[IO_MAX_JOBS];
    int njobs;
    unsigned long long *flag;          // pinned; written (= seq) behind every job's data.  Needs a ONE-workgroup launch
    unsigned long long seq;
};

// Jobs run in list order with the SAME element -> thread mapping: a later job may overwrite an earlier job's source
// (copy the accumulator out, then zero it) as long as both address it from the same base.
__global__ __launch_bounds__(256) void io_kernel(IoArgs a)
{
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    for (int q = 0; q < a.njobs; ++q) {
        const IoJob jb = a.j[q];
        const bool wide = (((size_t)jb.dst | (size_t)jb.src | (size_t)jb.bytes) & 15) == 0;
        if (wide) {
            double2 *d = (double2 *)jb.dst; const double2 *s = (const double2 *)jb.src;
            const size_t m = jb.bytes / 16;
            if (s) for (size_t i = tid; i < m; i += nth) d[i] = s[i];
            else for (size_t i = tid; i < m; i += nth) d[i] = make_double2(0.0, 0.0);
        } else {
            double *d = (double *)jb.dst; const double *s = (const double *)jb.src;
            const size_t m = jb.bytes / 8;
            if (s) for (size_t i = tid; i < m; i += nth) d[i] = s[i];
            else for (size_t i = tid; i < m; i += nth) d[i] = 0.0;
        }
    }
    if (a.flag) {                      // (one workgroup: every thread's stores are out system-wide before the flag)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- the shader-clock probe (cfmm_clock_probe_start, bench.py: roofline.effective_clock_ghz_live) -------------------------
// ONE wave on a stream of its own.  It sleeps (s_sleep: no vector work, no memory traffic) and every `period` ticks of the constant
// 100 MHz counter stores one sample {s_memtime = shader cycles, s_memrealtime = 100 MHz ticks} into mapped pinned memory; the
// ratio of the two differences between any two samples is the clock the shader engines ran at in between -- measured WHILE
// the solves run beside it, which a probe in front of or behind them cannot give (the chip clocks to its power budget).
// In front of the first sample the wave runs a chain of PROBE_CHAIN dependent v_fma_f64 between two stamp pairs: cycles per link is a
// property of the pipeline (the same on every box), so the chain shows that s_memtime counts SHADER cycles and not a fixed reference.
// ring: [cap][2] samples | [2 cap] stop word (the host sets it) | [2 cap + 1] samples written so far | [2 cap + 2 .. 4] the chain:
//       shader cycles, 100 MHz ticks, links
constexpr int PROBE_CAP = 8192;
constexpr int PROBE_CHAIN = 16384;
__global__ void __launch_bounds__(64) clock_probe_kernel(long long *ring, int cap, long long period, long long max_ticks)
{
    if (threadIdx.x != 0) return;
    {
        double x = (double)period * 1e-30, y = 1.0 + (double)cap * 1e-30;
        const long long c0 = clock64(), t0 = wall_clock64();
#pragma unroll 16
        for (int i = 0; i < PROBE_CHAIN; ++i) x = __builtin_fma(x, y, 1e-30);
        asm volatile("" :: "v"(x));
        const long long c1 = clock64(), t1 = wall_clock64();
        ring[2 * cap + 2] = c1 - c0; ring[2 * cap + 3] = t1 - t0; ring[2 * cap + 4] = PROBE_CHAIN;
    }
    const long long w0 = wall_clock64();
    long long next = w0, head = 0;
    for (;;) {
        const long long w = wall_clock64();
        if (w >= next) {
            const long long c = clock64(), w2 = wall_clock64();     // (back to back: their skew is the same in every sample and cancels in the differences)
            __hip_atomic_store(ring + 2 * head, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(ring + 2 * head + 1, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            ++head;
            __hip_atomic_store(ring + 2 * cap + 1, head, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            next += period;
            if (head >= cap || w2 - w0 >= max_ticks) break;
            if (__hip_atomic_load(ring + 2 * cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;      // (one bus read per sample, not per wake-up)
        }
        __builtin_amdgcn_s_sleep(16);                                // (~1000 shader cycles)
    }
}

// ---- measurement: what a hand-off between workgroups of ONE XCD costs (cfmm_time_xcd_handoff) -----------------
// The candidate: run iter_kernel's nu update on one workgroup per XCD and let the other 31 workgroups of that XCD pick the trial prices
// up from its L2 instead of repeating the update.  The followers wait for the publisher's WHOLE chain either way, so what the scheme
// costs on the launch's critical path is exactly this hand-off: `np` doubles stored, acknowledged by L2, a flag stored behind them, the
// flag seen by a spinning follower, the doubles loaded into its LDS.  Workgroups 0 .. 7 publish (they wait `delay` ticks first: in the
// real kernel the followers are already spinning when the update ends), all others follow the publisher of their own XCD (HW_REG_XCC_ID).
// stamps[4 b ..]: XCC id | publisher: data-ready tick, flag-stored tick | follower: flag-seen tick, data-in-LDS tick | wrong values seen
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15; }      // HW_REG_XCC_ID[3:0]
__global__ void __launch_bounds__(1024) xcd_handoff_kernel(double *pub, unsigned long long *flag, unsigned long long epoch, int np, long long delay, long long *stamps)
{
    extern __shared__ __attribute__((aligned(16))) double hl[];
    __shared__ int ok_s;
    const int tid = threadIdx.x, xcc = xcc_id();
    long long *st = stamps + 4 * blockIdx.x;
    if (blockIdx.x < 8) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < delay) __builtin_amdgcn_s_sleep(8);
        __syncthreads();
        const long long t_ready = wall_clock64();
        for (int j = tid; j < np; j += blockDim.x) pub[(size_t)xcc * np + j] = (double)epoch + (double)j;
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): this thread's stores are in L2
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(flag + 16 * xcc, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st[0] = xcc; st[1] = t_ready; st[2] = wall_clock64(); st[3] = 0;
        }
        return;
    }
    if (tid == 0) {
        int budget = 1 << 18;
        while (__hip_atomic_load(flag + 16 * xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch && --budget > 0) __builtin_amdgcn_s_sleep(1);
        ok_s = budget > 0;
        st[0] = xcc; st[1] = wall_clock64();
    }
    __syncthreads();
    int bad = 0;
    if (ok_s) {
        for (int j = tid; j < np; j += blockDim.x) hl[j] = pub[(size_t)xcc * np + j];
        __syncthreads();
        for (int j = tid; j < np; j += blockDim.x) bad += hl[j] != (double)epoch + (double)j;
    }
    bad = __syncthreads_count(bad != 0);
    if (tid == 0) { st[2] = wall_clock64(); st[3] = ok_s ? bad : -1; }
}

}  // namespace cfmm
