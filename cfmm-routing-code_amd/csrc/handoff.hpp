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

}  // namespace cfmm
