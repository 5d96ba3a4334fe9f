// One-shot all-reduce over the xGMI mesh for the 8-50 KB messages of the outer iteration.
//
// The pool-sharded iteration exchanges ONE small vector per dual evaluation: [psi | sum arb] (n + 1 doubles, 8-16 KB),
// or the 3 n integer limbs of the reproducible mode.  At that size a ring all-reduce is pure latency (2 (R - 1)
// dependent hops); MI355X's xGMI links are point to point and every GPU of a node reaches every other directly, so the
// whole exchange can be ONE hop: every rank stores its vector straight into a mailbox in each peer's HBM, raises a
// flag there, waits for the R flags in its own mailbox and sums the R vectors itself -- in RANK ORDER, so that every
// rank computes the same bits (no broadcast needed, and fp64 results do not depend on a ring's reduction order).
//
//   mailbox of a rank (device memory, exported to the peers with hipIpcGetMemHandle, one process per GPU):
//       flag[2][ONESHOT_MAX_RANKS]            epoch of the last vector rank r delivered into parity slot p
//       poison                                non-zero: some rank's wait has expired (sticky, see below)
//       slot[2][ONESHOT_MAX_RANKS][cap]       the vectors, 8-byte elements
//   Two parity slots: a rank can run at most one all-reduce ahead of a peer (it needs the peer's vector of epoch e + 1
//   to finish e + 1, and the peer sends that only after it has finished reading epoch e), so epoch e + 1 never lands on
//   data of epoch e still being read.
// All mailbox traffic uses system-scope relaxed atomics (write-through / cache-bypassing 8-byte accesses) with a
// system-scope fence between payload and flag: nothing of it may linger in an L2 that the other GPU cannot see.
// One workgroup per rank: the kernels of all ranks must be running at the same time for the exchange to complete, and one
// workgroup per GPU always is.  The spin is bounded; on expiry the result is poisoned with NaN (the solve then ends with
// CFMM_E_NUMERIC instead of hanging) AND the expiry is published: the rank raises the poison word in EVERY mailbox, every
// later exchange of every rank -- and every wait in progress -- sees it and returns NaN too, so a rank skew beyond the
// bound ends as one collective error on all ranks instead of one rank failing while its peers carry on with valid data
// and hang on the next exchange.  The poison stays until the mailboxes are attached again.
//
// RCCL stays the default and the checker (CFMM_ALLREDUCE=oneshot, or cfmm_oneshot_import, selects this path).
#pragma once
#include <hip/hip_runtime.h>

namespace cfmm {

constexpr int ONESHOT_MAX_RANKS = 16;
constexpr int ONESHOT_THREADS = 1024;
enum { ONESHOT_SUM_F64 = 0, ONESHOT_SUM_I64 = 1, ONESHOT_MAX_F64 = 2 };

__host__ __device__ inline size_t oneshot_flag_words() { return 2 * ONESHOT_MAX_RANKS + 8; }     // flags | poison word (+ padding: slots stay 64-byte aligned)
__host__ __device__ inline size_t oneshot_poison_word() { return 2 * ONESHOT_MAX_RANKS; }
__host__ __device__ inline size_t oneshot_bytes(size_t cap) { return (oneshot_flag_words() + 2 * (size_t)ONESHOT_MAX_RANKS * cap) * 8; }
// (+ one private word behind the mailbox: the rank's epoch counter)
__host__ __device__ inline size_t oneshot_alloc_bytes(size_t cap) { return oneshot_bytes(cap) + 64; }

struct OneShotArgs {
    unsigned long long *mail[ONESHOT_MAX_RANKS];    // every rank's mailbox (mail[rank] is the local one)
    unsigned long long *buf;                        // the vector, reduced in place
    unsigned long long *epoch_word;                 // this rank's count of exchanges done so far (device memory, private)
    const int *stop;                                // optional: a solve's status word; non-zero = the solve has ended, NO exchange
    int n_ranks, rank, count, op;
    size_t cap;
    int nslices;                                    // > 1 (sums of doubles only): buf holds `nslices` accumulator slices `stride`
    long long stride;                               //   elements apart; they are folded into slice 0 on the way out (no fold launch)
};
// The epoch is counted on the DEVICE, by the exchanges that actually happen: with `stop` set, launches enqueued behind the
// end of a solve return at once (every rank holds the same status at the same launch index, so they all skip the same
// exchanges), which lets every rank keep its own number of launches in flight beyond the end -- the host-side run-ahead
// of the single-GPU path -- without the ranks' epochs drifting apart.

__device__ __forceinline__ void sys_store(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned long long sys_load(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

__global__ void __launch_bounds__(ONESHOT_THREADS)
oneshot_allreduce_kernel(OneShotArgs a)
{
    if (a.stop && *a.stop != 0) return;
    const int tid = threadIdx.x, R = a.n_ranks;
    const unsigned long long epoch = *a.epoch_word + 1;     // (read by every thread before thread 0 writes it back, behind two barriers)
    const int par = (int)(epoch & 1);
    const size_t slot_off = oneshot_flag_words() + ((size_t)par * ONESHOT_MAX_RANKS + a.rank) * a.cap;
    // 1. my vector into everybody's mailbox (my own included)
    for (int j = tid; j < a.count; j += blockDim.x) {
        unsigned long long v = a.buf[j];
        if (a.nslices > 1) {                                // this rank's slices, folded in a fixed order
            double x = __longlong_as_double((long long)v);
            for (int sl = 1; sl < a.nslices; ++sl) { x += __longlong_as_double((long long)a.buf[(size_t)sl * a.stride + j]); a.buf[(size_t)sl * a.stride + j] = 0ull; }
            v = (unsigned long long)__double_as_longlong(x);
        }
        for (int r = 0; r < R; ++r) sys_store(a.mail[r] + slot_off + j, v);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope: the payload is out before any flag
    __syncthreads();
    if (tid < R) sys_store(a.mail[tid] + par * ONESHOT_MAX_RANKS + a.rank, epoch);
    // 2. wait for the R vectors of this epoch in MY mailbox
    __shared__ int ok;
    if (tid == 0) ok = 1;
    __syncthreads();
    if (tid < R) {
        const unsigned long long *f = a.mail[a.rank] + par * ONESHOT_MAX_RANKS + tid;
        const unsigned long long *poison = a.mail[a.rank] + oneshot_poison_word();
        long spins = 0;
        while (sys_load(f) != epoch) {
            __builtin_amdgcn_s_sleep(2);
            if ((spins & 1023) == 1023 && sys_load(poison) != 0) { ok = 0; break; }     // another rank has given up: fail with it
            if (++spins > (1L << 26)) { ok = 0; break; }   // (~ seconds: a peer has died)
        }
        if (sys_load(poison) != 0) ok = 0;                 // (sticky: an exchange behind a failed one fails on every rank)
    }
    __syncthreads();
    if (!ok && tid < R) sys_store(a.mail[tid] + oneshot_poison_word(), 1ull);        // publish the expiry to every rank (own mailbox included)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    // 3. reduce in rank order: the same bits on every rank
    const unsigned long long *mine = a.mail[a.rank] + oneshot_flag_words() + (size_t)par * ONESHOT_MAX_RANKS * a.cap;
    for (int j = tid; j < a.count; j += blockDim.x) {
        unsigned long long acc = sys_load(mine + j);
        for (int r = 1; r < R; ++r) {
            const unsigned long long v = sys_load(mine + (size_t)r * a.cap + j);
            if (a.op == ONESHOT_SUM_I64) acc += v;
            else {
                const double x = __longlong_as_double((long long)acc), y = __longlong_as_double((long long)v);
                acc = (unsigned long long)__double_as_longlong(a.op == ONESHOT_SUM_F64 ? x + y : fmax(x, y));
            }
        }
        if (!ok) acc = 0x7ff8000000000000ull;               // NaN: the exchange timed out
        a.buf[j] = acc;
    }
    if (tid == 0) *a.epoch_word = epoch;
}

}  // namespace cfmm
