// Device side of the peer transport (comm.hip): waiting for the ranks' rows of one reduction and adding them in rank order.
// Shared by the stand-alone all-reduce kernel (comm.hip) and the fused tail of a bundle-adjustment iteration (ba.hip).
#pragma once
#include "mqs_common.h"

namespace mqs {
namespace peer {

constexpr long long kSpinTicks = 200000000ll;       // 2 s of the 100 MHz wall clock: a peer that never arrives must not hang the GPU
// The first reduction of a problem: the ranks' host programs have not met yet (a peer may still be uploading its shard, loading
// its code object, or sit in a stalled host), so the wait is a ONE-workgroup kernel of its own with a bound that tolerates that
// (ba_iter.hip) -- not a chip full of workgroups spinning on the fabric, and not 2 s.
constexpr long long kSpinTicksFirst = 3000000000ll; // 30 s

// Every access to a receive buffer is a relaxed system-scope atomic: the buffers are fine-grained memory (not cached by this
// GPU), so coherence is per access and no cache-wide write-back / invalidate (what a system-scope FENCE costs on this part: the
// whole L2, serialised per XCD) is ever needed; ordering comes from completion: the sender's stores are acknowledged (an explicit
// s_waitcnt vmcnt(0), mqs_stores_landed(): a barrier alone does not wait for stores) before it stores the flag, the receiver
// issues its data loads after its flag loads have returned the sequence number.
__device__ __forceinline__ bool spin_until(const unsigned long long *flag, unsigned long long seq, long long ticks)
{
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
        if (wall_clock64() - t0 > ticks) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    return true;
}

// A wait that gave up: the sticky words the host reads (the receive buffer's device int, mqs_comm_peer_timed_out; and, inside an
// iteration of a problem, the problem's host-visible status word, mqs_ba_problem_status / the next mqs_ba_gn_iteration_dev).
__device__ __forceinline__ void report_timeout(const mqs_peer_recv &rv)
{
    *rv.timeout_flag = 1;
    if (rv.status) __hip_atomic_store(rv.status, MQS_STATUS_PEER_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Called by every thread of a workgroup: wait until every rank's pieces of reduction rv.seq have landed (bounded spin).
// `timed_out`: a word in LDS, zero on entry, visible to every thread of the workgroup.  Returns false (to every thread) when a
// piece did not arrive in time: the caller must then NOT read the rows.
__device__ __forceinline__ bool wait_flags(const mqs_peer_recv &rv, int tid, int nthreads, int *timed_out)
{
    bool ok = true;
    for (int f = tid; f < rv.world * rv.flags_per_rank; f += nthreads)
        ok = spin_until(rv.flags + (f / rv.flags_per_rank) * rv.flags_stride + f % rv.flags_per_rank, rv.seq, rv.spin_ticks) && ok;
    if (!ok) { *timed_out = 1; report_timeout(rv); }
    __syncthreads();                     // also keeps the compiler from moving the data loads above the flag loads
    return *timed_out == 0;
}

// Called by every thread of a workgroup: wait for every rank's row of reduction rv.seq, then out[i] = sum over ranks, in rank
// order, of row[rank][i] for i < n (the same bits on every rank).  `out` may be LDS or global.  After a wait that gave up the rows
// are not read and `out` becomes NaN throughout (returns false to every thread): a kernel that runs on it anyway -- the solve
// enqueued behind the stand-alone wait kernel -- then publishes NaN, never numbers computed from an older system.
__device__ __forceinline__ bool wait_and_sum(double *out, int n, const mqs_peer_recv &rv, int tid, int nthreads, int *timed_out)
{
    if (!wait_flags(rv, tid, nthreads, timed_out)) {
        for (int i = tid; i < n; i += nthreads) out[i] = __builtin_nan("");
        return false;
    }
    for (int i = tid; i < n; i += nthreads) {
        double t = 0.0;
        for (int q = 0; q < rv.world; ++q)
            t += __hip_atomic_load(rv.rows + (size_t)q * rv.row_stride + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        out[i] = t;
    }
    return true;
}

// store of one entry of this rank's row into every rank's receive buffer
__device__ __forceinline__ void push_entry(const mqs_peer_push &push, int i, double v)
{
    for (int q = 0; q < push.world; ++q) __hip_atomic_store(push.dst[q] + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// after mqs_stores_landed() + a barrier behind the stores of a piece of the row (they have landed): raise the
// piece's flag in every rank (lanes 0 .. world-1 of one wave)
__device__ __forceinline__ void publish_piece(const mqs_peer_push &push, int piece, int lane)
{
    if (lane < push.world) __hip_atomic_store(push.flag[lane] + piece, push.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace peer
}  // namespace mqs
