// Device side of the peer transport (comm.hip): waiting for the ranks' rows of one reduction and adding them in rank order.
// Shared by the stand-alone all-reduce kernel (comm.hip) and the fused tail of a bundle-adjustment iteration (ba.hip).
#pragma once
#include "mqs_common.h"

namespace mqs {
namespace peer {

constexpr long long kSpinTicks = 200000000ll;       // 2 s of the 100 MHz wall clock: a peer that never arrives must not hang the GPU

__device__ __forceinline__ bool spin_until(const unsigned long long *flag, unsigned long long seq)
{
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
        if (wall_clock64() - t0 > kSpinTicks) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    return true;
}

// Called by every thread of a workgroup: wait for every rank's row of reduction rv.seq, then out[i] = sum over ranks, in rank
// order, of row[rank][i] for i < n (the same bits on every rank).  `out` may be LDS or global.
__device__ __forceinline__ void wait_and_sum(double *out, int n, const mqs_peer_recv &rv, int tid, int nthreads)
{
    bool ok = true;
    for (int f = tid; f < rv.world * rv.flags_per_rank; f += nthreads)
        ok = spin_until(rv.flags + (f / rv.flags_per_rank) * rv.flags_stride + f % rv.flags_per_rank, rv.seq) && ok;
    if (!ok) *rv.timeout_flag = 1;
    __syncthreads();
    __threadfence_system();
    const volatile double *r = rv.rows;
    for (int i = tid; i < n; i += nthreads) {
        double t = 0.0;
        for (int q = 0; q < rv.world; ++q) t += r[(size_t)q * rv.row_stride + i];
        out[i] = t;
    }
}

// after this thread's stores of its piece of the row: make them visible system-wide, then raise the piece's flag in every rank
__device__ __forceinline__ void publish_piece(const mqs_peer_push &push, int piece, int lane)
{
    __threadfence_system();
    if (lane < push.world) __hip_atomic_store(push.flag[lane] + piece, push.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace peer
}  // namespace mqs
