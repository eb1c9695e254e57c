// wave_util.h -- wave64 helpers shared by the kernel files
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mipt {

__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// compacting append: lanes with `push` get consecutive indices in a queue whose length lives at `counter`
__device__ __forceinline__ uint32_t wave_append(bool push, uint32_t* counter)
{
    const unsigned long long m = __ballot(push);
    if (m == 0ull) return 0u;
    const int leader = __builtin_ctzll(m);
    uint32_t base = 0u;
    if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    return base + lane_rank(m);
}

// The same for a whole workgroup: ONE global atomic per workgroup and queue.  The queue-length words are single addresses, and
// returning atomics on one word serialise at ~88 per microsecond: one per wave (32 k of them for a 1080p frame) is 0.2-0.4 ms of a
// kernel that otherwise takes 0.05 (k_shade, round 1; k_wh_shade, round 3).  EVERY thread of the workgroup must call it (it
// synchronises); s2: two LDS words of the caller's, s2[0] zero on entry (it is zero again on return).
__device__ __forceinline__ uint32_t block_append(bool push, uint32_t* counter, uint32_t* s2)
{
    const unsigned long long m = __ballot(push);
    uint32_t wo = 0u;
    if ((threadIdx.x & 63u) == 0u && m != 0ull) wo = atomicAdd(&s2[0], (uint32_t)__popcll(m));
    wo = (uint32_t)__builtin_amdgcn_readfirstlane((int)wo);
    __syncthreads();
    if (threadIdx.x == 0u) { const uint32_t c = s2[0]; s2[1] = c ? atomicAdd(counter, c) : 0u; s2[0] = 0u; }
    __syncthreads();
    return s2[1] + wo + lane_rank(m);
}

// a workgroup's sum into a 64-bit counter with ONE atomic (every thread must call it; s1: one LDS word of the caller's, zero on entry).
// Atomics on one address -- returning or not -- take ~8 ns each: one per wave of a 2 M-thread launch is a quarter of a millisecond.
__device__ __forceinline__ void block_add64(unsigned long long* dst, uint32_t v, uint32_t* s1)
{
    uint32_t s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += (uint32_t)__shfl_down((int)s, off, 64);
    if ((threadIdx.x & 63u) == 0u && s) atomicAdd(s1, s);
    __syncthreads();
    if (threadIdx.x == 0u && *s1) atomicAdd(dst, (unsigned long long)*s1);
}

__device__ __forceinline__ void wave_add64(unsigned long long* dst, uint32_t v)
{
    // wave reduction by DPP-free shuffles, one atomic per wave
    uint32_t s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += (uint32_t)__shfl_down((int)s, off, 64);
    if ((threadIdx.x & 63u) == 0u && s) atomicAdd(dst, (unsigned long long)s);
}


}  // namespace mipt
