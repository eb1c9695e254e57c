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

__device__ __forceinline__ void wave_add64(unsigned long long* dst, uint32_t v)
{
    // wave reduction by DPP-free shuffles, one atomic per wave
    uint32_t s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += (uint32_t)__shfl_down((int)s, off, 64);
    if ((threadIdx.x & 63u) == 0u && s) atomicAdd(dst, (unsigned long long)s);
}


}  // namespace mipt
