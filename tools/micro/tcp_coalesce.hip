// tcp_coalesce.hip -- microbenchmark (round 3): what does a lane-divergent 16-byte buffer load cost in the vector L1, and does it get
// cheaper when the lanes of a pair / a quad read neighbouring 16-byte pieces of one 32- / 64-byte record (the same bytes per
// instruction, fewer cache lines per quad of lanes)?  Table sizes: 16 KB (L1 resident), 2 MB (L2 resident), 64 MB (beyond L2).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/tcp_coalesce.hip -o /tmp/tcp_coalesce && /tmp/tcp_coalesce
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int GROUP>      // lanes that read consecutive 16-byte pieces of one record: 1 (every lane its own record), 2, 4
__global__ __launch_bounds__(64) void k_chase(const void* table, unsigned bytes, unsigned iters, unsigned* out)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (int)bytes, 0x00020000);
    const unsigned lane = threadIdx.x, sub = lane & (GROUP - 1);
    unsigned rec = (blockIdx.x * 64u + (lane & ~(unsigned)(GROUP - 1))) * 2654435761u;       // the same record for the lanes of a group
    const unsigned n_rec = bytes / 64u;
    unsigned acc = 0;
    for (unsigned i = 0; i < iters; ++i) {
        const unsigned off = (rec % n_rec) * 64u + sub * 16u;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
        acc += v.x ^ v.w;
        rec = rec * 1664525u + 1013904223u + (v.y & 1u);          // dependent on the load: one load in flight per wave
    }
    if (acc == 0x12345u) out[0] = acc;
}

int main()
{
    for (size_t bytes : { (size_t)16 << 10, (size_t)2 << 20, (size_t)64 << 20 }) {
        void* d = nullptr; unsigned* out = nullptr;
        hipMalloc(&d, bytes); hipMalloc((void**)&out, 4);
        std::vector<unsigned> h(bytes / 4);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)rand();
        hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
        const unsigned iters = 2000, blocks = 256 * 24;
        for (int g : { 1, 2, 4 }) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (g == 1) hipLaunchKernelGGL(k_chase<1>, dim3(blocks), dim3(64), 0, 0, d, (unsigned)bytes, iters, out);
                if (g == 2) hipLaunchKernelGGL(k_chase<2>, dim3(blocks), dim3(64), 0, 0, d, (unsigned)bytes, iters, out);
                if (g == 4) hipLaunchKernelGGL(k_chase<4>, dim3(blocks), dim3(64), 0, 0, d, (unsigned)bytes, iters, out);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            const double loads = (double)blocks * iters;       // wave-level load instructions
            printf("table %6zu KB, %d lane(s) per record: %8.3f ms, %6.1f cycles per wave-load per CU (2.4 GHz, 256 CUs, 24 waves each)\n", bytes >> 10, g, ms,
                   ms * 1e-3 * 2.4e9 / (loads / 256.0));
        }
        hipFree(d); hipFree(out);
    }
    return 0;
}
