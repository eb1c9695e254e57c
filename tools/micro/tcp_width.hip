// tcp_width.hip -- microbenchmark (round 3): the cost of one lane-divergent buffer load per wave instruction and CU, by width
// (4 / 8 / 12 / 16 bytes per lane) and by where the table lives (L1 / L2 / beyond), 24 waves per CU each with ONE load in flight
// (a dependent chain, as a traversal step has), and with TWO independent loads in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int W, int K>      // W dwords per lane and load, K independent loads per step
__global__ __launch_bounds__(64) void k_chase(const void* table, unsigned bytes, unsigned iters, unsigned* out)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (int)bytes, 0x00020000);
    unsigned rec = (blockIdx.x * 64u + threadIdx.x) * 2654435761u;
    const unsigned n_rec = bytes / 64u;
    unsigned acc = 0;
    for (unsigned i = 0; i < iters; ++i) {
        unsigned got = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned off = ((rec + 977u * k) % n_rec) * 64u + 16u * k;
            if (W == 1) got += (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0);
            if (W == 2) { const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0); got += v.x ^ v.y; }
            if (W == 3) { const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)off, 0, 0); got += v.x ^ v.z; }
            if (W == 4) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0); got += v.x ^ v.w; }
        }
        acc += got;
        rec = rec * 1664525u + 1013904223u + (got & 1u);
    }
    if (acc == 0x12345u) out[0] = acc;
}

template <int W, int K> static float run(const void* d, size_t bytes, unsigned iters, unsigned blocks, unsigned* out)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k_chase<W, K>), dim3(blocks), dim3(64), 0, 0, d, (unsigned)bytes, iters, out);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    }
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    for (size_t bytes : { (size_t)16 << 10, (size_t)2 << 20, (size_t)64 << 20 }) {
        void* d = nullptr; unsigned* out = nullptr;
        (void)hipMalloc(&d, bytes); (void)hipMalloc((void**)&out, 4);
        std::vector<unsigned> h(bytes / 4);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)rand();
        (void)hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
        const unsigned iters = 2000, blocks = 256 * 24;
        const float t[8] = { run<1, 1>(d, bytes, iters, blocks, out), run<2, 1>(d, bytes, iters, blocks, out), run<3, 1>(d, bytes, iters, blocks, out), run<4, 1>(d, bytes, iters, blocks, out),
                             run<1, 2>(d, bytes, iters, blocks, out), run<2, 2>(d, bytes, iters, blocks, out), run<4, 2>(d, bytes, iters, blocks, out), run<4, 4>(d, bytes, iters, blocks, out) };
        const char* name[8] = { "b32 x1", "b64 x1", "b96 x1", "b128 x1", "b32 x2", "b64 x2", "b128 x2", "b128 x4" };
        const int k[8] = { 1, 1, 1, 1, 2, 2, 2, 4 };
        for (int i = 0; i < 8; ++i)
            printf("table %6zu KB, %-8s: %8.3f ms, %6.1f cycles per wave-load per CU\n", bytes >> 10, name[i], t[i], t[i] * 1e-3 * 2.4e9 / ((double)blocks * iters * k[i] / 256.0));
        (void)hipFree(d); (void)hipFree(out);
    }
    return 0;
}
