// tcp_mask.hip -- microbenchmark (round 3): does a lane-divergent 16-byte buffer load cost less when fewer lanes of the wave take part
// (exec mask), i.e. is the vector-memory pipeline's cost per instruction or per active lane?  And what does it cost when all lanes
// read the SAME address (a coherent tile at the top of the tree)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int ACTIVE, bool SAME>
__global__ __launch_bounds__(64) void k_chase(const void* table, unsigned bytes, unsigned iters, unsigned* out)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (int)bytes, 0x00020000);
    unsigned rec = (blockIdx.x * 64u + (SAME ? 0u : threadIdx.x)) * 2654435761u;
    const unsigned n_rec = bytes / 64u;
    unsigned acc = 0;
    if (threadIdx.x < (unsigned)ACTIVE) {
        for (unsigned i = 0; i < iters; ++i) {
            const unsigned off = (rec % n_rec) * 64u;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
            acc += v.x ^ v.w;
            rec = rec * 1664525u + 1013904223u + (v.y & 1u);
        }
    }
    if (acc == 0x12345u) out[0] = acc;
}

template <int ACTIVE, bool SAME> static void run(const char* what, const void* d, size_t bytes, unsigned* out)
{
    const unsigned iters = 2000, blocks = 256 * 24;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k_chase<ACTIVE, SAME>), dim3(blocks), dim3(64), 0, 0, d, (unsigned)bytes, iters, out);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    }
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    printf("table %6zu KB, %-28s: %8.3f ms, %6.1f cycles per wave-load per CU\n", bytes >> 10, what, ms, ms * 1e-3 * 2.4e9 / ((double)blocks * iters / 256.0));
}

int main()
{
    for (size_t bytes : { (size_t)16 << 10, (size_t)2 << 20 }) {
        void* d = nullptr; unsigned* out = nullptr;
        (void)hipMalloc(&d, bytes); (void)hipMalloc((void**)&out, 4);
        std::vector<unsigned> h(bytes / 4);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)rand();
        (void)hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
        run<64, false>("64 lanes, divergent", d, bytes, out);
        run<48, false>("48 lanes, divergent", d, bytes, out);
        run<32, false>("32 lanes, divergent", d, bytes, out);
        run<16, false>("16 lanes, divergent", d, bytes, out);
        run<8, false>("8 lanes, divergent", d, bytes, out);
        run<64, true>("64 lanes, one address", d, bytes, out);
        (void)hipFree(d); (void)hipFree(out);
    }
    return 0;
}
