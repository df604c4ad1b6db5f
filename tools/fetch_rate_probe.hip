// Sustained fetch rate of ONE workgroup per CU (GPU box): plain 16-byte loads into registers against LDS-DMA
// (global_load_lds_dwordx4), 256 / 512 threads, 8-64 KB per batch, data from L2 (small working set) or beyond it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fetch_rate_probe.hip -o /tmp/fetch_probe && /tmp/fetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GAS __attribute__((address_space(1)))
#define LAS __attribute__((address_space(3)))

template <int PER>   // float4 loads per thread and batch
__global__ void vgpr_stream(const float* __restrict__ x, float* __restrict__ sink, long long* st, int batches, size_t ws_floats) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const size_t wg_base = ((size_t)blockIdx.x * 7919 * 4096) % ws_floats;
    long long t0 = (long long)__builtin_readcyclecounter();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    size_t off = wg_base;
    for (int b = 0; b < batches; ++b) {
        f32x4 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) v[i] = *reinterpret_cast<const f32x4*>(x + (off + (size_t)(i * nt + tid) * 4) % ws_floats);
#pragma unroll
        for (int i = 0; i < PER; ++i) acc += v[i];
        off += (size_t)PER * nt * 4;
    }
    long long t1 = (long long)__builtin_readcyclecounter();
    if (acc.x == 12345.f) sink[tid] = acc.y;
    if (blockIdx.x == 0 && tid == 0) { st[0] = t0; st[1] = t1; }
}

template <int PER>
__global__ void dma_stream(const float* __restrict__ x, float* __restrict__ sink, long long* st, int batches, size_t ws_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6;
    const size_t wg_base = ((size_t)blockIdx.x * 7919 * 4096) % ws_floats;
    long long t0 = (long long)__builtin_readcyclecounter();
    size_t off = wg_base;
    for (int b = 0; b < batches; ++b) {
        char* dst = reinterpret_cast<char*>(lds) + (b & 1) * PER * nt * 16;
#pragma unroll
        for (int i = 0; i < PER; ++i)
            __builtin_amdgcn_global_load_lds((const void GAS*)(x + (off + (size_t)(i * nt + tid) * 4) % ws_floats),
                                             (LAS void*)(dst + (size_t)(i * nt + wave * 64) * 16), 16, 0, 0);
        if (b >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");   // the previous batch landed
        off += (size_t)PER * nt * 4;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = (long long)__builtin_readcyclecounter();
    if (lds[tid] == 12345.f) sink[tid] = lds[tid + 1];
    if (blockIdx.x == 0 && tid == 0) { st[0] = t0; st[1] = t1; }
}

int main() {
    const size_t big = (size_t)1 << 28;       // 1 GiB of floats / 4 = 256 Mi floats -> beyond every cache
    float* x; hipMalloc(&x, big * sizeof(float)); hipMemset(x, 0, big * sizeof(float));
    float* sink; hipMalloc(&sink, 4096); long long* st; hipMalloc(&st, 64);
    hipFuncSetAttribute((const void*)dma_stream<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipFuncSetAttribute((const void*)dma_stream<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    for (size_t ws : {(size_t)1 << 18, big}) {                 // 1 MiB working set (L2) / 1 GiB
        for (int threads : {256, 512}) {
            for (int grid : {1, 256}) {
                const int batches = 32;
                auto report = [&](const char* name, int per) {
                    long long h[2]; hipDeviceSynchronize(); hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
                    const double bytes = (double)batches * per * threads * 16;
                    printf("%-6s ws=%4zu MiB threads=%3d grid=%3d per-batch=%3d KB: %7lld cycles, %.1f B/clk/CU\n", name,
                           ws * 4 >> 20, threads, grid, per * threads * 16 / 1024, h[1] - h[0], bytes / (double)(h[1] - h[0]));
                };
                for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(vgpr_stream<8>, dim3(grid), dim3(threads), 0, 0, x, sink, st, batches, ws);
                report("vgpr", 8);
                for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(vgpr_stream<16>, dim3(grid), dim3(threads), 0, 0, x, sink, st, batches, ws);
                report("vgpr", 16);
                for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(dma_stream<4>, dim3(grid), dim3(threads), 2 * 4 * threads * 16, 0, x, sink, st, batches, ws);
                report("dma", 4);
                for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(dma_stream<8>, dim3(grid), dim3(threads), 2 * 8 * threads * 16, 0, x, sink, st, batches, ws);
                report("dma", 8);
            }
        }
    }
    return 0;
}
