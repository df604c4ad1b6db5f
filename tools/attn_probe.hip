// Phase stamps of the attention forward kernel + cold-load latency of one workgroup (GPU box):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude
//   -Ighn3_amd/csrc tools/attn_probe.hip -o /tmp/attn_probe && /tmp/attn_probe
#define GHN3_ATTN_PROBE 1
#include <cstdarg>
#include <cstdio>
#include <vector>
#include "../ghn3_amd/csrc/attention.hip"
void ghn3_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }

__global__ void latency_kernel(const float* __restrict__ x, float* __restrict__ sink, long long* st, int n_loads, size_t stride) {
    const int tid = threadIdx.x;
    long long t0 = (long long)__builtin_readcyclecounter();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* p = x + (size_t)blockIdx.x * 65536 + tid * 4;
    for (int i = 0; i < n_loads; ++i) acc += *reinterpret_cast<const f32x4*>(p + i * stride);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = (long long)__builtin_readcyclecounter();
    if (acc.x == 12345.f) sink[tid] = acc.y;
    if (blockIdx.x == 0 && tid == 0) { st[0] = t0; st[1] = t1; }
}

int main() {
    const int B = 1, N = 256, H = 16, C = 384;
    float *qkv, *bias, *P, *out; int* nn;
    hipMalloc(&qkv, sizeof(float) * B * N * 3 * C); hipMalloc(&bias, sizeof(float) * B * H * N * N);
    hipMalloc(&P, sizeof(float) * B * H * N * N); hipMalloc(&out, sizeof(float) * B * N * C); hipMalloc(&nn, sizeof(int));
    std::vector<float> h(B * N * 3 * C);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * (float)((i * 2654435761u) % 200) - 1.0f;
    hipMemcpy(qkv, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    hipMemset(bias, 0, sizeof(float) * B * H * N * N);
    hipMemcpy(nn, &N, sizeof(int), hipMemcpyHostToDevice);
    const char* names[] = {"start -> operand loads issued", "loads landed + S = K Q^T + scale/mask", "row max exchange (barrier)",
                           "exp + row sums", "sum exchange (barrier) .. P store + O^T += V^T P^T", "wave reduction (LDS)", "store"};
    for (int rep = 0; rep < 4; ++rep) {
        ghn3_attn_fwd(out, qkv, bias, P, nn, B, N, C, H, 0);
        hipDeviceSynchronize();
        long long st[64];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_attn_stamps), sizeof(st));
        printf("rep %d: total %lld cycles\n", rep, st[7] - st[0]);
        for (int i = 0; i < 7; ++i) printf("   %-58s %8lld\n", names[i], st[i + 1] - st[i]);
    }
    {
        float* big; hipMalloc(&big, sizeof(float) * 65536 * 300); hipMemset(big, 0, sizeof(float) * 65536 * 300);
        long long* dst; hipMalloc(&dst, 64);
        for (int n_loads : {1, 4, 13}) {
            for (int grid : {1, 128}) {
                for (int rep = 0; rep < 3; ++rep) {
                    hipLaunchKernelGGL(latency_kernel, dim3(grid), dim3(256), 0, 0, big, out, dst, n_loads, (size_t)1024);
                    hipDeviceSynchronize();
                    long long hst[2]; hipMemcpy(hst, dst, 16, hipMemcpyDeviceToHost);
                    printf("latency: %2d float4 loads per lane, grid %3d, rep %d: %lld cycles\n", n_loads, grid, rep, hst[1] - hst[0]);
                }
            }
        }
    }
    {   // backward: phase stamps of the first row-role and the first column-role workgroup
        float *dO, *dqkv, *dB;
        hipMalloc(&dO, sizeof(float) * B * N * C); hipMalloc(&dqkv, sizeof(float) * B * N * 3 * C); hipMalloc(&dB, sizeof(float) * B * H * N * N);
        hipMemcpy(dO, h.data(), sizeof(float) * B * N * C, hipMemcpyHostToDevice);
        hipMemset(dB, 0, sizeof(float) * B * H * N * N);
        const char* rn[] = {"start -> tile + staging loads issued, LDS written", "barrier", "operands from LDS, dP, dS, dBias store, dQ",
                            "amax + wave reduction (LDS)", "store"};
        const char* cn[] = {"start -> P + staging loads issued, LDS written", "barrier", "operands from LDS, dP, dS, dV, dK",
                            "two wave reductions + stores"};
        for (int rep = 0; rep < 4; ++rep) {
            ghn3_attn_bwd(dqkv, dO, qkv, P, out, nullptr, dB, nn, B, N, C, H, 0, 0);
            hipDeviceSynchronize();
            long long st[64];
            hipMemcpyFromSymbol(st, HIP_SYMBOL(g_attn_stamps), sizeof(st));
            printf("bwd (staged) rep %d: row role %lld cycles, column role %lld cycles\n", rep, st[21] - st[16], st[36] - st[32]);
            for (int i = 0; i < 5; ++i) printf("   row  %-58s %8lld\n", rn[i], st[17 + i] - st[16 + i]);
            for (int i = 0; i < 4; ++i) printf("   col  %-58s %8lld\n", cn[i], st[33 + i] - st[32 + i]);
        }
        hipEvent_t a, b_; hipEventCreate(&a); hipEventCreate(&b_);
        hipEventRecord(a, 0);
        for (int i = 0; i < 200; ++i) ghn3_attn_bwd(dqkv, dO, qkv, P, out, nullptr, dB, nn, B, N, C, H, 0, 0);
        hipEventRecord(b_, 0); hipEventSynchronize(b_);
        float msb; hipEventElapsedTime(&msb, a, b_);
        printf("bwd: %.2f us per launch (200 back to back, instrumented build)\n", msb * 5.0f);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 200; ++i) ghn3_attn_fwd(out, qkv, bias, P, nn, B, N, C, H, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%.2f us per launch (200 back to back, instrumented build)\n", ms * 5.0f);
    return 0;
}
