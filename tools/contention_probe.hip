// Why does the latency-bound Graphormer chain run ~1.75-2.2x slower beside the persistent W2 weight gradient, whatever the number
// of CUs it is left?  (GPU box)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/contention_probe.hip -o tools/contention_probe && ./tools/contention_probe
// Three probes, each ONE workgroup on the caller's stream (it always finds a free CU):
//   alu    a dependent chain of 200k v_fma_f32 in one wave: time / shader cycle = the CU's clock
//   chase  a dependent pointer chase over a 512 MB ring (every hop a new 4 KB page region: HBM latency incl. the fabric)
//   l2     the same chase over 1 MB (stays in the XCD's L2)
// run alone and beside a LOAD kernel on a second stream that occupies c CUs (8 waves, 128 KB of LDS: nothing co-resides):
//   mfma   v_mfma_f32_16x16x32_f16 back to back on registers, no memory traffic
//   write  streaming 16-byte stores over 2 GB            read   streaming 16-byte loads over 2 GB
//   mfma+write N   N x 8 MFMAs per 32 KB written (N = 32: ~1.2 TB/s of writes on 136 CUs, the weight gradient's rate);  NT = the
//   same with non-temporal stores / loads
// Prints each probe's wall time (100 MHz s_memrealtime), its shader cycles (s_memtime) and the ratio against the run alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void alu_probe(float* out, long long* t, int n) {
    extern __shared__ float plds[];                  // (64 KB: never on a CU that holds a 128 KB load workgroup)
    if (n < 0) out[1] = plds[threadIdx.x];
    const long long r0 = wall_clock64(), c0 = (long long)__builtin_readcyclecounter();
    float a = 1.0f + threadIdx.x * 1e-6f, b = 0.999999f;
    for (int k = 0; k < n; ++k) a = __builtin_fmaf(a, b, 1e-7f);
    const long long c1 = (long long)__builtin_readcyclecounter(), r1 = wall_clock64();
    if (threadIdx.x == 0) { t[0] = r1 - r0; t[1] = c1 - c0; }
    out[threadIdx.x] = a;
}

__global__ void chase_probe(const unsigned* __restrict__ ring, unsigned* out, long long* t, int hops) {
    extern __shared__ float plds[];
    if (hops < 0) out[1] = (unsigned)plds[threadIdx.x];
    unsigned p = out[0];                             // continues where the previous chase stopped: new lines every time
    const long long r0 = wall_clock64(), c0 = (long long)__builtin_readcyclecounter();
    for (int k = 0; k < hops; ++k) p = __builtin_nontemporal_load(ring + p);
    const long long c1 = (long long)__builtin_readcyclecounter(), r1 = wall_clock64();
    if (threadIdx.x == 0) { t[0] = r1 - r0; t[1] = c1 - c0; }
    out[0] = p;
}

// LOAD kernels: grid = c workgroups of 512 threads with 128 KB of dynamic LDS (one per CU, like gemm_p8w), running `iters` rounds
template <int MFMA, int WR, int RD>
__global__ __launch_bounds__(512) void load_kernel(float* __restrict__ buf, size_t n_f4, int iters, float* sink, int reps) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (tid + i)); b[i] = (_Float16)(0.002f * (tid - i)); }
    f32x4* p = reinterpret_cast<f32x4*>(buf);
    const size_t stride = (size_t)gridDim.x * 512;
    size_t pos = (size_t)blockIdx.x * 512 + tid;
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (MFMA) {
            for (int u = 0; u < reps; ++u)           // 8 MFMAs per wave and rep: ~256 cycles of a SIMD that holds two such waves
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        }
        if (WR) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (WR == 2) __builtin_nontemporal_store(acc[u], p + pos); else p[pos] = acc[u];
                pos += stride; if (pos >= n_f4) pos -= n_f4;
            }
        }
        if (RD) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4 v = (RD == 2) ? __builtin_nontemporal_load(p + pos) : p[pos];
                r += v; pos += stride; if (pos >= n_f4) pos -= n_f4;
            }
        }
    }
    float s = r[0] + r[1] + r[2] + r[3];
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) sink[0] = s + lds[tid];
}

struct Probe { const char* name; long long t[2]; };

int main() {
    hipStream_t sa, sb;
    CHK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    int least, greatest;
    CHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CHK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, least));
    // rings
    auto make_ring = [](size_t bytes, size_t step_bytes) {
        const size_t n = bytes / step_bytes;
        std::vector<unsigned> order(n);
        for (size_t i = 0; i < n; ++i) order[i] = (unsigned)i;
        unsigned s = 12345u;
        for (size_t i = n - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; std::swap(order[i], order[(s >> 8) % (i + 1)]); }
        std::vector<unsigned> ring(bytes / 4, 0u);
        for (size_t i = 0; i < n; ++i) ring[(size_t)order[i] * (step_bytes / 4)] = (unsigned)((size_t)order[(i + 1) % n] * (step_bytes / 4));
        // make slot 0 part of the cycle: start at order[0]
        return std::make_pair(ring, (unsigned)((size_t)order[0] * (step_bytes / 4)));
    };
    unsigned *d_big, *d_small, *d_out; long long* d_t; float *d_f, *d_buf, *d_sink;
    const size_t big = 512ull << 20, small = 1ull << 20;
    {
        auto rb = make_ring(big, 4096);
        // (one cycle through every slot: the chase may start at slot 0)
        CHK(hipMalloc(&d_big, big)); CHK(hipMemcpy(d_big, rb.first.data(), big, hipMemcpyHostToDevice));
        auto rs = make_ring(small, 256);
        CHK(hipMalloc(&d_small, small)); CHK(hipMemcpy(d_small, rs.first.data(), small, hipMemcpyHostToDevice));
    }
    CHK(hipMalloc(&d_out, 256)); CHK(hipMalloc(&d_t, 64)); CHK(hipMalloc(&d_f, 4096)); CHK(hipMalloc(&d_sink, 256));
    const size_t buf_bytes = 2ull << 30;
    CHK(hipMalloc(&d_buf, buf_bytes)); CHK(hipMemset(d_buf, 0, buf_bytes));
    const size_t n_f4 = buf_bytes / 16;
    const int lds_bytes = 128 * 1024;
    auto set_lds = [&](const void* f) { CHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); };
    set_lds((const void*)load_kernel<1, 0, 0>); set_lds((const void*)load_kernel<0, 1, 0>);
    set_lds((const void*)load_kernel<0, 0, 1>); set_lds((const void*)load_kernel<1, 1, 0>);
    set_lds((const void*)load_kernel<1, 2, 0>); set_lds((const void*)load_kernel<0, 0, 2>); set_lds((const void*)load_kernel<0, 2, 0>);

    struct Load { const char* name; int kind; int c; int reps; int iters; };
    const Load loads[] = {{"alone", -1, 0, 0, 0},
                          {"mfma          c=136", 0, 136, 4, 60000}, {"mfma          c=224", 0, 224, 4, 60000},
                          {"write (full)  c=136", 1, 136, 0, 40000}, {"read (full)   c=136", 2, 136, 0, 24000},
                          {"mfma+write 64 c=136", 3, 136, 64, 2500}, {"mfma+write 32 c=136", 3, 136, 32, 5000},
                          {"mfma+write 16 c=136", 3, 136, 16, 10000}, {"mfma+write  8 c=136", 3, 136, 8, 16000},
                          {"mfma+write 32 c=224", 3, 224, 32, 5000}, {"mfma+write 32 c=64 ", 3, 64, 32, 5000},
                          {"mfma+NTwrite 32 136", 4, 136, 32, 5000}, {"mfma+NTwrite 16 136", 4, 136, 16, 10000},
                          {"NT write (full) 136", 6, 136, 0, 40000}, {"NT read (full)  136", 5, 136, 0, 24000}};
    const int probe_lds = 64 * 1024;
    CHK(hipFuncSetAttribute((const void*)alu_probe, hipFuncAttributeMaxDynamicSharedMemorySize, probe_lds));
    CHK(hipFuncSetAttribute((const void*)chase_probe, hipFuncAttributeMaxDynamicSharedMemorySize, probe_lds));
    unsigned* d_out2;
    CHK(hipMalloc(&d_out2, 256)); CHK(hipMemset(d_out, 0, 256)); CHK(hipMemset(d_out2, 0, 256));
    long long base[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    printf("%-20s | %-9s | %-28s | %-34s | %-34s\n", "load", "writes", "alu chain: us, clock, ratio", "chase 512 MB: ns/hop cyc/hop ratio", "chase 1 MB: ns/hop cyc/hop ratio");
    for (const Load& L : loads) {
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipDeviceSynchronize());
        if (L.kind >= 0) {
            CHK(hipEventRecord(e0, sb));
            if (L.kind == 0) hipLaunchKernelGGL((load_kernel<1, 0, 0>), dim3(L.c), dim3(512), lds_bytes, sb, d_buf, n_f4, L.iters, d_sink, L.reps);
            if (L.kind == 1) hipLaunchKernelGGL((load_kernel<0, 1, 0>), dim3(L.c), dim3(512), lds_bytes, sb, d_buf, n_f4, L.iters, d_sink, L.reps);
            if (L.kind == 2) hipLaunchKernelGGL((load_kernel<0, 0, 1>), dim3(L.c), dim3(512), lds_bytes, sb, d_buf, n_f4, L.iters, d_sink, L.reps);
            if (L.kind == 3) hipLaunchKernelGGL((load_kernel<1, 1, 0>), dim3(L.c), dim3(512), lds_bytes, sb, d_buf, n_f4, L.iters, d_sink, L.reps);
            if (L.kind == 4) hipLaunchKernelGGL((load_kernel<1, 2, 0>), dim3(L.c), dim3(512), lds_bytes, sb, d_buf, n_f4, L.iters, d_sink, L.reps);
            if (L.kind == 5) hipLaunchKernelGGL((load_kernel<0, 0, 2>), dim3(L.c), dim3(512), lds_bytes, sb, d_buf, n_f4, L.iters, d_sink, L.reps);
            if (L.kind == 6) hipLaunchKernelGGL((load_kernel<0, 2, 0>), dim3(L.c), dim3(512), lds_bytes, sb, d_buf, n_f4, L.iters, d_sink, L.reps);
            CHK(hipEventRecord(e1, sb));
        }
        long long t[3][2] = {{0, 0}, {0, 0}, {0, 0}}, tt[3][2];
        const int n_alu = 20000, hops_big = 400, hops_small = 2000;
        int rounds = 0;
        for (int round = 0; round < 4; ++round) {       // the last round that ended while the load was still running counts
            hipLaunchKernelGGL(alu_probe, dim3(1), dim3(64), probe_lds, sa, d_f, d_t, n_alu);
            CHK(hipStreamSynchronize(sa)); CHK(hipMemcpy(tt[0], d_t, 16, hipMemcpyDeviceToHost));
            hipLaunchKernelGGL(chase_probe, dim3(1), dim3(64), probe_lds, sa, d_big, d_out, d_t, hops_big);
            CHK(hipStreamSynchronize(sa)); CHK(hipMemcpy(tt[1], d_t, 16, hipMemcpyDeviceToHost));
            hipLaunchKernelGGL(chase_probe, dim3(1), dim3(64), probe_lds, sa, d_small, d_out2, d_t, hops_small);
            CHK(hipStreamSynchronize(sa)); CHK(hipMemcpy(tt[2], d_t, 16, hipMemcpyDeviceToHost));
            if (L.kind >= 0 && hipEventQuery(e1) != hipErrorNotReady) break;
            if (round >= 1 || L.kind >= 0) {}
            for (int i = 0; i < 3; ++i) { t[i][0] = tt[i][0]; t[i][1] = tt[i][1]; }
            ++rounds;
        }
        CHK(hipDeviceSynchronize());
        float ms = 0.f;
        if (L.kind >= 0) CHK(hipEventElapsedTime(&ms, e0, e1));
        if (L.kind < 0) for (int i = 0; i < 3; ++i) { base[i][0] = t[i][0]; base[i][1] = t[i][1]; }
        if (rounds == 0) { printf("%-20s | the load (%.2f ms) ended before one round of probes: raise iters\n", L.name, ms); continue; }
        const double wr_tbs = (L.kind == 1 || L.kind == 3 || L.kind == 4 || L.kind == 6) ? (double)L.c * 512 * 64 * L.iters / (ms * 1e-3) / 1e12 : 0.0;
        const double us_alu = t[0][0] / 100.0;
        printf("%-20s | %4.2f TB/s | %7.1f us %5.0f MHz  x%.2f  | %7.1f ns %7.1f cyc  x%.2f       | %7.1f ns %7.1f cyc  x%.2f       (load %.1f ms, %d rounds)\n",
               L.name, wr_tbs, us_alu, t[0][1] / us_alu, (double)t[0][0] / base[0][0],
               t[1][0] * 10.0 / hops_big, (double)t[1][1] / hops_big, (double)t[1][0] / base[1][0],
               t[2][0] * 10.0 / hops_small, (double)t[2][1] / hops_small, (double)t[2][0] / base[2][0], ms, rounds);
    }
    return 0;
}
