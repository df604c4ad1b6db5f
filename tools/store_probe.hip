// Store-pattern micro-benchmark (GPU box): what does a CU's store path deliver for the epilogue patterns of the W2 GEMM kernels?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_probe.hip -o tools/store_probe && ./tools/store_probe
// One workgroup of 8 waves per CU writes a 256 x 256 fp32 tile (256 KB) of a [M x 3072] matrix, 64 x f32x4 (or 256 x f32) per lane:
//   mode 0: v_mfma_f32_16x16x32 D layout as used by gemm_p8w: lane & 15 = row, (lane >> 4) * 4 = column -> a 16-lane pass is 16 ROWS
//   mode 1: transposed product layout: lane & 15 = column, 4 rows per lane -> 4 dword stores, a 16-lane pass is 64 contiguous bytes
//   mode 2: (reference) lanes own 4 consecutive columns of one row, 64 lanes = one 1 KB run of a row
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void store_k(float* __restrict__ C, int ldc, long long* __restrict__ cyc, int reps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3, r16 = lane & 15, kc = lane >> 4;
    float* base = C + (size_t)blockIdx.x * 256 * ldc;            // this workgroup's 256 rows, columns 0..255
    f32x4 v = {1.f * tid, 2.f, 3.f, 4.f};
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
        float* tile = base + (rep & 7) * 256;                    // (8 column tiles of the row block in turn)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        if (MODE == 0) {
                            const int row = wr * 128 + a * 64 + mi * 16 + r16, col = wc * 64 + b * 32 + ni * 16 + 4 * kc;
                            *reinterpret_cast<f32x4*>(tile + (size_t)row * ldc + col) = v;
                        } else if (MODE == 1) {
                            const int row = wr * 128 + a * 64 + mi * 16 + 4 * kc, col = wc * 64 + b * 32 + ni * 16 + r16;
#pragma unroll
                            for (int e = 0; e < 4; ++e) tile[(size_t)(row + e) * ldc + col] = v[e];
                        } else {
                            const int q = ((a * 2 + b) * 4 + mi) * 2 + ni;          // 0 .. 31: (row group of 8, ...) of the wave's 32 rows
                            const int row = wave * 32 + q, col = 4 * lane;
                            *reinterpret_cast<f32x4*>(tile + (size_t)row * ldc + col) = v;
                        }
                        v.x += 1.f;
                    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = (long long)__builtin_readcyclecounter();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> static void run(const char* name, float* C, long long* cyc, int grid) {
    const int reps = 16;
    hipLaunchKernelGGL(store_k<MODE>, dim3(grid), dim3(512), 0, 0, C, 3072, cyc, reps);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(store_k<MODE>, dim3(grid), dim3(512), 0, 0, C, 3072, cyc, reps);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, sizeof(long long) * (grid < 256 ? grid : 256), hipMemcpyDeviceToHost);
    printf("%-60s grid %3d: %6lld cycles per 256 KB tile (workgroup 0), %5.2f TB/s aggregate\n", name, grid, h[0] / reps,
           (double)grid * reps * 262144.0 / (ms * 1e-3) * 1e-12);
}
int main() {
    float* C; hipMalloc(&C, (size_t)65536 * 3072 * 4);
    long long* cyc; hipMalloc(&cyc, 256 * 8);
    for (int grid : {1, 32, 256}) {
        run<0>("f32x4 per lane, lane = row (current epilogue)", C, cyc, grid);
        run<1>("4 x f32 per lane, lane = column (transposed product)", C, cyc, grid);
        run<2>("f32x4 per lane, 64 lanes = 1 KB of a row", C, cyc, grid);
    }
    return 0;
}
