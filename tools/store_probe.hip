// Store-pattern micro-benchmark (GPU box): what does a CU's store path deliver for the epilogue patterns of the W2 GEMM kernels?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_probe.hip -o tools/store_probe && ./tools/store_probe
// One workgroup of 8 waves per CU writes a 256 x 256 fp32 tile (256 KB) of a [M x 3072] matrix, 64 x f32x4 (or 256 x f32) per lane:
//   mode 0: v_mfma_f32_16x16x32 D layout as used by gemm_p8w: lane & 15 = row, (lane >> 4) * 4 = column -> a 16-lane pass is 16 ROWS
//   mode 1: transposed product layout: lane & 15 = column, 4 rows per lane -> 4 dword stores, a 16-lane pass is 64 contiguous bytes
//   mode 3: 8 rows x 128 B per instruction (what a DPP exchange between lanes r and r + 8 would give the current epilogue)
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
                        } else if (MODE == 3) {
                            // 8 rows x 128 B per instruction (full cache lines): lanes r16 < 8 write columns 0..15 of row r16, lanes
                            // r16 >= 8 columns 16..31 of row r16 - 8; the second instruction of the pair (ni = 1) the rows + 8
                            const int row = wr * 128 + a * 64 + mi * 16 + (r16 & 7) + 8 * ni, col = wc * 64 + b * 32 + (r16 >> 3) * 16 + 4 * kc;
                            *reinterpret_cast<f32x4*>(tile + (size_t)row * ldc + col) = v;
                        } else if (MODE == 4 || MODE == 5) {
                            // lane-adjacent runs over DIFFERENT rows: 4: 8 lanes = 128 B of a row, 8 consecutive rows per instruction
                            // (what the LDS-staged epilogue of gemm_p8w emits); 5: 16 lanes = 256 B, 4 consecutive rows
                            constexpr int LPR = MODE == 4 ? 8 : 16, RPI = 64 / LPR, CPI = LPR * 4;     // lanes per row, rows / columns per instruction
                            const int q = ((a * 2 + b) * 4 + mi) * 2 + ni;          // 0 .. 31: the wave's 128 x 64 part as (128 / RPI) x (64 / CPI) pieces
                            const int pr = q / (64 / CPI), pc = q % (64 / CPI);
                            const int row = wr * 128 + (pr % (128 / RPI)) * RPI + lane / LPR, col = wc * 64 + pc * CPI + 4 * (lane % LPR);
                            *reinterpret_cast<f32x4*>(tile + (size_t)row * ldc + col) = v;
                        } else {
                            const int q = ((a * 2 + b) * 4 + mi) * 2 + ni;          // 0 .. 31: (row group of 8, ...) of the wave's 32 rows
                            const int row = wave * 32 + q, col = 4 * lane;
                            *reinterpret_cast<f32x4*>(tile + (size_t)row * ldc + col) = v;
                        }
                        v.x += 1.f;
                    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // (the WORKGROUP's time: wave 0 alone finishes first -- the first version of this
    const long long t1 = (long long)__builtin_readcyclecounter();   //  probe timed wave 0 only and reported half the real figure)
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

// The epilogue's environment in gemm_p8w: 32 different accumulator registers per lane (no write-after-read on the data), optionally
// 128 KB of LDS per workgroup and only the waves of one wave row storing.  ROWS: 0 = all 8 waves, 1 = wave row 0 only (4 waves).
template <int ROWS, int DATA>
__global__ __launch_bounds__(512) void store_env_k(float* __restrict__ C, int ldc, long long* __restrict__ cyc, int reps, float seed) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3, r16 = lane & 15, kc = lane >> 4;
    float* base = C + (size_t)blockIdx.x * 256 * ldc;
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = DATA ? f32x4{seed * tid + i, 2.f, 3.f, 4.f} : f32x4{seed * (tid + i), seed * lane, 2.f * seed * wave, 3.f * seed + i * lane};
    if (seed == 123.f) lds[tid] = seed;
    __syncthreads();
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
        float* tile = base + (rep & 7) * 256;
        if (ROWS == 0 || wr == 0) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            const int row = wr * 128 + a * 64 + mi * 16 + r16, col = wc * 64 + b * 32 + ni * 16 + 4 * kc;
                            *reinterpret_cast<f32x4*>(tile + (size_t)row * ldc + col) = acc[((a * 2 + b) * 4 + mi) * 2 + ni];
                        }
        }
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = (long long)__builtin_readcyclecounter();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int ROWS, int DATA> static void run_env(const char* name, float* C, long long* cyc, int grid, int lds_bytes) {
    const int reps = 16;
    hipFuncSetAttribute((const void*)store_env_k<ROWS, DATA>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (int k = 0; k < 2; ++k) hipLaunchKernelGGL((store_env_k<ROWS, DATA>), dim3(grid), dim3(512), lds_bytes, 0, C, 3072, cyc, reps, 1.5f);
    hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, cyc, sizeof(long long) * (grid < 256 ? grid : 256), hipMemcpyDeviceToHost);
    printf("%-60s grid %3d: %6lld cycles per tile of the storing waves (workgroup 0), LDS %d KB\n", name, grid, h[0] / reps, lds_bytes / 1024);
}
int main() {
    float* C; hipMalloc(&C, (size_t)65536 * 3072 * 4);
    long long* cyc; hipMalloc(&cyc, 256 * 8);
    for (int grid : {1, 32, 256}) {
        run<0>("f32x4 per lane, lane = row (current epilogue)", C, cyc, grid);
        run<1>("4 x f32 per lane, lane = column (transposed product)", C, cyc, grid);
        run<2>("f32x4 per lane, 64 lanes = 1 KB of a row", C, cyc, grid);
        run<3>("f32x4 per lane, 8 rows x 128 B (full lines)", C, cyc, grid);
        run<4>("f32x4 per lane, 8 adjacent lanes = 128 B, 8 rows per instr", C, cyc, grid);
        run<5>("f32x4 per lane, 16 adjacent lanes = 256 B, 4 rows per instr", C, cyc, grid);
    }
    for (int lds : {0, 128 * 1024}) {
        run_env<0, 0>("32 accumulators per lane, all 8 waves (256 KB), varied data", C, cyc, 1, lds);
        run_env<1, 0>("32 accumulators per lane, wave row 0 only (128 KB), varied data", C, cyc, 1, lds);
        run_env<0, 1>("32 accumulators per lane, all 8 waves (256 KB), {tid + i, 2, 3, 4}", C, cyc, 1, lds);
    }
    return 0;
}
