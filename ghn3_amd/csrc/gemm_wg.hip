// Split-bf16 weight-gradient GEMM for gfx950 (tile code 48): C[M x N] (+)= alpha * sum_k A[k][m] * B[k][n].
//
// Replaces the autograd weight gradients of the Graphormer linears (/root/reference/ghn3/graphormer.py:208-248: to_qkv,
// to_out, ff.net.0, ff.net.3; dW = dY^T X) and their fused bias gradients (db = column sums of dY).  Both operands are
// fp32 activations [rows][features] in HBM, the reduction runs over ROWS (k-strided operands: GHN3_MODE_COL / GHN3_MODE_COL),
// K = B * N nodes (256 for the bench workload), so there is no k-contiguous copy to DMA from.  Rounds 1-2 ran these on the
// exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32, 64 x 64 tiles of the generic kernel): 30 us per layer alone,
// 0.9 ms per step on the side stream beside the dependent chain.  Here
//   * a workgroup (4 waves, 2 x 2) owns a 64 x 64 tile of C and walks K in chunks of 64 rows;
//   * a chunk of A [64 k][64 m] and of B [64 k][64 n] is read with coalesced 16-byte loads (a lane: 4 consecutive features
//     of TWO consecutive rows), split into bf16 hi = bf16(x), lo = bf16(x - hi) and written TRANSPOSED into LDS as
//     [feature][k] images (a pair of rows -> one ds_write_b32), 128-byte rows, 16-byte slot s of row r holds k chunk
//     s ^ ((r >> 1) & 7): the fragment reads of v_mfma_f32_16x16x32_bf16 are conflict-free ds_read_b128;
//   * products hi*hi + hi*lo + lo*hi (fp32 accumulate, the cross terms in their own accumulator): ~8e-6 relative, as the
//     split-bf16 linears of the chain (gemm_x3.hip);
//   * the loads of chunk c + 1 are issued before the MFMAs of chunk c (register prefetch, one LDS stage of 32 KB: several
//     workgroups per CU hide each other's staging);
//   * the products are taken transposed (B fragment first), a lane owns 4 consecutive columns of a row: float4 epilogue with
//     optional accumulate; the workgroups of column tile 0 also reduce their A chunks over k: the bias gradient, one writer
//     per element (deterministic).
// Contract (host-checked, runtime.hip): fp32 operands, COL / COL modes, no gathers / maps / activation / residual / split-K,
// M % 4 == 0, N % 4 == 0, lda / ldb / ldc % 4 == 0, 16-byte aligned bases.

#include "ghn3_internal.h"

#define GAS __attribute__((address_space(1)))
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef f32x4 GAS* gf4;
typedef const f32x4 GAS* gcf4;

namespace {

constexpr int WG_KC = 64;
// T x T tiles: T = 64 (4 waves, tile code 48) or T = 128 (8 waves, tile code 49: half the operand bytes per flop -- the
// launch that carries all layers' weight gradients behind the chain is bound by the L2 -> CU rate, 1.36 GB at T = 64)
template <int T> constexpr int wg_lds() { return 4 * T * 128 + 16 * T * 4; }   // A hi | A lo | B hi | B lo images + bias partials

__device__ __forceinline__ unsigned wg_pack(__bf16 a, __bf16 b) {
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

// rows k, k + 1 of 4 consecutive features -> hi / lo pairs at [feature f .. f + 3][k, k + 1] of the transposed images
__device__ __forceinline__ void wg_stage(char* s_hi, char* s_lo, int f, int k, const f32x4& r0, const f32x4& r1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h0 = (__bf16)r0[e], h1 = (__bf16)r1[e];
        const __bf16 l0 = (__bf16)(r0[e] - (float)h0), l1 = (__bf16)(r1[e] - (float)h1);
        const int row = f + e;
        const int off = row * 128 + (((k >> 3) ^ ((row >> 1) & 7)) << 4) + ((k & 7) << 1);
        *reinterpret_cast<unsigned*>(s_hi + off) = wg_pack(h0, h1);
        *reinterpret_cast<unsigned*>(s_lo + off) = wg_pack(l0, l1);
    }
}

// one T x T tile (tile id `tile` of the launch)
template <int T>
__device__ __forceinline__ void wg_tile(const GemmProbDev* __restrict__ probs, int n_probs, int tile, char* sm) {
    constexpr int WG_BM = T, WG_BN = T;
    constexpr int TN = T / 32;                           // 16-column MFMA tiles per wave (wave tile 32 x T / 2)
    constexpr int FPR = T / 4;                           // float4 per operand row of a chunk
    char* sAh = sm;
    char* sAl = sm + T * 128;
    char* sBh = sm + 2 * T * 128;
    char* sBl = sm + 3 * T * 128;
    float* sbias = reinterpret_cast<float*>(sm + 4 * T * 128);

    int lo = 0, hi_ = n_probs - 1;
    while (lo < hi_) {
        const int mid = (lo + hi_ + 1) >> 1;
        if (probs[mid].tile_start <= tile) lo = mid; else hi_ = mid - 1;
    }
    const GemmProbDev* P = probs + lo;
    const int t_id = tile - P->tile_start;
    const int n0 = (t_id % P->tiles_n) * WG_BN, m0 = (t_id / P->tiles_n) * WG_BM;
    const int M = P->M, N = P->N, K = P->K, lda = P->lda, ldb = P->ldb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * (T / 2);
    const bool want_bias = (P->flags & GHN3_GEMM_BIASGRAD) && n0 == 0;

    // staging role: features f4 .. f4 + 3, rows 2 kp + 32 i + {0, 1} of the chunk (i = 0, 1).  A wave covers 8 feature groups
    // (128 contiguous bytes of a row) x 8 row pairs: its ds_write_b32 of one element then land in 8 swizzle slots x 4 words =
    // 32 banks, two lanes each -- the LDS rate for 256 bytes.  (Round 3 mapping, 16 feature groups x 4 row pairs per wave: the
    // rows 4 t + e of one instruction share their parity and rows 16 apart share their slot -> 16 banks, four lanes each;
    // SQ_LDS_BANK_CONFLICT was 60 % of the kernel's LDS cycles, profiles/r04z_pmc_xl_f16.txt.)
    constexpr int FGW = FPR / 8;                         // waves along the feature groups
    const int f4 = ((lane & 7) + 8 * (wave % FGW)) * 4, kp = (lane >> 3) + 8 * (wave / FGW);
    const float GAS* Ag = (const float GAS*)P->A;
    const float GAS* Bg = (const float GAS*)P->B;
    const bool a_in = m0 + f4 < M, b_in = n0 + f4 < N;       // (M % 4 == 0, N % 4 == 0: a float4 is inside or outside)
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 ra[2][2], rb[2][2];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int k = k0 + 2 * kp + 32 * i + r;
                ra[i][r] = (a_in && k < K) ? *reinterpret_cast<gcf4>(Ag + (int64_t)k * lda + m0 + f4) : zero;
                rb[i][r] = (b_in && k < K) ? *reinterpret_cast<gcf4>(Bg + (int64_t)k * ldb + n0 + f4) : zero;
            }
    };

    f32x4 acc0[2][TN], acc1[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { acc0[i][j] = zero; acc1[i][j] = zero; }
    f32x4 bsum = zero;

    load_chunk(0);
    for (int k0 = 0; k0 < K; k0 += WG_KC) {
        if (k0 > 0) __syncthreads();                      // the fragment reads of the previous chunk are done
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            wg_stage(sAh, sAl, f4, 2 * kp + 32 * i, ra[i][0], ra[i][1]);
            wg_stage(sBh, sBl, f4, 2 * kp + 32 * i, rb[i][0], rb[i][1]);
            if (want_bias) bsum += ra[i][0] + ra[i][1];
        }
        if (k0 + WG_KC < K) load_chunk(k0 + WG_KC);       // in flight during this chunk's products
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ch = 4 * h + lq;
            bf16x8 xh[2], xl[2], wh[TN], wl[TN];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm0 + 16 * i + l15;
                const int off = row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
                xh[i] = *reinterpret_cast<const bf16x8*>(sAh + off);
                xl[i] = *reinterpret_cast<const bf16x8*>(sAl + off);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn0 + 16 * j + l15;
                const int off = row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
                wh[j] = *reinterpret_cast<const bf16x8*>(sBh + off);
                wl[j] = *reinterpret_cast<const bf16x8*>(sBl + off);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], xh[i], acc0[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], xl[i], acc1[i][j], 0, 0, 0);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], xh[i], acc1[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: lane owns C[m][n .. n + 3], m = tile row l15, n = 4 lq
    const float alpha = P->alpha;
    const bool accum = (P->flags & GHN3_GEMM_ACCUM) != 0;
    float GAS* C = (float GAS*)P->C;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm0 + 16 * i + l15;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + 16 * j + 4 * lq;
            if (m >= M || n >= N) continue;
            const int64_t ci = (int64_t)m * P->ldc + n;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (acc0[i][j][e] + acc1[i][j][e]) * alpha;
            if (accum) v += *reinterpret_cast<gcf4>(C + ci);
            *reinterpret_cast<gf4>(C + ci) = v;
        }
    }
    // ---- bias gradient: db[m] += sum_k A[k][m] -- the 16 row-pair owners of a feature add up in a fixed order
    if (want_bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sbias[kp * T + f4 + e] = bsum[e];
        __syncthreads();
        if (tid < T && m0 + tid < M) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += sbias[q * T + tid];
            float GAS* db = (float GAS*)P->bias + (int64_t)(m0 + tid) * P->bias_stride;
            *db += t;                                       // (the bias gradient always accumulates, ghn3_hip.h)
        }
    }
}

// A launch may cap its grid: the workgroups then stride over the tiles.  The weight gradients run on the side stream beside
// the dependent chain; an uncapped launch (432 small workgroups per layer) back-fills every wave slot that frees up, so the
// chain's fat workgroups (768 threads, 168 VGPRs) find no empty CU until the launch has drained (r04h: up to 100 us stalls
// at the hand-offs).  Measured r04i: capped launches stretch the side stream beyond the chain (128 workgroups: +0.13 ms per
// step, 64: +0.55 ms) -- the cap stays available (op.i[3]) but the compiled programs do not use it.
template <int T>
__global__ __launch_bounds__(T * 4) void gemm_wg_kernel(const GemmProbDev* __restrict__ probs, int n_probs, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) char wg_sm[];
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        wg_tile<T>(probs, n_probs, tile, wg_sm);
        __syncthreads();                                    // the LDS images are reused by the next tile
    }
}

}  // namespace

// tile_edge: 64 (tile code 48) or 128 (tile code 49)
int ghn3_gemm_wg_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int grid_cap, int tile_edge, hipStream_t stream) {
    if (n_probs <= 0 || total_tiles <= 0) return GHN3_OK;
    const int grid = grid_cap > 0 && grid_cap < total_tiles ? grid_cap : total_tiles;
    if (tile_edge == 128) {
        static bool attr = false;
        if (!attr) {
            hipFuncSetAttribute((const void*)gemm_wg_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, wg_lds<128>());
            attr = true;
        }
        hipLaunchKernelGGL(gemm_wg_kernel<128>, dim3(grid), dim3(512), wg_lds<128>(), stream, d_probs, n_probs, total_tiles);
    } else {
        hipLaunchKernelGGL(gemm_wg_kernel<64>, dim3(grid), dim3(256), wg_lds<64>(), stream, d_probs, n_probs, total_tiles);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ghn3_set_error("wgrad x3 gemm launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    return GHN3_OK;
}
