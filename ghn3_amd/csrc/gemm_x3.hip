// Split-bf16 ("x3") GEMM for the latency-bound Graphormer linears of GHN-3 (gfx950)  --  GHN3_GEMM_X3.
//
// The Graphormer GEMMs (graphormer.py:38-44,121,141 and their dgrad) have M = B * N_nodes rows (256 for the bench
// graph) and K, N in {C, 3C, 4C}: 0.1-0.3 GFLOP each, one dependent launch after the other.  On the exact-fp32 matrix
// instruction (v_mfma_f32_32x32x2_f32, 64 cycles per 2 k) their K loop is issue bound on the few CUs a 256-row output
// can occupy.  Here every product is taken on the 16-bit matrix cores instead, at near-fp32 accuracy:
//     a = a_hi + a_lo,  a_hi = bf16(a), a_lo = bf16(a - a_hi)          (16 significant bits)
//     a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi                          (the dropped a_lo b_lo is 2^-16 relative)
// three v_mfma_f32_16x16x32_bf16 per 32 k instead of sixteen fp32 instructions: ~5x fewer matrix-core cycles, fp32
// accumulation, the small cross terms summed in their own accumulator.  Measured against fp64: 8e-6 relative.
//
// Structure (one workgroup = one BM x BN output tile, 4 waves as 2 x 2):
//   * the weights live in HBM as persistent bf16 hi / lo copies (GHN3_OP_CAST16 + GHN3_CAST_SPLIT, refreshed only when
//     the parameters change), k-contiguous: their K slice goes global -> LDS by DMA (global_load_lds_dwordx4, no VGPR
//     round trip), ALL pieces of the slice issued up front -- the slice (<= 384 k) fits LDS next to the A tile, so
//     there is no ring and no steady-state loop: every load of the workgroup is in flight at once, which is what a
//     latency-bound launch wants;
//   * the activations are fp32 in HBM: each thread loads its part of the A tile (32-byte pieces, 8 threads cover a
//     256-byte row segment), splits it on the VALU while the weight DMA is in flight and writes hi / lo to LDS;
//   * LDS images are [k-tile][row][128 B] with the 16-byte slot s of row r holding k-chunk s ^ ((r >> 1) & 7):
//     conflict-free ds_read_b128 fragment reads for the 16-lane groups of the 16 x 16 x 32 instruction;
//   * the products are taken TRANSPOSED (D^T = W X^T: the weight fragment is the MFMA A operand), so that a lane owns
//     4 CONSECUTIVE output columns of one row: bias / residual / aux reads and the stores are 16-byte accesses;
//   * k-tiles are consumed as they land (counted s_waitcnt vmcnt + s_barrier per 64-wide k-tile);
//   * a K range longer than one slice (decoder.conv.0 forward) is walked slice by slice; K splits over workgroups
//     are separate problems whose partial planes the consuming LayerNorm op sums in a fixed order.

#include "ghn3_internal.h"

#define GAS __attribute__((address_space(1)))
#define LAS __attribute__((address_space(3)))
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const f32x4 GAS* gcf4;
typedef f32x4 GAS* gf4;
typedef const unsigned short GAS* gch;


template <int N> __device__ __forceinline__ void x3_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ float x3_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float x3_gelu_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// wait until at most `rem` k-tiles of the weight DMA (PB pieces per thread each) are still in flight (`rem` is a
// constant after unrolling: the switch folds away)
template <int PB>
__device__ __forceinline__ void x3_wait_tiles(int rem) {
    switch (rem) {
    case 0: x3_wait_vmcnt<0>(); break;
    case 1: x3_wait_vmcnt<PB>(); break;
    case 2: x3_wait_vmcnt<2 * PB>(); break;
    case 3: x3_wait_vmcnt<3 * PB>(); break;
    case 4: x3_wait_vmcnt<4 * PB>(); break;
    default: x3_wait_vmcnt<5 * PB>(); break;
    }
}

// NKT = k-tiles (64 k each) of a K slice: a template parameter, so that the staging code has no run-time branches and
// the compiler's own vmcnt bookkeeping for the A loads stays exact (with a run-time tile count it waited for the whole
// weight DMA before the first conversion).
template <int BM, int BN, int NKT>
__global__ __launch_bounds__(256) void gemm_x3_kernel(const GemmProbDev* __restrict__ probs, int n_probs) {
    extern __shared__ __attribute__((aligned(16))) char x3_smem[];
    constexpr int TM = BM / 32, TN = BN / 32;          // 16 x 16 MFMA tiles per wave (2 x 2 waves)
    constexpr int TPR = 256 / BM;                      // threads per A row
    constexpr int CPT = 8 / TPR;                       // 8-float chunks per thread and k-tile
    constexpr int PB = 2 * BN / 32;                    // weight DMA pieces per thread and k-tile (hi + lo)
    static_assert(NKT * PB <= 48, "vmcnt is a 6-bit counter");
    constexpr int nkt = NKT;

    int lo = 0, hi_ = n_probs - 1;
    while (lo < hi_) {
        const int mid = (lo + hi_ + 1) >> 1;
        if (probs[mid].tile_start <= (int)blockIdx.x) lo = mid; else hi_ = mid - 1;
    }
    const GemmProbDev* P = probs + lo;
    const int t_id = blockIdx.x - P->tile_start;
    const int n0 = (t_id % P->tiles_n) * BN, m0 = (t_id / P->tiles_n) * BM;     // column tiles fastest: the workgroups
    const int M = P->M, N = P->N, K = P->K;                                      // of an XCD share few weight tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    constexpr int slice = NKT * 64;                    // k per staging round (== P->k_chunk; K % slice == 0)

    f32x4 acc0[TM][TN], acc1[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { acc0[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // A: this thread's row and first chunk; B: this thread's DMA row / swizzled source chunk
    const int a_r = tid / TPR, a_c = tid % TPR;
    // (optional row gathers of A and C: the decoder's fc dgrad multiplies the rows of ONE grid position, nn.py:738 backward)
    int a_idx = min(m0 + a_r, M - 1);
    if (P->a_gather) a_idx = P->a_gather[a_idx];
    const float GAS* a_row = (const float GAS*)P->A + (int64_t)a_idx * P->lda;
    const int b_r = tid >> 3;                                           // + 32 j
    const int b_ck = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;                 // source k offset of this lane's 16-byte slot
    gch Bh = (gch)P->B, Bl = (gch)P->B2;

    for (int k0 = 0; k0 < K; k0 += slice) {
        char* sAh = x3_smem;
        char* sAl = sAh + nkt * BM * 128;
        char* sBh = sAl + nkt * BM * 128;
        char* sBl = sBh + nkt * BN * 128;

        // ---- 1. A tile: fp32 global loads (32 B per chunk), all issued before anything else.  Written as asm: hipcc
        // waits for ALL outstanding VMEM (vmcnt(0), i.e. the whole weight DMA issued below) before the first use of a
        // plain load's result; with asm loads the wait is ours -- vmcnt(NKT * PB) -- and the conversions run while the
        // DMA is still landing.
        f32x4 av[NKT][CPT][2];
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const float GAS* src = a_row + (k0 + t * 64 + (a_c + TPR * c) * 8);       // (K % slice == 0: inside)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(av[t][c][0]) : "v"(src) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(av[t][c][1]) : "v"(src) : "memory");
            }
        __builtin_amdgcn_sched_barrier(0);               // (hipcc otherwise sinks the DMA behind the conversions)
        // ---- 2. weight slice: LDS DMA, k-tile major (hi pieces then lo pieces of a k-tile)
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
#pragma unroll
            for (int j = 0; j < BN / 32; ++j) {
                const int row = min(n0 + b_r + 32 * j, N - 1);
                const int64_t src = (int64_t)row * P->ldb + k0 + t * 64 + b_ck;
                const int dst = t * BN * 128 + (wave * 64 + 256 * j) * 16;
                __builtin_amdgcn_global_load_lds((const void GAS*)(Bh + src), (LAS void*)(sBh + dst), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const void GAS*)(Bl + src), (LAS void*)(sBl + dst), 16, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- 3. split A into bf16 hi / lo and write the LDS images (the weight DMA is still in flight)
        x3_wait_vmcnt<NKT * PB>();                       // the A loads are older than every DMA piece: they have landed
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int c = 0; c < CPT; ++c)                // (ties the registers to the wait: no use may move above it)
                asm volatile("" : "+v"(av[t][c][0]), "+v"(av[t][c][1]));
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                bf16x8 h, l;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = av[t][c][e >> 2][e & 3];
                    const __bf16 xh = (__bf16)x;
                    h[e] = xh;
                    l[e] = (__bf16)(x - (float)xh);
                }
                const int off = t * BM * 128 + a_r * 128 + (((a_c + TPR * c) ^ ((a_r >> 1) & 7)) << 4);
                *reinterpret_cast<bf16x8*>(sAh + off) = h;
                *reinterpret_cast<bf16x8*>(sAl + off) = l;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- 4. k-tiles as they land
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            x3_wait_tiles<PB>(NKT - 1 - t);               // k-tile t has landed (this thread's pieces)
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ch = 4 * h + lq;
                bf16x8 xh[TM], xl[TM], wh[TN], wl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm0 + 16 * i + l15;
                    const int off = t * BM * 128 + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
                    xh[i] = *reinterpret_cast<const bf16x8*>(sAh + off);
                    xl[i] = *reinterpret_cast<const bf16x8*>(sAl + off);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = wn0 + 16 * j + l15;
                    const int off = t * BN * 128 + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
                    wh[j] = *reinterpret_cast<const bf16x8*>(sBh + off);
                    wl[j] = *reinterpret_cast<const bf16x8*>(sBl + off);
                }
                // (the two cross terms of a tile go to the same accumulator: issued a full round apart)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc0[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], xh[i], acc0[i][j], 0, 0, 0);
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], xl[i], acc1[i][j], 0, 0, 0);
                    }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], xh[i], acc1[i][j], 0, 0, 0);
            }
        }
        if (k0 + slice < K) __syncthreads();            // the next slice overwrites the LDS images
    }

    // ---- epilogue: lane owns C[m][n .. n + 3], m = tile row l15, n = 4 lq
    const float alpha = P->alpha;
    const int act = P->act, dact = P->dact;
    const float GAS* bias = (const float GAS*)P->bias;
    const float GAS* res = (const float GAS*)P->residual;
    const float GAS* aux_in = (const float GAS*)P->aux_in;
    float GAS* aux_out = (float GAS*)P->aux_out;
    float GAS* C = (float GAS*)P->C;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm0 + 16 * i + l15;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + 16 * j + 4 * lq;
            if (m >= M || n >= N) continue;
            const int64_t ci = (int64_t)(P->c_gather ? P->c_gather[m] : m) * P->ldc + n;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (acc0[i][j][e] + acc1[i][j][e]) * alpha;
            if (bias) { const f32x4 b = *reinterpret_cast<gcf4>(bias + n); v += b; }
            if (aux_out) *reinterpret_cast<gf4>(aux_out + ci) = v;
            if (act == GHN3_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (act == GHN3_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = x3_gelu(v[e]);
            }
            if (dact != GHN3_DACT_NONE) {
                const f32x4 a = *reinterpret_cast<gcf4>(aux_in + ci);
                if (dact == GHN3_DACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = a[e] > 0.f ? v[e] : 0.f;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= x3_gelu_grad(a[e]);
                }
            }
            if (res) { const f32x4 r = *reinterpret_cast<gcf4>(res + ci); v += r; }
            *reinterpret_cast<gf4>(C + ci) = v;
        }
    }
}

typedef void (*x3_fn)(const GemmProbDev*, int);
// [variant: 0 = 32 x 64, 1 = 64 x 64, 2 = 32 x 32][k-tiles of the slice]; nullptr: the slice does not fit the LDS
static const int g_x3_bm[3] = {32, 64, 32}, g_x3_bn[3] = {64, 64, 32};
static x3_fn g_x3[3][7] = {
    {nullptr, gemm_x3_kernel<32, 64, 1>, gemm_x3_kernel<32, 64, 2>, gemm_x3_kernel<32, 64, 3>, gemm_x3_kernel<32, 64, 4>,
     nullptr, gemm_x3_kernel<32, 64, 6>},
    {nullptr, gemm_x3_kernel<64, 64, 1>, gemm_x3_kernel<64, 64, 2>, gemm_x3_kernel<64, 64, 3>, gemm_x3_kernel<64, 64, 4>,
     nullptr, nullptr},
    {nullptr, gemm_x3_kernel<32, 32, 1>, gemm_x3_kernel<32, 32, 2>, gemm_x3_kernel<32, 32, 3>, gemm_x3_kernel<32, 32, 4>,
     nullptr, gemm_x3_kernel<32, 32, 6>},
};
static bool g_x3_ready = false;

int ghn3_gemm_x3_init() {
    if (g_x3_ready) return GHN3_OK;
    for (int v = 0; v < 3; ++v)
        for (int k = 0; k < 7; ++k) {
            if (!g_x3[v][k]) continue;
            hipError_t e = hipFuncSetAttribute((const void*)g_x3[v][k], hipFuncAttributeMaxDynamicSharedMemorySize,
                                               160 * 1024);
            if (e != hipSuccess) { ghn3_set_error("hipFuncSetAttribute(x3): %s", hipGetErrorString(e)); return GHN3_E_HIP; }
        }
    g_x3_ready = true;
    return GHN3_OK;
}

// tile code 40 / 41 / 42 -> tile edges; slice / 64 must name an instantiated kernel
int ghn3_gemm_x3_tile(int code, int slice, int* bm, int* bn) {
    if (code < 40 || code > 42 || slice <= 0 || (slice & 63) || slice / 64 > 6 || !g_x3[code - 40][slice / 64]) return 0;
    *bm = g_x3_bm[code - 40]; *bn = g_x3_bn[code - 40];
    return 1;
}

int ghn3_gemm_x3_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int code, int slice,
                        hipStream_t stream) {
    if (n_probs <= 0 || total_tiles <= 0) return GHN3_OK;
    int bm = 0, bn = 0;
    if (!ghn3_gemm_x3_tile(code, slice, &bm, &bn)) {
        ghn3_set_error("x3 gemm: no kernel for tile code %d with a K slice of %d", code, slice);
        return GHN3_E_LIMIT;
    }
    const int lds = (slice / 64) * (bm + bn) * 256;
    hipLaunchKernelGGL(g_x3[code - 40][slice / 64], dim3(total_tiles), dim3(256), lds, stream, d_probs, n_probs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ghn3_set_error("x3 gemm launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    return GHN3_OK;
}
