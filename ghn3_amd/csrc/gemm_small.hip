// Latency-optimised fp32 MFMA GEMM for the small problems of the GHN-3 path (gfx950).
//
// The Graphormer GEMMs (graphormer.py:38-44,121,141 and their dgrad / wgrad) have M = B*N_nodes <= ~2000 and
// K, N in {C, 3C, 4C}: a few hundred MFLOP each, far too little to fill 256 CUs with 64x64 / 128x128 tiles,
// and with a K loop that is latency- rather than throughput-bound.  This kernel trades tile reuse for
// parallelism and short dependency chains:
//   * one workgroup = one 32 x 32 output tile; its 16 waves split the K range (split-K inside the workgroup,
//     combined through 64 KB of LDS), so a C = 384 projection is ONE 32-deep chunk (12 MFMA steps) per wave and a
//     4C = 1536 reduction three: the per-wave dependency chain is 1-3 memory round trips instead of 12-48;
//   * operands go global -> registers -> v_mfma_f32_32x32x2_f32 directly (no LDS staging): a ROW-mode operand
//     row is read as float4 by the two half-wave lanes that need it, a COL-mode operand as coalesced dwords;
// Same operand / epilogue contract as the tiled kernels (include/ghn3_hip.h), exact fp32.

#include "ghn3_internal.h"

#define GAS __attribute__((address_space(1)))
typedef const float GAS* gcf;
typedef float GAS* gf;
typedef const int GAS* gci;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 GAS* gcf4;
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ROWM GHN3_MODE_ROW
#define COLM GHN3_MODE_COL
#define KC 32            // K chunk per pipeline stage (16 MFMA steps)
#define SW 16            // waves per workgroup = K slices

__device__ __forceinline__ int s_map_row(int r, gci gather, int q, int s) {
    if (gather) r = gather[r];
    if (q > 0) {
        // (r / q) * s + r % q through a float reciprocal (exact after the correction steps for 0 <= r < 2^24, the
        // range the host enforces): ~8 VALU instructions instead of the ~40 of an integer division
        int d = (int)((float)r * __builtin_amdgcn_rcpf((float)q));
        int m = r - d * q;
        if (m < 0) { m += q; --d; } else if (m >= q) { m -= q; ++d; }
        if (m < 0) { m += q; --d; } else if (m >= q) { m -= q; ++d; }
        r = d * s + m;
    }
    return r;
}
__device__ __forceinline__ float s_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float s_gelu_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// One operand of the 32-row tile.  lane = (i = lane & 31, h = lane >> 5); MFMA step s of a chunk consumes
// element (row i, k = k0 + 2 s + h).
template <int MODE>
struct SmallOperand {
    gcf base; gci gather; int q, s, ld;
    int row, k_end, h;
    bool row_ok;
    gcf rptr;                      // ROW mode: start of this lane's row
    f32x4 raw4[KC / 4];            // ROW mode: 8 float4 per chunk
    float raw1[KC / 2];            // COL mode: 16 dwords per chunk

    __device__ __forceinline__ void init(const float* b, const int* g, int q_, int s_, int ld_, int rows, int origin,
                                         int kend, int lane) {
        base = (gcf)b; gather = (gci)g; q = q_; s = s_; ld = ld_; k_end = kend;
        row = origin + (lane & 31); h = lane >> 5;
        row_ok = row < rows;
        if (MODE == ROWM) rptr = base + (int64_t)s_map_row(row_ok ? row : 0, gather, q, s) * ld;
    }
    __device__ __forceinline__ void issue(int k0) {
        if (MODE == ROWM) {
#pragma unroll
            for (int c = 0; c < KC / 4; ++c) {
                f32x4 x = {0.f, 0.f, 0.f, 0.f};
                if (row_ok && k0 + 4 * c < k_end) x = *reinterpret_cast<gcf4>(rptr + k0 + 4 * c);
                raw4[c] = x;
            }
        } else {
#pragma unroll
            for (int st = 0; st < KC / 2; ++st) {
                const int k = k0 + 2 * st + h;
                float x = 0.f;
                if (row_ok && k < k_end) x = base[(int64_t)s_map_row(k, gather, q, s) * ld + row];
                raw1[st] = x;
            }
        }
    }
    // element consumed by MFMA step st of the chunk that starts at k0
    __device__ __forceinline__ float value(int st, int k0) const {
        if (MODE == ROWM) {
            const f32x4 x = raw4[st >> 1];
            const float lo = (st & 1) ? x.z : x.x, hi = (st & 1) ? x.w : x.y;
            const float val = h ? hi : lo;
            return (k0 + 2 * st + h < k_end) ? val : 0.f;
        }
        return raw1[st];
    }
};

template <int AM, int BMD>
__global__ __launch_bounds__(1024) void gemm_small_kernel(const GemmProbDev* __restrict__ probs, int n_probs) {
    __shared__ float red[SW][16][64];
    __shared__ float bgs[SW][64];
    int lo = 0, hi_ = n_probs - 1;
    while (lo < hi_) {
        int mid = (lo + hi_ + 1) >> 1;
        if (probs[mid].tile_start <= (int)blockIdx.x) lo = mid; else hi_ = mid - 1;
    }
    const GemmProbDev* P = probs + lo;
    const int t = blockIdx.x - P->tile_start;
    const int m0 = (t % P->tiles_m) * 32, n0 = (t / P->tiles_m) * 32;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int M = P->M, N = P->N, K = P->K;
    // K slice of this wave (multiple of 4 so that ROW-mode float4 loads stay aligned)
    const int slice = (((K + SW - 1) / SW) + 3) / 4 * 4;
    const int kb = w * slice;
    const int ke = min(K, kb + slice);

    SmallOperand<AM> oa; SmallOperand<BMD> ob;
    oa.init(P->A, P->a_gather, P->a_q, P->a_s, P->lda, M, m0, ke, lane);
    ob.init(P->B, P->b_gather, P->b_q, P->b_s, P->ldb, N, n0, ke, lane);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bg = 0.f;
    const bool do_bg = (AM == COLM) && (P->flags & GHN3_GEMM_BIASGRAD) && n0 == 0;

    // 16 waves per workgroup (4 per SIMD) hide the load latency of one another; a wave runs 1-3 chunks.
    for (int k0 = kb; k0 < ke; k0 += KC) {
        oa.issue(k0); ob.issue(k0);
#pragma unroll
        for (int st = 0; st < KC / 2; ++st) {
            const float a = oa.value(st, k0), b = ob.value(st, k0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            if (AM == COLM) bg += a;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w][r][lane] = acc[r];
    if (AM == COLM) bgs[w][lane] = bg;
    __syncthreads();

    if (do_bg && tid < 32 && m0 + tid < M) {
        float sum = 0.f;
#pragma unroll
        for (int ww = 0; ww < SW; ++ww) sum += bgs[ww][tid] + bgs[ww][tid + 32];
        gf dbias = (gf)P->bias;
        dbias[(int64_t)s_map_row(m0 + tid, (gci)P->c_gather, P->c_q, P->c_s) * P->bias_stride] += sum;
    }

    gf C = (gf)P->C;
    gcf bias = (gcf)P->bias;
    gcf residual = (gcf)P->residual;
    gcf aux_in = (gcf)P->aux_in;
    gf aux_out = (gf)P->aux_out;
    const int act = P->act, dact = P->dact;
    const bool accum = (P->flags & GHN3_GEMM_ACCUM) != 0;
    const bool use_bias = bias && !(P->flags & GHN3_GEMM_BIASGRAD);
    const float alpha = P->alpha;
    {
        const int e = tid;                      // 1024 threads = 32 x 32 outputs
        const int rl = e >> 5, cl = e & 31;
        const int row = m0 + rl, col = n0 + cl;
        if (row < M && col < N) {
            const int hh = (rl >> 2) & 1;
            const int r = (rl & 3) + 4 * (rl >> 3);
            const int ln = cl + 32 * hh;
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < SW; ++ww) v += red[ww][r][ln];
            v *= alpha;
            if (use_bias) {
                int bi = col;
                if (P->bias_q > 0) bi = (col / P->bias_q) * P->bias_s + (col % P->bias_q);
                v += bias[(int64_t)bi * P->bias_stride];
            }
            const int64_t ci = (int64_t)s_map_row(row, (gci)P->c_gather, P->c_q, P->c_s) * P->ldc + col;
            if (aux_out) aux_out[ci] = v;
            if (act == GHN3_ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == GHN3_ACT_GELU) v = s_gelu(v);
            if (dact == GHN3_DACT_RELU) v = (aux_in[ci] > 0.f) ? v : 0.f;
            else if (dact == GHN3_DACT_GELU) v *= s_gelu_grad(aux_in[ci]);
            if (residual) v += residual[ci];
            if (accum) v += C[ci];
            C[ci] = v;
        }
    }
}

typedef void (*small_fn)(const GemmProbDev*, int);
static small_fn g_small[2][2] = {
    {gemm_small_kernel<ROWM, ROWM>, gemm_small_kernel<ROWM, COLM>},
    {gemm_small_kernel<COLM, ROWM>, gemm_small_kernel<COLM, COLM>}};

int ghn3_gemm_small_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int a_mode, int b_mode,
                           hipStream_t stream) {
    if (n_probs <= 0 || total_tiles <= 0) return GHN3_OK;
    hipLaunchKernelGGL(g_small[a_mode][b_mode], dim3(total_tiles), dim3(64 * SW), 0, stream, d_probs, n_probs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ghn3_set_error("small gemm launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    return GHN3_OK;
}
