// Latency-optimised fp32 MFMA GEMM for the small problems of the GHN-3 path (gfx950).
//
// The Graphormer GEMMs (graphormer.py:38-44,121,141 and their dgrad / wgrad) have M = B*N_nodes <= ~2000 and
// K, N in {C, 3C, 4C}: a few hundred MFLOP each, far too little to fill 256 CUs with 64x64 / 128x128 tiles,
// and with a K loop that is latency- rather than throughput-bound.  This kernel trades tile reuse for
// parallelism and short dependency chains:
//   * one workgroup = one 32 x 32 output tile; its 16 waves split every 128-wide K chunk 16 ways (split-K inside
//     the workgroup, combined through 64 KB of LDS): a C = 384 projection is 3 chunks of 4 MFMA steps per wave;
//   * operand chunks (32 rows x 128 k) are fetched COOPERATIVELY with fully coalesced float4 loads (one per
//     thread and operand: a ROW-mode row contributes 512 contiguous bytes, a COL-mode k-row 128), staged in
//     double-buffered LDS and read back in MFMA layout (ROW: one conflict-free ds_read_b128 per operand and
//     chunk) -- the former direct-to-register version touched 32 cache lines per load instruction and was
//     bound by the L1 line rate;
//   * the next chunk's global loads are in flight while the current one is multiplied: one barrier per chunk.
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
#define KC 128           // K chunk per pipeline stage: 16 waves x 4 MFMA steps x 2
#define SW 16            // waves per workgroup = K slices
#define LD_ROW 132       // LDS row stride of a ROW-mode chunk image [32][KC]   (132 % 64 == 4: b128 reads conflict free)
#define LD_COL 36        // LDS row stride of a COL-mode chunk image [KC][32]
#define OP_FLOATS 4608   // floats per operand image (max(32 * 132, 128 * 36))

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

// Cooperative fetch of one operand chunk: 1024 threads x one float4.
//   ROW mode: thread -> (row = tid / 32, k = 4 (tid % 32));  COL mode: thread -> (k = tid / 8, row = 4 (tid % 8)).
template <int MODE>
struct ChunkLoader {
    gcf base; gci gather; int q, s, ld, rows, K, origin;
    gcf rptr;          // ROW: start of this thread's row
    bool row_ok;       // ROW: row inside the matrix
    int r, c;          // ROW: (row, 4-float column)   COL: (k row, 4-float row group)

    __device__ __forceinline__ void init(const float* b, const int* g, int q_, int s_, int ld_, int rows_, int K_,
                                         int origin_, int tid) {
        base = (gcf)b; gather = (gci)g; q = q_; s = s_; ld = ld_; rows = rows_; K = K_; origin = origin_;
        if (MODE == ROWM) {
            r = tid >> 5; c = (tid & 31) * 4;
            row_ok = origin + r < rows;
            rptr = base + (int64_t)s_map_row(row_ok ? origin + r : 0, gather, q, s) * ld;
        } else {
            r = tid >> 3; c = (tid & 7) * 4;
        }
    }
    __device__ __forceinline__ f32x4 load(int k0) const {
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (MODE == ROWM) {
            const int k = k0 + c;
            if (row_ok && k < K) {
                if (k + 3 < K) x = *reinterpret_cast<gcf4>(rptr + k);
                else { x.x = rptr[k]; if (k + 1 < K) x.y = rptr[k + 1]; if (k + 2 < K) x.z = rptr[k + 2]; }
            }
        } else {
            const int k = k0 + r, row = origin + c;
            if (k < K && row < rows) {
                gcf p = base + (int64_t)s_map_row(k, gather, q, s) * ld + row;
                if (row + 3 < rows) x = *reinterpret_cast<gcf4>(p);
                else { x.x = p[0]; if (row + 1 < rows) x.y = p[1]; if (row + 2 < rows) x.z = p[2]; }
            }
        }
        return x;
    }
    __device__ __forceinline__ void store(float* img, f32x4 x) const {
        if (MODE == ROWM) *reinterpret_cast<f32x4*>(img + r * LD_ROW + c) = x;
        else *reinterpret_cast<f32x4*>(img + r * LD_COL + c) = x;
    }
    // the 4 operand values of lane (i, h) for the 4 MFMA steps of wave w: k = 8 w + 4 h + step
    static __device__ __forceinline__ f32x4 fragment(const float* img, int w, int i, int h) {
        const int k = 8 * w + 4 * h;
        if (MODE == ROWM) return *reinterpret_cast<const f32x4*>(img + i * LD_ROW + k);
        f32x4 x;
        x.x = img[k * LD_COL + i]; x.y = img[(k + 1) * LD_COL + i];
        x.z = img[(k + 2) * LD_COL + i]; x.w = img[(k + 3) * LD_COL + i];
        return x;
    }
};

// LayerNorm row prologue (see the kernel).  One instance per thread; the 32 threads (tid & 31) of a row cooperate.
// K <= 512 (every released GHN-3: C <= 384): the thread's <= 4 float4 of the row are loaded ONCE, all loads in flight
// together, the statistics and the transformed values are computed from registers and handed to the K loop chunk by
// chunk (no second pass over memory, no A loads inside the loop).  Wider rows take the two-pass path.
struct LnRow {
    int kind;
    bool row_ok, writer, cached;
    float mu, rs, s1, s2;
    gcf gamma, beta, xrow, resrow;
    gf outrow;
    f32x4 o[4];
    static __device__ __forceinline__ float rsum32(float v) {       // sum over the 32 lanes that share a row
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        return v;
    }
    __device__ __forceinline__ void init(const GemmProbDev* P, gcf arow, bool ok, int row, int c0, int K, bool wr) {
        row_ok = ok; writer = wr && ok;
        gamma = (gcf)P->ln_p[0];
        mu = 0.f; rs = 1.f; s1 = 0.f; s2 = 0.f;
        cached = K <= 4 * KC;
        const float invK = 1.0f / (float)K;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (kind == 1) {
            beta = (gcf)P->ln_p[1];
            outrow = P->ln_p[4] ? (gf)P->ln_p[4] + (int64_t)row * P->lda : nullptr;
            if (cached) {
                f32x4 x[4], g[4], b[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = c0 + j * KC;
                    const bool in = ok && k < K;
                    x[j] = in ? *reinterpret_cast<gcf4>(arow + k) : z;
                    g[j] = in ? *reinterpret_cast<gcf4>(gamma + k) : z;
                    b[j] = in ? *reinterpret_cast<gcf4>(beta + k) : z;
                }
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) sum += (x[j].x + x[j].y) + (x[j].z + x[j].w);
                mu = rsum32(sum) * invK;
                float v = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (c0 + j * KC < K) {
                        const float a = x[j].x - mu, bb = x[j].y - mu, c = x[j].z - mu, d = x[j].w - mu;
                        v += (a * a + bb * bb) + (c * c + d * d);
                    }
                }
                rs = rsqrtf(rsum32(v) * invK + P->ln_eps);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = c0 + j * KC;
                    o[j] = z;
                    if (ok && k < K) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[j][e] = (x[j][e] - mu) * rs * g[j][e] + b[j][e];
                        if (writer && outrow) *reinterpret_cast<f32x4 GAS*>(outrow + k) = o[j];
                    }
                }
            } else {
                float sum = 0.f;
                for (int k = c0; k < K; k += KC)
                    if (ok) { const f32x4 x = *reinterpret_cast<gcf4>(arow + k); sum += (x.x + x.y) + (x.z + x.w); }
                mu = rsum32(sum) * invK;
                float v = 0.f;
                for (int k = c0; k < K; k += KC)
                    if (ok) {
                        const f32x4 x = *reinterpret_cast<gcf4>(arow + k);
                        const float a = x.x - mu, b = x.y - mu, c = x.z - mu, d = x.w - mu;
                        v += (a * a + b * b) + (c * c + d * d);
                    }
                rs = rsqrtf(rsum32(v) * invK + P->ln_eps);
            }
            if (writer && (c0 == 0)) {
                if (P->ln_p[2]) ((gf)P->ln_p[2])[row] = mu;
                if (P->ln_p[3]) ((gf)P->ln_p[3])[row] = rs;
            }
        } else {
            xrow = (gcf)P->ln_p[1] + (int64_t)row * P->lda;
            resrow = P->ln_p[4] ? (gcf)P->ln_p[4] + (int64_t)row * P->lda : nullptr;
            outrow = P->ln_p[5] ? (gf)P->ln_p[5] + (int64_t)row * P->lda : nullptr;
            if (ok) { mu = ((gcf)P->ln_p[2])[row]; rs = ((gcf)P->ln_p[3])[row]; }
            if (cached) {
                f32x4 dy[4], x[4], g[4], r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = c0 + j * KC;
                    const bool in = ok && k < K;
                    dy[j] = in ? *reinterpret_cast<gcf4>(arow + k) : z;
                    x[j] = in ? *reinterpret_cast<gcf4>(xrow + k) : z;
                    g[j] = in ? *reinterpret_cast<gcf4>(gamma + k) : z;
                    r[j] = (in && resrow) ? *reinterpret_cast<gcf4>(resrow + k) : z;
                }
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dg = dy[j][e] * g[j][e];                 // (zero outside the row: g = dy = 0)
                        a1 += dg;
                        a2 += dg * (x[j][e] - mu) * rs;
                    }
                s1 = rsum32(a1) * invK;
                s2 = rsum32(a2) * invK;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = c0 + j * KC;
                    o[j] = z;
                    if (ok && k < K) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float xh = (x[j][e] - mu) * rs;
                            o[j][e] = rs * (dy[j][e] * g[j][e] - s1 - xh * s2) + r[j][e];
                        }
                        if (writer && outrow) *reinterpret_cast<f32x4 GAS*>(outrow + k) = o[j];
                    }
                }
            } else {
                float a1 = 0.f, a2 = 0.f;
                for (int k = c0; k < K; k += KC)
                    if (ok) {
                        const f32x4 dy = *reinterpret_cast<gcf4>(arow + k);
                        const f32x4 x = *reinterpret_cast<gcf4>(xrow + k);
                        const f32x4 g = *reinterpret_cast<gcf4>(gamma + k);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float dg = dy[e] * g[e];
                            a1 += dg;
                            a2 += dg * (x[e] - mu) * rs;
                        }
                    }
                s1 = rsum32(a1) * invK;
                s2 = rsum32(a2) * invK;
            }
        }
    }
    // chunk `c` of this thread's row, already transformed (cached rows)
    __device__ __forceinline__ f32x4 chunk(int c) const {
        return c == 0 ? o[0] : c == 1 ? o[1] : c == 2 ? o[2] : o[3];
    }
    // v = the raw A values of columns k .. k + 3 of this thread's row (K % 4 == 0: a float4 is inside or outside)
    __device__ __forceinline__ f32x4 apply(f32x4 v, int k, int K) const {
        f32x4 r_ = {0.f, 0.f, 0.f, 0.f};
        if (!row_ok || k >= K) return r_;
        const f32x4 g = *reinterpret_cast<gcf4>(gamma + k);
        if (kind == 1) {
            const f32x4 b = *reinterpret_cast<gcf4>(beta + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) r_[e] = (v[e] - mu) * rs * g[e] + b[e];
        } else {
            const f32x4 x = *reinterpret_cast<gcf4>(xrow + k);
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (resrow) r = *reinterpret_cast<gcf4>(resrow + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (x[e] - mu) * rs;
                r_[e] = rs * (v[e] * g[e] - s1 - xh * s2) + r[e];
            }
        }
        if (writer && outrow) *reinterpret_cast<f32x4 GAS*>(outrow + k) = r_;
        return r_;
    }
};

// LN: instantiation with the LayerNorm row prologue (kept out of the plain kernel: its registers and code slowed every
// launch from 13 to 20 us when it was a run-time branch)
template <int AM, int BMD, bool LN>
__global__ __launch_bounds__(1024) void gemm_small_kernel(const GemmProbDev* __restrict__ probs, int n_probs) {
    // two stages x (A image + B image); the split-K reduction buffers alias them after the K loop
    __shared__ __attribute__((aligned(16))) float lds[4 * OP_FLOATS + SW * 64];
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(lds);         // [SW][16][64] = 16384 floats
    float (*bgs)[64] = reinterpret_cast<float (*)[64]>(lds + 4 * OP_FLOATS); // [SW][64]
    int lo = 0, hi_ = n_probs - 1;
    while (lo < hi_) {
        int mid = (lo + hi_ + 1) >> 1;
        if (probs[mid].tile_start <= (int)blockIdx.x) lo = mid; else hi_ = mid - 1;
    }
    const GemmProbDev* P = probs + lo;
    const int t = blockIdx.x - P->tile_start;
    const int m0 = (t % P->tiles_m) * 32, n0 = (t / P->tiles_m) * 32;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int M = P->M, N = P->N, K = P->K;

    ChunkLoader<AM> la; ChunkLoader<BMD> lb;
    la.init(P->A, P->a_gather, P->a_q, P->a_s, P->lda, M, K, m0, tid);
    lb.init(P->B, P->b_gather, P->b_q, P->b_s, P->ldb, N, K, n0, tid);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bg = 0.f;
    const bool do_bg = (AM == COLM) && (P->flags & GHN3_GEMM_BIASGRAD) && n0 == 0;

    // Epilogue operands of this thread's output element (bias, dact input, residual, old C) do not depend on the
    // products: they are fetched now, under the K loop, instead of as a second memory round trip at the end.
    const int e_rl = tid >> 5, e_cl = tid & 31;
    const int e_row = m0 + e_rl, e_col = n0 + e_cl;
    const bool e_ok = e_row < M && e_col < N;
    const bool accum = (P->flags & GHN3_GEMM_ACCUM) != 0;
    const bool use_bias = P->bias && !(P->flags & GHN3_GEMM_BIASGRAD);
    int64_t e_ci = 0;
    float e_bias = 0.f, e_aux = 0.f, e_res = 0.f, e_old = 0.f;
    if (e_ok) {
        e_ci = (int64_t)s_map_row(e_row, (gci)P->c_gather, P->c_q, P->c_s) * P->ldc + e_col;
        if (use_bias) {
            int bi = e_col;
            if (P->bias_q > 0) bi = (e_col / P->bias_q) * P->bias_s + (e_col % P->bias_q);
            e_bias = ((gcf)P->bias)[(int64_t)bi * P->bias_stride];
        }
        if (P->dact != GHN3_DACT_NONE) e_aux = ((gcf)P->aux_in)[e_ci];
        if (P->residual) e_res = ((gcf)P->residual)[e_ci];
        if (accum) e_old = ((gcf)P->C)[e_ci];
    }

    // Row prologue of A (ghn3_gemm_problem::ln_kind; ROW mode only): the LayerNorm forward / backward that produces
    // this operand runs here, per workgroup on its own 32 rows (32 threads per row, the same threads that stage the
    // row's chunks), instead of as a separate launch on the latency-bound chain.
    LnRow ln;
    ln.kind = (LN && AM == ROWM) ? P->ln_kind : 0;
    f32x4 ra, rb = lb.load(0);                        // (in flight under the prologue's loads and reductions)
    if (LN && AM == ROWM && ln.kind) ln.init(P, la.rptr, la.row_ok, m0 + la.r, la.c, K, n0 == 0);
    const int nchunks = (K + KC - 1) / KC;
    const bool ln_cached = LN && AM == ROWM && ln.kind && ln.cached;
    if (ln_cached) ra = ln.chunk(0);
    else {
        ra = la.load(0);
        if (LN && AM == ROWM && ln.kind) ra = ln.apply(ra, la.c, K);
    }
    la.store(lds, ra); lb.store(lds + OP_FLOATS, rb);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        float* cur = lds + (c & 1) * 2 * OP_FLOATS;
        float* nxt = lds + ((c + 1) & 1) * 2 * OP_FLOATS;
        const bool more = c + 1 < nchunks;
        if (more) {                                                                // in flight during the MFMAs
            rb = lb.load((c + 1) * KC);
            if (ln_cached) ra = ln.chunk(c + 1);
            else {
                ra = la.load((c + 1) * KC);
                if (LN && AM == ROWM && ln.kind) ra = ln.apply(ra, (c + 1) * KC + la.c, K);
            }
        }
        const f32x4 a = ChunkLoader<AM>::fragment(cur, w, li, lh);
        const f32x4 b = ChunkLoader<BMD>::fragment(cur + OP_FLOATS, w, li, lh);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        if (AM == COLM) bg += (a.x + a.y) + (a.z + a.w);
        if (more) { la.store(nxt, ra); lb.store(nxt + OP_FLOATS, rb); }
        __syncthreads();          // chunk c + 1 visible; everybody is done reading chunk c (its stage is reused next)
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
    gf aux_out = (gf)P->aux_out;
    const int act = P->act, dact = P->dact;
    if (e_ok) {                                  // 1024 threads = 32 x 32 outputs
        const int hh = (e_rl >> 2) & 1;
        const int r = (e_rl & 3) + 4 * (e_rl >> 3);
        const int ln = e_cl + 32 * hh;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < SW; ++ww) v += red[ww][r][ln];
        v = v * P->alpha + e_bias;
        if (aux_out) aux_out[e_ci] = v;
        if (act == GHN3_ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == GHN3_ACT_GELU) v = s_gelu(v);
        if (dact == GHN3_DACT_RELU) v = (e_aux > 0.f) ? v : 0.f;
        else if (dact == GHN3_DACT_GELU) v *= s_gelu_grad(e_aux);
        C[e_ci] = v + e_res + e_old;
    }
}

typedef void (*small_fn)(const GemmProbDev*, int);
static small_fn g_small[2][2] = {
    {gemm_small_kernel<ROWM, ROWM, false>, gemm_small_kernel<ROWM, COLM, false>},
    {gemm_small_kernel<COLM, ROWM, false>, gemm_small_kernel<COLM, COLM, false>}};
static small_fn g_small_ln[2] = {gemm_small_kernel<ROWM, ROWM, true>, gemm_small_kernel<ROWM, COLM, true>};

int ghn3_gemm_small_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int a_mode, int b_mode, int with_ln,
                           hipStream_t stream) {
    if (n_probs <= 0 || total_tiles <= 0) return GHN3_OK;
    small_fn fn = (with_ln && a_mode == GHN3_MODE_ROW) ? g_small_ln[b_mode] : g_small[a_mode][b_mode];
    hipLaunchKernelGGL(fn, dim3(total_tiles), dim3(64 * SW), 0, stream, d_probs, n_probs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ghn3_set_error("small gemm launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    return GHN3_OK;
}
