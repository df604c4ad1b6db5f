// Fused multi-head self-attention with additive edge bias for graph sizes N <= 1024 (gfx950).
//
// Replaces ghn3/graphormer.py:121-140:
//     attn = (q @ k^T) * d^-0.5 + edge_bias ; attn.masked_fill(~mask, -2**15) ; softmax ; attn @ v
// and its autograd backward.  Head dims on this path are tiny (d = 8..24) and N <= ~10^3, so the
// products run as exact fp32 VALU FMAs:
//   * scores: one query row per wavefront, one key per lane (the 64-wide slices of a score row live in
//     registers); K of one (graph, head) is staged through LDS with an odd row stride (conflict-free walks);
//     the row softmax is a wave-level reduction (2 scalars per row).
//   * P.V (and dS.K in the backward): the normalised row is parked in LDS and the wave switches to an
//     (output column e, key parity g) lane mapping, so no d-wide cross-lane reduction is needed.
//   * every global->LDS staging loop keeps 8 loads in flight per thread (latency-bound otherwise: one
//     workgroup per CU at N = 256).
// The probabilities are written once (B,H,N,N) for the backward pass.
//
// Mask semantics (quirks Q5/Q6): pair mask = valid(i) & valid(j); masked scores are set to -32768 (not
// -inf), so fully padded query rows produce a uniform distribution over all N_max keys, as the reference.

#include "ghn3_internal.h"

#define ATT_DMAX 32
#define ATT_WAVES 4

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Stage one head slice X[j][e] = src[j * row_stride + e] (j < N, e < d) into LDS with row stride lds_ld.
__device__ __forceinline__ void stage_head(float* __restrict__ dst, const float* __restrict__ src, int N, int d,
                                           int row_stride, int lds_ld, int tid) {
    const int total = N * d;
    constexpr int U = 8;
    for (int b0 = tid; b0 < total; b0 += 256 * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = b0 + u * 256;
            const int j = idx / d, e = idx - j * d;
            v[u] = idx < total ? src[(size_t)j * row_stride + e] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = b0 + u * 256;
            const int j = idx / d, e = idx - j * d;
            if (idx < total) dst[j * lds_ld + e] = v[u];
        }
    }
}

// out[r][e] = sum_j Prow[r][j] * X[j][e] for the RW rows of this wave; lane = e + 32 * g, g = key parity.
template <int RW>
__device__ __forceinline__ void rows_times_matrix(const float* __restrict__ prow, int p_ld,
                                                  const float* __restrict__ X, int x_ld, int N, int d, int lane,
                                                  float (&out)[RW]) {
    const int e = lane & 31, g = lane >> 5;
    float acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = 0.f;
    if (e < d) {
        for (int j = g; j < N; j += 2) {
            const float x = X[j * x_ld + e];
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[r] += prow[r * p_ld + j] * x;
        }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) out[r] = acc[r] + __shfl_xor(acc[r], 32, 64);
}

template <int NT, int RW, int DT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(float* __restrict__ out, const float* __restrict__ qkv,
                                                       const float* __restrict__ bias, float* __restrict__ Psave,
                                                       const int* __restrict__ n_nodes, int N, int C, int H,
                                                       float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int d = DT ? DT : C / H, ldk = d | 1;      // DT > 0: head dim known at compile time
    float* KV = sm;                                   // N x ldk (K, later V)
    float* Ps = sm + (size_t)N * ldk;                 // ATT_WAVES * RW rows of N (+1 pad)
    const int p_ld = N + 1;
    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nb = n_nodes[b];
    const float* base = qkv + (size_t)b * N * 3 * C;
    const size_t bh = ((size_t)b * H + h) * N;
    const int row0 = (blockIdx.x * ATT_WAVES + w) * RW;
    float* myP = Ps + (size_t)w * RW * p_ld;

    stage_head(KV, base + C + h * d, N, d, 3 * C, ldk, tid);
    __syncthreads();

#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int i = row0 + r;
        if (i < N) {
            float q[ATT_DMAX];
#pragma unroll
            for (int e = 0; e < ATT_DMAX; ++e) q[e] = (e < d) ? base[(size_t)i * 3 * C + h * d + e] : 0.f;
            float bv[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = lane + 64 * t;
                bv[t] = (bias && j < N) ? bias[(bh + i) * N + j] : 0.f;
            }
            float p[NT];
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = lane + 64 * t;
                float s = -INFINITY;
                if (j < N) {
                    s = 0.f;
                    const float* kr = KV + j * ldk;
#pragma unroll
                    for (int e = 0; e < ATT_DMAX; ++e)
                        if (e < d) s += q[e] * kr[e];
                    s = s * scale + bv[t];
                    if (!(i < nb && j < nb)) s = -32768.f;
                }
                p[t] = s;
                mx = fmaxf(mx, s);
            }
            mx = wave_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = lane + 64 * t;
                const float e_ = (j < N) ? __expf(p[t] - mx) : 0.f;
                p[t] = e_;
                sum += e_;
            }
            sum = wave_sum(sum);
            const float inv = 1.f / sum;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = lane + 64 * t;
                if (j < N) {
                    const float pv = p[t] * inv;
                    myP[r * p_ld + j] = pv;
                    if (Psave) Psave[(bh + i) * N + j] = pv;
                }
            }
        } else {
            for (int j = lane; j < N; j += 64) myP[r * p_ld + j] = 0.f;
        }
    }
    __syncthreads();
    stage_head(KV, base + 2 * C + h * d, N, d, 3 * C, ldk, tid);
    __syncthreads();
    float o[RW];
    rows_times_matrix<RW>(myP, p_ld, KV, ldk, N, d, lane, o);
    if (lane < d) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int i = row0 + r;
            if (i < N) out[((size_t)b * N + i) * C + h * d + lane] = o[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, pass 1 (row-wise): dP = dO V^T ; dS = P * (dP - rowsum(P*dP)) with masked entries zeroed ;
// dQ = scale * dS K ; dBias += dS ; dS stored for pass 2.
// ------------------------------------------------------------------------------------------------
template <int NT, int RW, int DT>
__global__ __launch_bounds__(256) void attn_bwd_rows_kernel(float* __restrict__ dqkv, const float* __restrict__ dO,
                                                            const float* __restrict__ qkv,
                                                            const float* __restrict__ P, float* __restrict__ dS,
                                                            float* __restrict__ dBias,
                                                            const int* __restrict__ n_nodes, int N, int C, int H,
                                                            float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int d = DT ? DT : C / H, ldk = d | 1;
    float* Ks = sm;
    float* Vs = sm + (size_t)N * ldk;
    float* Ds = Vs + (size_t)N * ldk;                 // ATT_WAVES * RW rows of dS
    const int p_ld = N + 1;
    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nb = n_nodes[b];
    const float* base = qkv + (size_t)b * N * 3 * C;
    const size_t bh = ((size_t)b * H + h) * N;
    const int row0 = (blockIdx.x * ATT_WAVES + w) * RW;
    float* myD = Ds + (size_t)w * RW * p_ld;
    stage_head(Ks, base + C + h * d, N, d, 3 * C, ldk, tid);
    stage_head(Vs, base + 2 * C + h * d, N, d, 3 * C, ldk, tid);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int i = row0 + r;
        if (i < N) {
            float g[ATT_DMAX];
#pragma unroll
            for (int e = 0; e < ATT_DMAX; ++e) g[e] = (e < d) ? dO[((size_t)b * N + i) * C + h * d + e] : 0.f;
            float pr[NT], dp[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = lane + 64 * t;
                pr[t] = (j < N) ? P[(bh + i) * N + j] : 0.f;
            }
            float delta = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = lane + 64 * t;
                float dv = 0.f;
                if (j < N) {
                    const float* vr = Vs + j * ldk;
#pragma unroll
                    for (int e = 0; e < ATT_DMAX; ++e)
                        if (e < d) dv += g[e] * vr[e];
                }
                dp[t] = dv;
                delta += pr[t] * dv;
            }
            delta = wave_sum(delta);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = lane + 64 * t;
                if (j < N) {
                    float ds = pr[t] * (dp[t] - delta);
                    if (!(i < nb && j < nb)) ds = 0.f;          // masked_fill blocks the gradient
                    myD[r * p_ld + j] = ds;
                    dS[(bh + i) * N + j] = ds;
                    if (dBias) dBias[(bh + i) * N + j] += ds;
                }
            }
        } else {
            for (int j = lane; j < N; j += 64) myD[r * p_ld + j] = 0.f;
        }
    }
    // wave-local LDS hand-off (each wave reads back only its own rows)
    __syncthreads();
    float dq[RW];
    rows_times_matrix<RW>(myD, p_ld, Ks, ldk, N, d, lane, dq);
    if (lane < d) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int i = row0 + r;
            if (i < N) dqkv[((size_t)b * N + i) * 3 * C + h * d + lane] = dq[r] * scale;
        }
    }
}

// backward, pass 2 (column-wise, one key per lane): dV[j] = sum_i P[i][j] dO[i] ; dK[j] = scale sum_i dS[i][j] Q[i].
// A block owns 64 keys; its 16 waves split the query rows (4 rows per step, loads batched) and combine their
// partial sums with LDS float atomics.
#define COLS_WAVES 16
template <int DT>
__global__ __launch_bounds__(1024) void attn_bwd_cols_kernel(float* __restrict__ dqkv, const float* __restrict__ dO,
                                                             const float* __restrict__ qkv,
                                                             const float* __restrict__ P, const float* __restrict__ dS,
                                                             int N, int C, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int d = DT ? DT : C / H;
    float* Qs = sm;                       // N x d   (row-broadcast reads)
    float* Gs = sm + (size_t)N * d;       // N x d   dO
    float* red = Gs + (size_t)N * d;      // 64 lanes x (2d+1)
    const int ld2 = 2 * d + 1;
    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* base = qkv + (size_t)b * N * 3 * C;
    const size_t bh = ((size_t)b * H + h) * N;
    {
        const int total = N * d;
        for (int idx = tid; idx < total; idx += 1024) {
            const int i = idx / d, e = idx - i * d;
            Qs[idx] = base[(size_t)i * 3 * C + h * d + e];
            Gs[idx] = dO[((size_t)b * N + i) * C + h * d + e];
        }
        for (int idx = tid; idx < 64 * ld2; idx += 1024) red[idx] = 0.f;
    }
    __syncthreads();
    const int j = blockIdx.x * 64 + lane;
    float dv[ATT_DMAX], dk[ATT_DMAX];
#pragma unroll
    for (int e = 0; e < ATT_DMAX; ++e) { dv[e] = 0.f; dk[e] = 0.f; }
    if (j < N) {
        constexpr int U = 4;
        for (int i0 = w * U; i0 < N; i0 += COLS_WAVES * U) {
            float pv[U], ds[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u;
                pv[u] = (i < N) ? P[(bh + i) * N + j] : 0.f;
                ds[u] = (i < N) ? dS[(bh + i) * N + j] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = min(i0 + u, N - 1);
                const float* qr = Qs + i * d;
                const float* gr = Gs + i * d;
#pragma unroll
                for (int e = 0; e < ATT_DMAX; ++e)
                    if (e < d) { dv[e] += pv[u] * gr[e]; dk[e] += ds[u] * qr[e]; }
            }
        }
        float* my = red + (size_t)lane * ld2;
#pragma unroll
        for (int e = 0; e < ATT_DMAX; ++e)
            if (e < d) { atomicAdd(&my[e], dv[e]); atomicAdd(&my[d + e], dk[e]); }
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * 2 * d; idx += 1024) {
        const int l = idx / (2 * d), c = idx - l * (2 * d);
        const int jj = blockIdx.x * 64 + l;
        if (jj < N) {
            const float s_ = red[(size_t)l * ld2 + c];
            if (c < d) dqkv[((size_t)b * N + jj) * 3 * C + 2 * C + h * d + c] = s_;            // dV
            else dqkv[((size_t)b * N + jj) * 3 * C + C + h * d + (c - d)] = s_ * scale;          // dK
        }
    }
}

// ------------------------------------------------------------------------------------------------
typedef void (*attn_fwd_fn)(float*, const float*, const float*, float*, const int*, int, int, int, float);
typedef void (*attn_bwd_fn)(float*, const float*, const float*, const float*, float*, float*, const int*, int, int,
                            int, float);
static const int kMaxLds = 160 * 1024;

typedef void (*attn_cols_fn)(float*, const float*, const float*, const float*, const float*, int, int, int, float);
struct AttnCfg { attn_fwd_fn fwd; attn_bwd_fn bwd; attn_cols_fn cols; int rw; };

template <int DT> static AttnCfg pick_n(int N) {
    if (N <= 256) return {attn_fwd_kernel<4, 4, DT>, attn_bwd_rows_kernel<4, 4, DT>, attn_bwd_cols_kernel<DT>, 4};
    if (N <= 512) return {attn_fwd_kernel<8, 4, DT>, attn_bwd_rows_kernel<8, 4, DT>, attn_bwd_cols_kernel<DT>, 4};
    return {attn_fwd_kernel<16, 2, DT>, attn_bwd_rows_kernel<16, 2, DT>, attn_bwd_cols_kernel<DT>, 2};
}
static AttnCfg pick(int N, int d) {
    switch (d) {                      // head dims of the released GHN-3 models: 8 (T, S), 16 (L), 24 (XL)
    case 8: return pick_n<8>(N);
    case 16: return pick_n<16>(N);
    case 24: return pick_n<24>(N);
    default: return pick_n<0>(N);
    }
}

template <int DT> static int set_attrs() {
    const void* fns[] = {(const void*)attn_fwd_kernel<4, 4, DT>, (const void*)attn_fwd_kernel<8, 4, DT>,
                         (const void*)attn_fwd_kernel<16, 2, DT>, (const void*)attn_bwd_rows_kernel<4, 4, DT>,
                         (const void*)attn_bwd_rows_kernel<8, 4, DT>, (const void*)attn_bwd_rows_kernel<16, 2, DT>,
                         (const void*)attn_bwd_cols_kernel<DT>};
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
        if (e != hipSuccess) { ghn3_set_error("attn hipFuncSetAttribute: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    }
    return GHN3_OK;
}

int ghn3_attn_init() {
    int rc = set_attrs<0>();
    if (!rc) rc = set_attrs<8>();
    if (!rc) rc = set_attrs<16>();
    if (!rc) rc = set_attrs<24>();
    return rc;
}

static int check_dims(int N, int C, int H) {
    if (H <= 0 || C % H != 0 || C / H > ATT_DMAX) {
        ghn3_set_error("attention: head dim %d/%d unsupported (max %d)", C, H, ATT_DMAX);
        return GHN3_E_LIMIT;
    }
    if (N > 1024 || N <= 0) { ghn3_set_error("attention: N=%d outside [1,1024]", N); return GHN3_E_LIMIT; }
    return GHN3_OK;
}

int ghn3_attn_fwd(float* out, const float* qkv, const float* bias, float* P, const int* n_nodes, int B, int N, int C,
                  int H, hipStream_t s) {
    int rc = check_dims(N, C, H);
    if (rc) return rc;
    const int d = C / H, ldk = d | 1;
    const AttnCfg cfg = pick(N, d);
    const size_t lds = ((size_t)N * ldk + (size_t)ATT_WAVES * cfg.rw * (N + 1)) * sizeof(float);
    if (lds > (size_t)kMaxLds) { ghn3_set_error("attention fwd: LDS %zu too large (N=%d d=%d)", lds, N, d); return GHN3_E_LIMIT; }
    const float scale = 1.0f / sqrtf((float)d);
    const int rpb = ATT_WAVES * cfg.rw;
    dim3 grid((N + rpb - 1) / rpb, H, B);
    hipLaunchKernelGGL(cfg.fwd, grid, dim3(256), lds, s, out, qkv, bias, P, n_nodes, N, C, H, scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ghn3_set_error("attn fwd launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    return GHN3_OK;
}

int ghn3_attn_bwd(float* dqkv, const float* dO, const float* qkv, const float* P, const float* O, float* dS,
                  float* dBias, const int* n_nodes, int B, int N, int C, int H, hipStream_t s) {
    (void)O;
    int rc = check_dims(N, C, H);
    if (rc) return rc;
    const int d = C / H, ldk = d | 1;
    const AttnCfg cfg = pick(N, d);
    const float scale = 1.0f / sqrtf((float)d);
    const size_t lds1 = ((size_t)2 * N * ldk + (size_t)ATT_WAVES * cfg.rw * (N + 1)) * sizeof(float);
    const size_t lds2 = ((size_t)2 * N * d + 64 * (2 * d + 1)) * sizeof(float);
    if (lds1 > (size_t)kMaxLds || lds2 > (size_t)kMaxLds) {
        ghn3_set_error("attention bwd: LDS %zu/%zu too large (N=%d d=%d)", lds1, lds2, N, d);
        return GHN3_E_LIMIT;
    }
    const int rpb = ATT_WAVES * cfg.rw;
    dim3 grid((N + rpb - 1) / rpb, H, B);
    hipLaunchKernelGGL(cfg.bwd, grid, dim3(256), lds1, s, dqkv, dO, qkv, P, dS, dBias, n_nodes, N, C, H, scale);
    dim3 grid2((N + 63) / 64, H, B);
    hipLaunchKernelGGL(cfg.cols, grid2, dim3(1024), lds2, s, dqkv, dO, qkv, P, dS, N, C, H, scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ghn3_set_error("attn bwd launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    return GHN3_OK;
}
