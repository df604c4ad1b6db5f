// Fused multi-head self-attention with additive edge bias for graph sizes N <= 1024 (gfx950).
//
// Replaces ghn3/graphormer.py:121-140:
//     attn = (q @ k^T) * d^-0.5 + edge_bias ; attn.masked_fill(~mask, -2**15) ; softmax ; attn @ v
// and its autograd backward.  Head dims on this path are tiny (d = 8..24) and N <= ~10^3, so the
// score / PV products run as fp32 VALU FMAs with one query row per wavefront and one key per lane
// (64-wide rows of the score matrix live in registers); K, then V, of one (graph, head) are staged through
// LDS with an odd row stride (conflict-free column walks).  The row softmax is a wave-level reduction.
// The probabilities are written once (B,H,N,N) for the backward pass.
//
// Mask semantics (quirks Q5/Q6): pair mask = valid(i) & valid(j); masked scores are set to -32768 (not
// -inf), so fully padded query rows produce a uniform distribution over all N_max keys, as the reference.

#include "ghn3_internal.h"

#define ATT_ROWS_PER_WAVE 4
#define ATT_ROWS_PER_BLOCK 16
#define ATT_DMAX 32

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

template <int NT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(float* __restrict__ out, const float* __restrict__ qkv,
                                                       const float* __restrict__ bias, float* __restrict__ Psave,
                                                       const int* __restrict__ n_nodes, int N, int C, int H,
                                                       float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int d = C / H, ldk = d | 1;
    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nb = n_nodes[b];
    const float* base = qkv + (size_t)b * N * 3 * C;
    const size_t bh = ((size_t)b * H + h) * N;

    for (int idx = tid; idx < N * d; idx += 256) {
        int j = idx / d, e = idx - j * d;
        sm[j * ldk + e] = base[(size_t)j * 3 * C + C + h * d + e];
    }
    __syncthreads();

    float p[ATT_ROWS_PER_WAVE][NT];
#pragma unroll
    for (int r = 0; r < ATT_ROWS_PER_WAVE; ++r) {
        const int i = blockIdx.x * ATT_ROWS_PER_BLOCK + w * ATT_ROWS_PER_WAVE + r;
        if (i < N) {
            float q[ATT_DMAX];
#pragma unroll
            for (int e = 0; e < ATT_DMAX; ++e) q[e] = (e < d) ? base[(size_t)i * 3 * C + h * d + e] : 0.f;
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = lane + 64 * t;
                float s = -INFINITY;
                if (j < N) {
                    s = 0.f;
                    const float* kr = sm + j * ldk;
#pragma unroll
                    for (int e = 0; e < ATT_DMAX; ++e)
                        if (e < d) s += q[e] * kr[e];
                    s = s * scale;
                    if (bias) s += bias[(bh + i) * N + j];
                    if (!(i < nb && j < nb)) s = -32768.f;
                }
                p[r][t] = s;
                mx = fmaxf(mx, s);
            }
            mx = wave_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = lane + 64 * t;
                float e_ = (j < N) ? __expf(p[r][t] - mx) : 0.f;
                p[r][t] = e_;
                sum += e_;
            }
            sum = wave_sum(sum);
            const float inv = 1.f / sum;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = lane + 64 * t;
                p[r][t] *= inv;
                if (Psave && j < N) Psave[(bh + i) * N + j] = p[r][t];
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < N * d; idx += 256) {
        int j = idx / d, e = idx - j * d;
        sm[j * ldk + e] = base[(size_t)j * 3 * C + 2 * C + h * d + e];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ATT_ROWS_PER_WAVE; ++r) {
        const int i = blockIdx.x * ATT_ROWS_PER_BLOCK + w * ATT_ROWS_PER_WAVE + r;
        if (i < N) {
            float acc[ATT_DMAX];
#pragma unroll
            for (int e = 0; e < ATT_DMAX; ++e) acc[e] = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = lane + 64 * t;
                if (j < N) {
                    const float* vr = sm + j * ldk;
                    const float pv = p[r][t];
#pragma unroll
                    for (int e = 0; e < ATT_DMAX; ++e)
                        if (e < d) acc[e] += pv * vr[e];
                }
            }
            float mine = 0.f;
#pragma unroll
            for (int e = 0; e < ATT_DMAX; ++e) {
                if (e < d) {
                    float s_ = wave_sum(acc[e]);
                    if (lane == e) mine = s_;
                }
            }
            if (lane < d) out[((size_t)b * N + i) * C + h * d + lane] = mine;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, pass 1 (row-wise): dP = dO V^T ; dS = P * (dP - rowsum(P*dP)) with masked entries zeroed ;
// dQ = scale * dS K ; dBias += dS ; dS stored for pass 2.
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void attn_bwd_rows_kernel(float* __restrict__ dqkv, const float* __restrict__ dO,
                                                            const float* __restrict__ qkv,
                                                            const float* __restrict__ P, float* __restrict__ dS,
                                                            float* __restrict__ dBias,
                                                            const int* __restrict__ n_nodes, int N, int C, int H,
                                                            float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int d = C / H, ldk = d | 1;
    float* Ks = sm;                 // N x ldk
    float* Vs = sm + (size_t)N * ldk;
    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nb = n_nodes[b];
    const float* base = qkv + (size_t)b * N * 3 * C;
    const size_t bh = ((size_t)b * H + h) * N;
    for (int idx = tid; idx < N * d; idx += 256) {
        int j = idx / d, e = idx - j * d;
        Ks[j * ldk + e] = base[(size_t)j * 3 * C + C + h * d + e];
        Vs[j * ldk + e] = base[(size_t)j * 3 * C + 2 * C + h * d + e];
    }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < ATT_ROWS_PER_WAVE; ++r) {
        const int i = blockIdx.x * ATT_ROWS_PER_BLOCK + w * ATT_ROWS_PER_WAVE + r;
        if (i >= N) continue;
        float g[ATT_DMAX];
#pragma unroll
        for (int e = 0; e < ATT_DMAX; ++e) g[e] = (e < d) ? dO[((size_t)b * N + i) * C + h * d + e] : 0.f;
        float pr[NT], dp[NT];
        float delta = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = lane + 64 * t;
            float pv = 0.f, dv = 0.f;
            if (j < N) {
                pv = P[(bh + i) * N + j];
                const float* vr = Vs + j * ldk;
#pragma unroll
                for (int e = 0; e < ATT_DMAX; ++e)
                    if (e < d) dv += g[e] * vr[e];
            }
            pr[t] = pv; dp[t] = dv;
            delta += pv * dv;
        }
        delta = wave_sum(delta);
        float dq[ATT_DMAX];
#pragma unroll
        for (int e = 0; e < ATT_DMAX; ++e) dq[e] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = lane + 64 * t;
            if (j < N) {
                float ds = pr[t] * (dp[t] - delta);
                if (!(i < nb && j < nb)) ds = 0.f;          // masked_fill blocks the gradient
                dS[(bh + i) * N + j] = ds;
                if (dBias) dBias[(bh + i) * N + j] += ds;
                const float* kr = Ks + j * ldk;
#pragma unroll
                for (int e = 0; e < ATT_DMAX; ++e)
                    if (e < d) dq[e] += ds * kr[e];
            }
        }
        float mine = 0.f;
#pragma unroll
        for (int e = 0; e < ATT_DMAX; ++e) {
            if (e < d) {
                float s_ = wave_sum(dq[e]);
                if (lane == e) mine = s_;
            }
        }
        if (lane < d) dqkv[((size_t)b * N + i) * 3 * C + h * d + lane] = mine * scale;
    }
}

// backward, pass 2 (column-wise, one key per lane): dV[j] = sum_i P[i][j] dO[i] ; dK[j] = scale sum_i dS[i][j] Q[i].
// A block owns 64 keys; its 4 waves split the query rows and reduce through LDS.
__global__ __launch_bounds__(256) void attn_bwd_cols_kernel(float* __restrict__ dqkv, const float* __restrict__ dO,
                                                            const float* __restrict__ qkv,
                                                            const float* __restrict__ P, const float* __restrict__ dS,
                                                            int N, int C, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int d = C / H;
    float* Qs = sm;                       // N x d   (row-broadcast reads)
    float* Gs = sm + (size_t)N * d;       // N x d   dO
    float* red = Gs + (size_t)N * d;      // 4 waves x 64 lanes x 2d
    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* base = qkv + (size_t)b * N * 3 * C;
    const size_t bh = ((size_t)b * H + h) * N;
    for (int idx = tid; idx < N * d; idx += 256) {
        int i = idx / d, e = idx - i * d;
        Qs[idx] = base[(size_t)i * 3 * C + h * d + e];
        Gs[idx] = dO[((size_t)b * N + i) * C + h * d + e];
    }
    __syncthreads();
    const int j = blockIdx.x * 64 + lane;
    float dv[ATT_DMAX], dk[ATT_DMAX];
#pragma unroll
    for (int e = 0; e < ATT_DMAX; ++e) { dv[e] = 0.f; dk[e] = 0.f; }
    if (j < N) {
        for (int i = w; i < N; i += 4) {
            const float pv = P[(bh + i) * N + j];
            const float ds = dS[(bh + i) * N + j];
            const float* qr = Qs + i * d;
            const float* gr = Gs + i * d;
#pragma unroll
            for (int e = 0; e < ATT_DMAX; ++e)
                if (e < d) { dv[e] += pv * gr[e]; dk[e] += ds * qr[e]; }
        }
    }
    const int ld2 = 2 * d + 1;
    float* my = red + ((size_t)w * 64 + lane) * ld2;
#pragma unroll
    for (int e = 0; e < ATT_DMAX; ++e)
        if (e < d) { my[e] = dv[e]; my[d + e] = dk[e]; }
    __syncthreads();
    // 256 threads: thread -> (key lane l, slice): sum over the 4 waves
    for (int idx = tid; idx < 64 * 2 * d; idx += 256) {
        const int l = idx / (2 * d), c = idx - l * (2 * d);
        const int jj = blockIdx.x * 64 + l;
        if (jj < N) {
            float s_ = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) s_ += red[((size_t)ww * 64 + l) * ld2 + c];
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

int ghn3_attn_init() {
    const void* fns[] = {(const void*)attn_fwd_kernel<4>, (const void*)attn_fwd_kernel<8>,
                         (const void*)attn_fwd_kernel<16>, (const void*)attn_bwd_rows_kernel<4>,
                         (const void*)attn_bwd_rows_kernel<8>, (const void*)attn_bwd_rows_kernel<16>,
                         (const void*)attn_bwd_cols_kernel};
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
        if (e != hipSuccess) { ghn3_set_error("attn hipFuncSetAttribute: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    }
    return GHN3_OK;
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
    const size_t lds = (size_t)N * ldk * sizeof(float);
    if (lds > (size_t)kMaxLds) { ghn3_set_error("attention fwd: LDS %zu too large", lds); return GHN3_E_LIMIT; }
    const float scale = 1.0f / sqrtf((float)d);
    dim3 grid((N + ATT_ROWS_PER_BLOCK - 1) / ATT_ROWS_PER_BLOCK, H, B);
    attn_fwd_fn fn = N <= 256 ? attn_fwd_kernel<4> : (N <= 512 ? attn_fwd_kernel<8> : attn_fwd_kernel<16>);
    hipLaunchKernelGGL(fn, grid, dim3(256), lds, s, out, qkv, bias, P, n_nodes, N, C, H, scale);
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
    const float scale = 1.0f / sqrtf((float)d);
    const size_t lds1 = (size_t)2 * N * ldk * sizeof(float);
    const size_t lds2 = ((size_t)2 * N * d + 4 * 64 * (2 * d + 1)) * sizeof(float);
    if (lds1 > (size_t)kMaxLds || lds2 > (size_t)kMaxLds) {
        ghn3_set_error("attention bwd: LDS %zu/%zu too large (N=%d d=%d)", lds1, lds2, N, d);
        return GHN3_E_LIMIT;
    }
    dim3 grid((N + ATT_ROWS_PER_BLOCK - 1) / ATT_ROWS_PER_BLOCK, H, B);
    attn_bwd_fn fn = N <= 256 ? attn_bwd_rows_kernel<4>
                              : (N <= 512 ? attn_bwd_rows_kernel<8> : attn_bwd_rows_kernel<16>);
    hipLaunchKernelGGL(fn, grid, dim3(256), lds1, s, dqkv, dO, qkv, P, dS, dBias, n_nodes, N, C, H, scale);
    dim3 grid2((N + 63) / 64, H, B);
    hipLaunchKernelGGL(attn_bwd_cols_kernel, grid2, dim3(256), lds2, s, dqkv, dO, qkv, P, dS, N, C, H, scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ghn3_set_error("attn bwd launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    return GHN3_OK;
}
