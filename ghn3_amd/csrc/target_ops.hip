// Target-network execution, first native slice: the dominant chain of the DeepNets-1M search space,
//
//     ReLU -> depthwise k x k convolution (stride, dilation) -> pointwise 1 x 1 convolution -> BatchNorm (batch statistics)
//
// = `DilConv` and each half of `SepConv` (/root/reference/ghn3/ops.py:198-240), forward and backward, as one op family on
// NHWC (channels-last) fp32 activations.  The reference runs it as four ATen / MIOpen modules per direction (ReLU, grouped
// conv, 1x1 conv, batch norm: ~9 passes over the activations forward, more backward) with weights the GHN predicted; here
//
//   forward   dwpw_fwd      one pass: ReLU + depthwise taps produce a [64 pixels x 32 channels] operand chunk in LDS, the
//                           pointwise product runs on the bf16 matrix cores with SPLIT operands: both factors as three bf16
//                           pieces (24 bits of mantissa) and six products -- fp32's own rounding, ~1e-7; the matrix work is
//                           free here, the op is memory- and launch-bound -- fp32 accumulate, against weight chunks converted
//                           while staged (two pieces, ~1e-5, for more than 256 output columns per workgroup: registers), the epilogue writes the pre-norm activations z once and leaves per-tile
//                           (mean, M2) of every channel;
//             bn_finalize   Chan-combines the tiles in a fixed order -> mean, rstd;   bn_apply  normalises (one pass).
//   backward  bn_bwd_partial + reduce_rows   sum dout, sum dout.xhat -> dgamma, dbeta;
//             dwpw_bwd_data  dz is formed on the fly from (dout, z, statistics) as the operand of dy = dz W_pw (matrix
//                            cores, same split arithmetic);
//             pw_wgrad       dW_pw = dz^T y over pixel chunks: dz formed on the fly again, y RE-COMPUTED from x (never
//                            stored), partial products per chunk + fixed-order reduction;
//             dw_bwd_data    dx = 1[x > 0] . transposed depthwise taps of dy;   dw_wgrad  dW_dw partials + reduction.
//
// Everything is HBM-bound (K = C_in <= 512: a few hundred flops per byte at most), so the design minimises passes: x is read
// by three kernels, z by three, nothing else of activation size except dy is materialised.  All reductions are deterministic
// (fixed-order partial slots).  Weights are read IN PLACE: w_dw [C_in][ks][ks], w_pw [C_out][C_in], gamma / beta [C_out] are
// views of the GHN's flat prediction buffer; their gradients are written densely for autograd to route back into it.
// Limits (checked by the host): C_in, C_out multiples of 4 and <= 512, ks <= 7 (odd or even), N H W C < 2^31.
// w_dw == nullptr (ks = 1): the same family as ReLU -> 1x1 convolution (stride) -> BatchNorm = `ReLUConvBN` with a 1x1 kernel,
// the preprocessing layer of every cell and the `conv_1x1` op (ops.py:180-198): the depthwise stage degenerates to the ReLU.

#include <algorithm>
#include "ghn3_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TP = 64;       // pixels per tile (4 waves x 16-row MFMA tiles)
constexpr int KC = 32;       // reduction chunk = one v_mfma_f32_16x16x32_bf16 k-step
constexpr int LDK = 40;      // LDS row pitch in 16-bit elements (80 bytes: 16-byte aligned fragments)
constexpr int MAXT = 49;     // taps (ks <= 7)

struct Desc {
    int N, H, W, C_in, C_out, ks, stride, pad, dil, Ho, Wo;
    float eps;
};

// two values at once on the hardware converter (v_cvt_pk_bf16_f32, round to nearest even as bf16_rn): low half = a
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf16_pk(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ unsigned short bf16_rn(float v) { return (unsigned short)bf16_pk(v, v); }
// v = p[0] + p[1] (+ p[2]) in bf16 pieces: 16 (24) bits of mantissa.  With S = 3 pieces and the six products
// 11 + 12 + 21 + 22 + 13 + 31 the matrix-core product carries fp32's own rounding (~1e-7); S = 2 (hi.hi + lo.hi + hi.lo) ~1e-5.
template <int S>
__device__ __forceinline__ void splitS(float v, unsigned short* base, int idx, int stride) {
    float r = v;
#pragma unroll
    for (int q = 0; q < S; ++q) {
        const unsigned short h = bf16_rn(r);
        base[q * stride + idx] = h;
        r -= __uint_as_float((unsigned)h << 16);
    }
}
// acc[e] += sum_k A[lane & 15][k] B[4 (lane >> 4) + e][k]   (B fragment first: a lane owns 4 consecutive columns of a row)
__device__ __forceinline__ f32x4 mfma_bt(u16x8 b, u16x8 a, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
}
__device__ __forceinline__ u16x8 frag(const unsigned short* base, int row, int kc) {
    return *reinterpret_cast<const u16x8*>(base + row * LDK + 8 * kc);
}

template <int S>
__device__ __forceinline__ f32x4 mma_terms(const u16x8 (&a)[S], const u16x8 (&b)[S], f32x4 acc) {
    if (S == 3) { acc = mfma_bt(b[2], a[0], acc); acc = mfma_bt(b[0], a[2], acc); acc = mfma_bt(b[1], a[1], acc); }   // smallest first
    acc = mfma_bt(b[1], a[0], acc);
    acc = mfma_bt(b[0], a[1], acc);
    acc = mfma_bt(b[0], a[0], acc);
    return acc;
}

// ReLU + depthwise taps of 8 consecutive channels c .. c + 7 at output pixel (n, oh, ow); ih0 / iw0 = input origin of the
// pixel's window; wt = taps x `wld` floats in LDS, channel cc of the chunk at wt[t * wld + cc]
__device__ __forceinline__ void dw_taps8(const float* __restrict__ x, const Desc& d, int n, int ih0, int iw0, int c,
                                         const float* wt, int wld, int cc, float (&out)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = 0.f;
    if (n < 0 || c >= d.C_in) return;
    const bool second = c + 4 < d.C_in;
    int t = 0;
    for (int kh = 0; kh < d.ks; ++kh) {
        const int ih = ih0 + kh * d.dil;
        for (int kw = 0; kw < d.ks; ++kw, ++t) {
            const int iw = iw0 + kw * d.dil;
            if (ih < 0 || ih >= d.H || iw < 0 || iw >= d.W) continue;
            const float* px = x + ((int64_t)(n * d.H + ih) * d.W + iw) * d.C_in + c;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(px);
            const f32x4 v1 = second ? *reinterpret_cast<const f32x4*>(px + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            const float* w = wt + t * wld + cc;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                out[e] = fmaf(fmaxf(v0[e], 0.f), w[e], out[e]);
                out[4 + e] = fmaf(fmaxf(v1[e], 0.f), w[4 + e], out[4 + e]);
            }
        }
    }
}

__device__ __forceinline__ void decode_pixel(const Desc& d, int p, int P, int& n, int& ih0, int& iw0) {
    if (p >= P) { n = -1; ih0 = iw0 = 0; return; }
    const int hw = d.Ho * d.Wo;
    n = p / hw;
    const int r = p - n * hw, oh = r / d.Wo, ow = r - oh * d.Wo;
    ih0 = oh * d.stride - d.pad;
    iw0 = ow * d.stride - d.pad;
}

// dz of 8 consecutive channels of output pixel p:  gamma rstd (dout - s1 / P - xhat s2 / P),  xhat = (z - mean) rstd
__device__ __forceinline__ void dz8(const float* __restrict__ dout, const float* __restrict__ z, const float* __restrict__ stats,
                                    const float* __restrict__ gamma, const float* __restrict__ s12, int C, int p, int P, int c,
                                    float (&out)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = 0.f;
    if (p >= P) return;
    const float invP = 1.f / (float)P;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int cc = c + 4 * h;
        if (cc >= C) break;
        const f32x4 g = *reinterpret_cast<const f32x4*>(dout + (int64_t)p * C + cc);
        if (!stats) {                                                      // (`dout` IS dz: tnet_dz_kernel ran before)
#pragma unroll
            for (int e = 0; e < 4; ++e) out[4 * h + e] = g[e];
            continue;
        }
        const f32x4 zz = *reinterpret_cast<const f32x4*>(z + (int64_t)p * C + cc);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(stats + cc), rs = *reinterpret_cast<const f32x4*>(stats + C + cc);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + cc);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(s12 + cc), a2 = *reinterpret_cast<const f32x4*>(s12 + C + cc);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (zz[e] - mu[e]) * rs[e];
            out[4 * h + e] = ga[e] * rs[e] * (g[e] - a1[e] * invP - xh * a2[e] * invP);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: z = pw(dw(relu(x))), per-tile channel statistics
// ---------------------------------------------------------------------------------------------------------------------
template <int NT, int S>
__global__ __launch_bounds__(256) void tnet_dwpw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w_dw,
                                                            const float* __restrict__ w_pw, float* __restrict__ z,
                                                            float* __restrict__ part, const Desc d, const int P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* pix = reinterpret_cast<int*>(smem);                                   // [3][TP]
    float* wt = reinterpret_cast<float*>(smem + 3 * TP * 4);                   // [MAXT][KC]
    unsigned short* As = reinterpret_cast<unsigned short*>(smem + 3 * TP * 4 + MAXT * KC * 4);      // [S][TP][LDK]
    unsigned short* Bs = As + S * TP * LDK;                                                          // [S][16 NT][LDK]
    float* red = reinterpret_cast<float*>(Bs + S * 16 * NT * LDK);             // [5][16 NT]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, kc = lane >> 4;
    const int tile = blockIdx.x, p0 = tile * TP, taps = d.ks * d.ks;
    if (tid < TP) {
        int n, ih0, iw0;
        decode_pixel(d, p0 + tid, P, n, ih0, iw0);
        pix[tid] = n; pix[TP + tid] = ih0; pix[2 * TP + tid] = iw0;
    }
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < d.C_in; c0 += KC) {
        __syncthreads();                                                       // (previous chunk's fragments are consumed)
        for (int i = tid; i < taps * KC; i += 256) {
            const int t = i / KC, cc = i % KC, c = c0 + cc;
            wt[t * KC + cc] = c < d.C_in ? (w_dw ? w_dw[(int64_t)c * taps + t] : 1.f) : 0.f;
        }
        for (int i = tid; i < 16 * NT * 8; i += 256) {
            const int n = i >> 3, k = (i & 7) * 4, c = c0 + k;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < d.C_out && c < d.C_in) v = *reinterpret_cast<const f32x4*>(w_pw + (int64_t)n * d.C_in + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) splitS<S>(v[e], Bs, n * LDK + k + e, 16 * NT * LDK);
        }
        __syncthreads();
        {
            const int i = tid >> 2, cc = (tid & 3) * 8;
            float y[8];
            dw_taps8(x, d, pix[i], pix[TP + i], pix[2 * TP + i], c0 + cc, wt, KC, cc, y);
#pragma unroll
            for (int e = 0; e < 8; ++e) splitS<S>(y[e], As, i * LDK + cc + e, TP * LDK);
        }
        __syncthreads();
        u16x8 af[S];
#pragma unroll
        for (int q = 0; q < S; ++q) af[q] = frag(As + q * TP * LDK, 16 * w + r16, kc);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (16 * j < d.C_out) {
                u16x8 bf[S];
#pragma unroll
                for (int q = 0; q < S; ++q) bf[q] = frag(Bs + q * 16 * NT * LDK, 16 * j + r16, kc);
                acc[j] = mma_terms<S>(af, bf, acc[j]);
            }
        }
    }
    // ---- epilogue: z, then per-tile (mean, M2) of every channel
    const int prow = p0 + 16 * w + r16;
    const bool valid = prow < P;
    const int cnt = min(TP, P - p0);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = 16 * j + 4 * kc;
        if (valid && col < d.C_out) *reinterpret_cast<f32x4*>(z + (int64_t)prow * d.C_out + col) = acc[j];
    }
    auto tile_sum = [&](bool centred) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = 16 * j + 4 * kc;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = 0.f;
                if (valid) { v = acc[j][e]; if (centred) { v -= red[4 * 16 * NT + col + e]; v *= v; } }
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                if (r16 == 0) red[w * 16 * NT + col + e] = v;
            }
        }
        __syncthreads();
    };
    tile_sum(false);
    for (int c = tid; c < 16 * NT; c += 256)
        red[4 * 16 * NT + c] = (red[c] + red[16 * NT + c] + red[2 * 16 * NT + c] + red[3 * 16 * NT + c]) / (float)cnt;
    tile_sum(true);
    for (int c = tid; c < d.C_out; c += 256) {
        part[((int64_t)tile * 2) * d.C_out + c] = red[4 * 16 * NT + c];
        part[((int64_t)tile * 2 + 1) * d.C_out + c] = red[c] + red[16 * NT + c] + red[2 * 16 * NT + c] + red[3 * 16 * NT + c];
    }
}

// (mean, M2) of the tiles -> stats[0..C) = mean, stats[C..2C) = 1 / sqrt(var + eps), stats[2C..3C) = biased variance.
// 16 channels per workgroup x 16 tile lanes; lane g combines tiles g, g + 16, ... in order, then the 16 lanes in order.
__global__ __launch_bounds__(256) void tnet_bn_finalize_kernel(const float* __restrict__ part, int n_tiles, int P, int C, float eps,
                                                               float* __restrict__ stats) {
    __shared__ float sn[16][16], sm[16][16], s2[16][16];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    float n = 0.f, mean = 0.f, M2 = 0.f;
    if (c < C)
        for (int t = g; t < n_tiles; t += 16) {
            const float nb = (float)min(TP, P - t * TP), mb = part[((int64_t)t * 2) * C + c], Mb = part[((int64_t)t * 2 + 1) * C + c];
            const float nn = n + nb, dl = mb - mean;
            mean += dl * nb / nn;
            M2 += Mb + dl * dl * n * nb / nn;
            n = nn;
        }
    sn[g][cl] = n; sm[g][cl] = mean; s2[g][cl] = M2;
    __syncthreads();
    if (g == 0 && c < C) {
        for (int q = 1; q < 16; ++q) {
            const float nb = sn[q][cl];
            if (nb > 0.f) {
                const float nn = n + nb, dl = sm[q][cl] - mean;
                mean += dl * nb / nn;
                M2 += s2[q][cl] + dl * dl * n * nb / nn;
                n = nn;
            }
        }
        const float var = M2 / n;
        stats[c] = mean;
        stats[C + c] = 1.f / sqrtf(var + eps);
        stats[2 * C + c] = var;
    }
}

__global__ __launch_bounds__(256) void tnet_bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ out, int64_t total4, int C) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i * 4) % C);
        const f32x4 v = *reinterpret_cast<const f32x4*>(z + i * 4);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(stats + c), rs = *reinterpret_cast<const f32x4*>(stats + C + c);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[e] - mu[e]) * rs[e] * ga[e] + be[e];
        *reinterpret_cast<f32x4*>(out + i * 4) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------
// per 64-pixel tile: part[tile][0][c] = sum dout, part[tile][1][c] = sum dout xhat        (C <= 1024)
__global__ __launch_bounds__(256) void tnet_bn_bwd_partial_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                                  const float* __restrict__ stats, float* __restrict__ part,
                                                                  int P, int C) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = reinterpret_cast<float*>(smem);                               // [ng][2][C]
    const int nq = C / 4, ng = max(1, 256 / nq);
    const int g = threadIdx.x / nq, q = threadIdx.x % nq, c = 4 * q;
    const int p0 = blockIdx.x * TP, p1 = min(P, p0 + TP);
    if (g < ng) {
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(stats + c), rs = *reinterpret_cast<const f32x4*>(stats + C + c);
        for (int p = p0 + g; p < p1; p += ng) {
            const f32x4 gd = *reinterpret_cast<const f32x4*>(dout + (int64_t)p * C + c);
            const f32x4 zz = *reinterpret_cast<const f32x4*>(z + (int64_t)p * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[e] += gd[e]; s2[e] += gd[e] * (zz[e] - mu[e]) * rs[e]; }
        }
        *reinterpret_cast<f32x4*>(red + (g * 2) * C + c) = s1;
        *reinterpret_cast<f32x4*>(red + (g * 2 + 1) * C + c) = s2;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        float s = 0.f;
        for (int k = 0; k < ng; ++k) s += red[k * 2 * C + i];
        part[(int64_t)blockIdx.x * 2 * C + i] = s;
    }
}

// out[map(col)] = sum_r part[r][col] in a fixed order; 64 columns per workgroup x 16 row lanes.  tr_c > 0: the columns are
// (t, c) pairs, col = t * tr_c + c, written transposed to out[c * tr_t + t].
__global__ __launch_bounds__(256) void tnet_reduce_rows_kernel(const float* __restrict__ part, int n_rows, int64_t n_cols,
                                                               float* __restrict__ out, int tr_c, int tr_t) {
    __shared__ f32x4 sm[16][16];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int64_t col = ((int64_t)blockIdx.x * 16 + cl) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (col < n_cols)
        for (int r = g; r < n_rows; r += 16) s += *reinterpret_cast<const f32x4*>(part + (int64_t)r * n_cols + col);
    sm[g][cl] = s;
    __syncthreads();
    if (g == 0 && col < n_cols) {
        for (int q = 1; q < 16; ++q) s += sm[q][cl];
        if (tr_c > 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t cc = col + e;
                out[(cc % tr_c) * tr_t + cc / tr_c] = s[e];
            }
        } else {
            *reinterpret_cast<f32x4*>(out + col) = s;
        }
    }
}

// dy [P][C_in] = dz [P][C_out] W_pw [C_out][C_in]; dz formed on the fly
template <int NT, int S>
__global__ __launch_bounds__(256) void tnet_dwpw_bwd_data_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                                 const float* __restrict__ stats, const float* __restrict__ gamma,
                                                                 const float* __restrict__ s12, const float* __restrict__ w_pw,
                                                                 float* __restrict__ dy, const Desc d, const int P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short* As = reinterpret_cast<unsigned short*>(smem);              // [S][TP][LDK]
    unsigned short* Bs = As + S * TP * LDK;                                    // [S][16 NT][LDK]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, kc = lane >> 4;
    const int p0 = blockIdx.x * TP;
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < d.C_out; c0 += KC) {
        __syncthreads();
        {   // A[i][k] = dz[p0 + i][c0 + k]
            const int i = tid >> 2, cc = (tid & 3) * 8;
            float v[8];
            dz8(dout, z, stats, gamma, s12, d.C_out, p0 + i, P, c0 + cc, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) splitS<S>(v[e], As, i * LDK + cc + e, TP * LDK);
        }
        // B[n = ci][k] = W_pw[c0 + k][n]
        for (int i = tid; i < KC * 4 * NT; i += 256) {
            const int k = i / (4 * NT), n = (i % (4 * NT)) * 4, co = c0 + k;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (co < d.C_out && n < d.C_in) v = *reinterpret_cast<const f32x4*>(w_pw + (int64_t)co * d.C_in + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) splitS<S>(v[e], Bs, (n + e) * LDK + k, 16 * NT * LDK);
        }
        __syncthreads();
        u16x8 af[S];
#pragma unroll
        for (int q = 0; q < S; ++q) af[q] = frag(As + q * TP * LDK, 16 * w + r16, kc);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (16 * j < d.C_in) {
                u16x8 bf[S];
#pragma unroll
                for (int q = 0; q < S; ++q) bf[q] = frag(Bs + q * 16 * NT * LDK, 16 * j + r16, kc);
                acc[j] = mma_terms<S>(af, bf, acc[j]);
            }
        }
    }
    const int prow = p0 + 16 * w + r16;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = 16 * j + 4 * kc;
        if (prow < P && col < d.C_in) *reinterpret_cast<f32x4*>(dy + (int64_t)prow * d.C_in + col) = acc[j];
    }
}

// part[chunk][co][ci] = sum over the chunk's pixels of dz[p][co] y[p][ci]; workgroup = (chunk, 64 co, 64 ci)
__global__ __launch_bounds__(256) void tnet_pw_wgrad_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                            const float* __restrict__ stats, const float* __restrict__ gamma,
                                                            const float* __restrict__ s12, const float* __restrict__ x,
                                                            const float* __restrict__ w_dw, float* __restrict__ part,
                                                            const Desc d, const int P, const int chunk_px) {
    constexpr int S = 3;
    __shared__ __attribute__((aligned(16))) unsigned short As[S * 64 * LDK], Bs[S * 64 * LDK];
    __shared__ float wt[MAXT * 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, kc = lane >> 4;
    const int chunk = blockIdx.x, co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64, taps = d.ks * d.ks;
    const int pa = chunk * chunk_px, pb = min(P, pa + chunk_px);
    for (int i = tid; i < taps * 64; i += 256) {
        const int t = i >> 6, cc = i & 63, c = ci0 + cc;
        wt[t * 64 + cc] = c < d.C_in ? (w_dw ? w_dw[(int64_t)c * taps + t] : 1.f) : 0.f;
    }
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int pk = pa; pk < pb; pk += KC) {
        __syncthreads();
        const int k = tid >> 3, m8 = (tid & 7) * 8, p = pk + k;
        {   // A[m = co][k = pixel]
            float v[8];
            dz8(dout, z, stats, gamma, s12, d.C_out, p < pb ? p : P, P, co0 + m8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) splitS<S>(v[e], As, (m8 + e) * LDK + k, 64 * LDK);
        }
        {   // B[n = ci][k = pixel] = y re-computed
            int n, ih0, iw0;
            decode_pixel(d, p < pb ? p : P, P, n, ih0, iw0);
            float y[8];
            dw_taps8(x, d, n, ih0, iw0, ci0 + m8, wt, 64, m8, y);
#pragma unroll
            for (int e = 0; e < 8; ++e) splitS<S>(y[e], Bs, (m8 + e) * LDK + k, 64 * LDK);
        }
        __syncthreads();
        u16x8 af[S];
#pragma unroll
        for (int q = 0; q < S; ++q) af[q] = frag(As + q * 64 * LDK, 16 * w + r16, kc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u16x8 bf[S];
#pragma unroll
            for (int q = 0; q < S; ++q) bf[q] = frag(Bs + q * 64 * LDK, 16 * j + r16, kc);
            acc[j] = mma_terms<S>(af, bf, acc[j]);
        }
    }
    const int co = co0 + 16 * w + r16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ci = ci0 + 16 * j + 4 * kc;
        if (co < d.C_out && ci < d.C_in)
            *reinterpret_cast<f32x4*>(part + ((int64_t)chunk * d.C_out + co) * d.C_in + ci) = acc[j];
    }
}

// dx[n, ih, iw, c] = 1[x > 0] sum over the taps that read this input: dy[n, oh, ow, c] w_dw[c][kh][kw]
__global__ __launch_bounds__(256) void tnet_dw_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ w_dw, float* __restrict__ dx,
                                                               const Desc d, int64_t total4) {
    const int nq = d.C_in / 4, taps = d.ks * d.ks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % nq) * 4;
        int64_t pix = i / nq;
        const int iw = (int)(pix % d.W);
        pix /= d.W;
        const int ih = (int)(pix % d.H), n = (int)(pix / d.H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int kh = 0; kh < d.ks; ++kh) {
            const int th = ih + d.pad - kh * d.dil;
            if (th < 0 || th % d.stride) continue;
            const int oh = th / d.stride;
            if (oh >= d.Ho) continue;
            for (int kw = 0; kw < d.ks; ++kw) {
                const int tw = iw + d.pad - kw * d.dil;
                if (tw < 0 || tw % d.stride) continue;
                const int ow = tw / d.stride;
                if (ow >= d.Wo) continue;
                const f32x4 g = *reinterpret_cast<const f32x4*>(dy + ((int64_t)(n * d.Ho + oh) * d.Wo + ow) * d.C_in + c);
                const int t = kh * d.ks + kw;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(g[e], w_dw ? w_dw[(int64_t)(c + e) * taps + t] : 1.f, acc[e]);
            }
        }
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = xv[e] > 0.f ? acc[e] : 0.f;
        *reinterpret_cast<f32x4*>(dx + i * 4) = acc;
    }
}

// part[chunk][t][c] = sum over the chunk's output pixels of dy[p][c] relu(x[tap t of p][c])
__global__ __launch_bounds__(256) void tnet_dw_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            float* __restrict__ part, const Desc d, const int P, const int chunk_px) {
    // Threads = (channel quad) x (pixel lane): a thread sums its channels over every PL-th pixel of the chunk for up to TG taps at
    // a time (registers), the pixel lanes are then added through LDS.  (The first version gave one (tap, channel quad) to a
    // thread and walked the chunk's 128 pixels in sequence with taps x C / 4 of the 256 threads busy: 108 us per call, 9 ms of
    // the training loop's 75 ms of GPU time per step.)
    constexpr int TG = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);                               // [PL][TG][nq]
    const int nq = d.C_in / 4, taps = d.ks * d.ks, PL = max(1, 256 / nq);
    const int cq = threadIdx.x % nq, pl = threadIdx.x / nq;
    const bool live = pl < PL;
    const int pa = blockIdx.x * chunk_px, pb = min(P, pa + chunk_px);
    const int hw = d.Ho * d.Wo;
    for (int t0 = 0; t0 < taps; t0 += TG) {
        f32x4 acc[TG];
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (live) {
            for (int p = pa + pl; p < pb; p += PL) {
                const int n = p / hw, r = p - n * hw, oh = r / d.Wo, ow = r - oh * d.Wo;
                const int ih0 = oh * d.stride - d.pad, iw0 = ow * d.stride - d.pad;
                const f32x4 g = *reinterpret_cast<const f32x4*>(dy + (int64_t)p * d.C_in + 4 * cq);
                const float* xn = x + (int64_t)n * d.H * d.W * d.C_in + 4 * cq;
#pragma unroll
                for (int tt = 0; tt < TG; ++tt) {
                    const int t = t0 + tt;
                    if (t < taps) {
                        const int kh = t / d.ks, kw = t - kh * d.ks;
                        const int ih = ih0 + kh * d.dil, iw = iw0 + kw * d.dil;
                        if (ih >= 0 && ih < d.H && iw >= 0 && iw < d.W) {
                            const f32x4 xv = *reinterpret_cast<const f32x4*>(xn + ((int64_t)ih * d.W + iw) * d.C_in);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[tt][e] = fmaf(g[e], fmaxf(xv[e], 0.f), acc[tt][e]);
                        }
                    }
                }
            }
        }
        __syncthreads();                                                       // (the previous tap group's sums have been read)
        if (live) {
#pragma unroll
            for (int tt = 0; tt < TG; ++tt) red[(pl * TG + tt) * nq + cq] = acc[tt];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < TG * nq; i += 256) {
            const int tt = i / nq, c4 = i - tt * nq, t = t0 + tt;
            if (t >= taps) continue;
            f32x4 sum = red[tt * nq + c4];
            for (int l = 1; l < PL; ++l) {
                const f32x4 v = red[(l * TG + tt) * nq + c4];
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] += v[e];
            }
            *reinterpret_cast<f32x4*>(part + ((int64_t)blockIdx.x * taps + t) * d.C_in + 4 * c4) = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6, second slice: the DENSE convolution chain  [ReLU ->] kh x kw convolution (stride, padding, dilation) -> BatchNorm
// = `ReLUConvBN` with a k x k kernel (the `conv_3x3 / 5x5 / 7x7` ops, /root/reference/ghn3/ops.py:180-198, 297) and its
// 1 x k / k x 1 halves.  Implicit GEMM on the same machinery as the depthwise / pointwise family above: a workgroup owns 64
// output pixels x all output channels, the reduction runs over (tap, 32 input channels) chunks -- the operand chunk of a tap is
// the (ReLU of the) shifted input pixels, read straight from x; the weights come from a [tap][C_out][C_in] re-pack of the
// predicted [C_out][C_in][kh][kw] tensor (one small launch: a workgroup's weight chunks are then coalesced rows instead of
// 4-byte gathers with a stride of kh kw floats) -- split operands with three bf16 pieces, fp32 accumulate, z + per-tile
// statistics in the epilogue.  Backward: dz on the fly (as above); dx by the transposed implicit GEMM over the taps that read
// an input pixel; dW per (tap, 64 x 64 block, pixel chunk) partials + fixed-order reduction, written back in the parameter's
// own [C_out][C_in][kh][kw] order.  The stock path runs these layers as MIOpen implicit-GEMM / Winograd kernels between two
// layout transposes, ~4 launches forward and ~8 backward per layer, plus ReLU and BatchNorm launches.
// ---------------------------------------------------------------------------------------------------------------------
struct CDesc {
    int N, H, W, C_in, C_out, kh, kw, sh, sw, ph, pw, dil, Ho, Wo, relu;
    float eps;
};

__global__ __launch_bounds__(256) void tnet_conv_w_repack_kernel(const float* __restrict__ w, float* __restrict__ w_r, int C_out,
                                                                 int C_in, int taps) {
    const int64_t total = (int64_t)C_out * C_in * taps, cc = (int64_t)C_out * C_in;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / cc, r = i - t * cc;                      // r = co * C_in + ci
        w_r[i] = w[r * taps + t];
    }
}

// (relu of) 8 consecutive channels c .. c + 7 of input pixel (n, ih, iw); zeros outside the image / beyond C_in
__device__ __forceinline__ void conv_in8(const float* __restrict__ x, const CDesc& d, int n, int ih, int iw, int c, float (&out)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = 0.f;
    if (n < 0 || c >= d.C_in || ih < 0 || ih >= d.H || iw < 0 || iw >= d.W) return;
    const float* px = x + ((int64_t)(n * d.H + ih) * d.W + iw) * d.C_in + c;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(px);
    const f32x4 v1 = c + 4 < d.C_in ? *reinterpret_cast<const f32x4*>(px + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        out[e] = d.relu ? fmaxf(v0[e], 0.f) : v0[e];
        out[4 + e] = d.relu ? fmaxf(v1[e], 0.f) : v1[e];
    }
}

template <int NT, int S>
__global__ __launch_bounds__(256) void tnet_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w_r,
                                                            float* __restrict__ z, float* __restrict__ part, const CDesc d, const int P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* pix = reinterpret_cast<int*>(smem);                                   // [3][TP]
    unsigned short* As = reinterpret_cast<unsigned short*>(smem + 3 * TP * 4);                       // [S][TP][LDK]
    unsigned short* Bs = As + S * TP * LDK;                                                          // [S][16 NT][LDK]
    float* red = reinterpret_cast<float*>(Bs + S * 16 * NT * LDK);             // [5][16 NT]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, kc = lane >> 4;
    const int tile = blockIdx.x, p0 = tile * TP, taps = d.kh * d.kw;
    if (tid < TP) {
        const int p = p0 + tid;
        int n = -1, ih0 = 0, iw0 = 0;
        if (p < P) {
            const int hw = d.Ho * d.Wo;
            n = p / hw;
            const int r = p - n * hw, oh = r / d.Wo, ow = r - oh * d.Wo;
            ih0 = oh * d.sh - d.ph;
            iw0 = ow * d.sw - d.pw;
        }
        pix[tid] = n; pix[TP + tid] = ih0; pix[2 * TP + tid] = iw0;
    }
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < taps; ++t) {
        const int dh = (t / d.kw) * d.dil, dw_ = (t % d.kw) * d.dil;
        const float* wt = w_r + (int64_t)t * d.C_out * d.C_in;
        for (int c0 = 0; c0 < d.C_in; c0 += KC) {
            __syncthreads();                                                   // (previous chunk's fragments are consumed; pix is written)
            for (int i = tid; i < 16 * NT * 8; i += 256) {
                const int n = i >> 3, k = (i & 7) * 4, c = c0 + k;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < d.C_out && c < d.C_in) v = *reinterpret_cast<const f32x4*>(wt + (int64_t)n * d.C_in + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) splitS<S>(v[e], Bs, n * LDK + k + e, 16 * NT * LDK);
            }
            {
                const int i = tid >> 2, cc = (tid & 3) * 8;
                float y[8];
                conv_in8(x, d, pix[i], pix[TP + i] + dh, pix[2 * TP + i] + dw_, c0 + cc, y);
#pragma unroll
                for (int e = 0; e < 8; ++e) splitS<S>(y[e], As, i * LDK + cc + e, TP * LDK);
            }
            __syncthreads();
            u16x8 af[S];
#pragma unroll
            for (int q = 0; q < S; ++q) af[q] = frag(As + q * TP * LDK, 16 * w + r16, kc);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (16 * j < d.C_out) {
                    u16x8 bf[S];
#pragma unroll
                    for (int q = 0; q < S; ++q) bf[q] = frag(Bs + q * 16 * NT * LDK, 16 * j + r16, kc);
                    acc[j] = mma_terms<S>(af, bf, acc[j]);
                }
            }
        }
    }
    // ---- epilogue: z, then per-tile (mean, M2) of every channel (as tnet_dwpw_fwd_kernel)
    const int prow = p0 + 16 * w + r16;
    const bool valid = prow < P;
    const int cnt = min(TP, P - p0);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = 16 * j + 4 * kc;
        if (valid && col < d.C_out) *reinterpret_cast<f32x4*>(z + (int64_t)prow * d.C_out + col) = acc[j];
    }
    auto tile_sum = [&](bool centred) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = 16 * j + 4 * kc;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = 0.f;
                if (valid) { v = acc[j][e]; if (centred) { v -= red[4 * 16 * NT + col + e]; v *= v; } }
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                if (r16 == 0) red[w * 16 * NT + col + e] = v;
            }
        }
        __syncthreads();
    };
    tile_sum(false);
    for (int c = tid; c < 16 * NT; c += 256)
        red[4 * 16 * NT + c] = (red[c] + red[16 * NT + c] + red[2 * 16 * NT + c] + red[3 * 16 * NT + c]) / (float)cnt;
    tile_sum(true);
    for (int c = tid; c < d.C_out; c += 256) {
        part[((int64_t)tile * 2) * d.C_out + c] = red[4 * 16 * NT + c];
        part[((int64_t)tile * 2 + 1) * d.C_out + c] = red[c] + red[16 * NT + c] + red[2 * 16 * NT + c] + red[3 * 16 * NT + c];
    }
}

// dx [P_in][C_in] = relu'(x) . sum over taps of dz[output pixel that reads this input through the tap][C_out] W_tap [C_out][C_in]
template <int NT, int S>
__global__ __launch_bounds__(256) void tnet_conv_bwd_data_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                                 const float* __restrict__ stats, const float* __restrict__ gamma,
                                                                 const float* __restrict__ s12, const float* __restrict__ w_r,
                                                                 const float* __restrict__ x, float* __restrict__ dx, const CDesc d,
                                                                 const int P_out, const int P_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* pix = reinterpret_cast<int*>(smem);                                   // [3][TP]: n, ih + ph, iw + pw of the input pixels
    unsigned short* As = reinterpret_cast<unsigned short*>(smem + 3 * TP * 4);                       // [S][TP][LDK]
    unsigned short* Bs = As + S * TP * LDK;                                                          // [S][16 NT][LDK]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, kc = lane >> 4;
    const int p0 = blockIdx.x * TP, taps = d.kh * d.kw;
    if (tid < TP) {
        const int p = p0 + tid;
        int n = -1, a = 0, b = 0;
        if (p < P_in) {
            const int hw = d.H * d.W;
            n = p / hw;
            const int r = p - n * hw, ih = r / d.W, iw = r - ih * d.W;
            a = ih + d.ph; b = iw + d.pw;
        }
        pix[tid] = n; pix[TP + tid] = a; pix[2 * TP + tid] = b;
    }
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < taps; ++t) {
        const int dh = (t / d.kw) * d.dil, dw_ = (t % d.kw) * d.dil;
        const float* wt = w_r + (int64_t)t * d.C_out * d.C_in;
        for (int c0 = 0; c0 < d.C_out; c0 += KC) {
            __syncthreads();
            {   // A[i][k] = dz[output pixel of (input pixel i, tap t)][c0 + k]
                const int i = tid >> 2, cc = (tid & 3) * 8;
                const int n = pix[i], th = pix[TP + i] - dh, tw = pix[2 * TP + i] - dw_;
                int po = P_out;                                                // (no such output pixel: zeros)
                if (n >= 0 && th >= 0 && tw >= 0 && th % d.sh == 0 && tw % d.sw == 0) {
                    const int oh = th / d.sh, ow = tw / d.sw;
                    if (oh < d.Ho && ow < d.Wo) po = (n * d.Ho + oh) * d.Wo + ow;
                }
                float v[8];
                dz8(dout, z, stats, gamma, s12, d.C_out, po, P_out, c0 + cc, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) splitS<S>(v[e], As, i * LDK + cc + e, TP * LDK);
            }
            // B[n = ci][k] = W_tap[c0 + k][n]
            for (int i = tid; i < KC * 4 * NT; i += 256) {
                const int k = i / (4 * NT), n = (i % (4 * NT)) * 4, co = c0 + k;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (co < d.C_out && n < d.C_in) v = *reinterpret_cast<const f32x4*>(wt + (int64_t)co * d.C_in + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) splitS<S>(v[e], Bs, (n + e) * LDK + k, 16 * NT * LDK);
            }
            __syncthreads();
            u16x8 af[S];
#pragma unroll
            for (int q = 0; q < S; ++q) af[q] = frag(As + q * TP * LDK, 16 * w + r16, kc);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (16 * j < d.C_in) {
                    u16x8 bf[S];
#pragma unroll
                    for (int q = 0; q < S; ++q) bf[q] = frag(Bs + q * 16 * NT * LDK, 16 * j + r16, kc);
                    acc[j] = mma_terms<S>(af, bf, acc[j]);
                }
            }
        }
    }
    const int prow = p0 + 16 * w + r16;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = 16 * j + 4 * kc;
        if (prow < P_in && col < d.C_in) {
            f32x4 v = acc[j];
            if (d.relu) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (int64_t)prow * d.C_in + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = xv[e] > 0.f ? v[e] : 0.f;
            }
            *reinterpret_cast<f32x4*>(dx + (int64_t)prow * d.C_in + col) = v;
        }
    }
}

// part[chunk][t][co][ci] = sum over the chunk's output pixels of dz[p][co] act(x[tap t of p][ci]); workgroup = (chunk, 64 co, (t, 64 ci))
__global__ __launch_bounds__(256) void tnet_conv_wgrad_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                              const float* __restrict__ stats, const float* __restrict__ gamma,
                                                              const float* __restrict__ s12, const float* __restrict__ x,
                                                              float* __restrict__ part, const CDesc d, const int P, const int chunk_px) {
    // (round 6, second version: the loads of pixel step k + 1 are in flight during the products of step k -- registers -> the
    // other LDS stage, one barrier per step; the first version loaded, cut and multiplied in sequence with two barriers)
    constexpr int S = 3;
    constexpr int STAGE = S * 64 * LDK;
    __shared__ __attribute__((aligned(16))) unsigned short As[2 * STAGE], Bs[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, kc = lane >> 4;
    const int ci_tiles = (d.C_in + 63) / 64, taps = d.kh * d.kw;
    const int chunk = blockIdx.x, co0 = blockIdx.y * 64, t = blockIdx.z / ci_tiles, ci0 = (blockIdx.z % ci_tiles) * 64;
    const int dh = (t / d.kw) * d.dil, dw_ = (t % d.kw) * d.dil;
    const int pa = chunk * chunk_px, pb = min(P, pa + chunk_px), hw = d.Ho * d.Wo;
    // Thread = (pixel k of the step, 8 consecutive channels m8 ..): lanes of a wave hold 32 consecutive pixels of two channel
    // groups.  The operands need the pixel index along k, i.e. a transposition on the way into LDS: neighbouring lanes (pixels
    // k, k + 1) exchange halves so that every lane writes 32-bit words (two pixels of one channel) -- 12 conflict-free stores per
    // operand instead of 24 two-byte ones that collided four ways (the first version's layout: 118 us per call).
    const int k = tid & 31, m8 = (tid >> 5) * 8;
    float va[8], vb[8];
    auto put = [&](const float (&v)[8], unsigned short* base) {
        const bool odd = k & 1;
        // even lanes take channels 0..3 of both pixels (k, k + 1), odd lanes channels 4..7: four floats change lanes
        float lo[4], hi[4];                                                    // values at the even / the odd pixel of the pair
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float got = __shfl_xor(odd ? v[e] : v[4 + e], 1, 64);
            lo[e] = odd ? got : v[e];
            hi[e] = odd ? v[4 + e] : got;
        }
        unsigned* row = reinterpret_cast<unsigned*>(base + (m8 + (odd ? 4 : 0)) * LDK + (k & ~1));
#pragma unroll
        for (int q = 0; q < S; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned h = bf16_pk(lo[e], hi[e]);
                row[(q * 64 * LDK + e * LDK) / 2] = h;
                lo[e] -= __uint_as_float(h << 16);
                hi[e] -= __uint_as_float(h & 0xffff0000u);
            }
        }
    };
    auto fetch = [&](int pk) {
        const int p = pk + k;
        dz8(dout, z, stats, gamma, s12, d.C_out, p < pb ? p : P, P, co0 + m8, va);       // A[m = co][k = pixel]
        int n = -1, ih = 0, iw = 0;                                                      // B[n = ci][k = pixel]: act(x) at the tap's input pixel
        if (p < pb) {
            n = p / hw;
            const int r = p - n * hw, oh = r / d.Wo, ow = r - oh * d.Wo;
            ih = oh * d.sh - d.ph + dh; iw = ow * d.sw - d.pw + dw_;
        }
        conv_in8(x, d, n, ih, iw, ci0 + m8, vb);
    };
    auto stage = [&](int buf) {
        put(va, As + buf * STAGE);
        put(vb, Bs + buf * STAGE);
    };
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    fetch(pa);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int pk = pa; pk < pb; pk += KC, buf ^= 1) {
        const bool more = pk + KC < pb;
        if (more) fetch(pk + KC);
        u16x8 af[S];
#pragma unroll
        for (int q = 0; q < S; ++q) af[q] = frag(As + buf * STAGE + q * 64 * LDK, 16 * w + r16, kc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u16x8 bf[S];
#pragma unroll
            for (int q = 0; q < S; ++q) bf[q] = frag(Bs + buf * STAGE + q * 64 * LDK, 16 * j + r16, kc);
            acc[j] = mma_terms<S>(af, bf, acc[j]);
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
    }
    const int co = co0 + 16 * w + r16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ci = ci0 + 16 * j + 4 * kc;
        if (co < d.C_out && ci < d.C_in)
            *reinterpret_cast<f32x4*>(part + (((int64_t)chunk * taps + t) * d.C_out + co) * d.C_in + ci) = acc[j];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
inline int nt_of(int C) { return C <= 64 ? 4 : C <= 128 ? 8 : C <= 256 ? 16 : 32; }
// operand pieces: three (fp32-equivalent products) up to 256 output columns per workgroup, two for the widest tiles (registers)
// (GHN3_TNET_TERMS=2: two pieces everywhere -- three products, ~1e-5 -- for the A/B of profiles/r06y_*)
inline int terms_of(int NT) {
    static const int small = getenv("GHN3_TNET_TERMS") ? atoi(getenv("GHN3_TNET_TERMS")) : 3;
    return NT <= 16 ? (small == 2 ? 2 : 3) : 2;
}
inline size_t fwd_lds(int NT) { const int S = terms_of(NT); return 3 * TP * 4 + MAXT * KC * 4 + S * TP * LDK * 2 + S * 16 * NT * LDK * 2 + 5 * 16 * NT * 4; }
inline size_t bwd_lds(int NT) { const int S = terms_of(NT); return S * TP * LDK * 2 + S * 16 * NT * LDK * 2; }

struct Plan { int P, n_tiles, pw_chunks, pw_chunk_px, dw_chunks, dw_chunk_px; };

Plan make_plan(const Desc& d) {
    Plan pl;
    pl.P = d.N * d.Ho * d.Wo;
    pl.n_tiles = (pl.P + TP - 1) / TP;
    const int blocks = ((d.C_out + 63) / 64) * ((d.C_in + 63) / 64);
    int chunks = (768 + blocks - 1) / blocks;
    chunks = std::max(1, std::min(chunks, (pl.P + KC - 1) / KC));
    pl.pw_chunk_px = ((pl.P + chunks - 1) / chunks + KC - 1) / KC * KC;
    pl.pw_chunks = (pl.P + pl.pw_chunk_px - 1) / pl.pw_chunk_px;
    pl.dw_chunk_px = std::max(16, std::min(128, ((pl.P + 255) / 256 + 15) / 16 * 16));   // >= 256 workgroups where the image allows
    pl.dw_chunks = (pl.P + pl.dw_chunk_px - 1) / pl.dw_chunk_px;
    return pl;
}

int check_desc(const ghn3_dwpw_desc* g, Desc& d) {
    if (!g) { ghn3_set_error("dwpw: null descriptor"); return GHN3_E_ARG; }
    d = Desc{g->N, g->H, g->W, g->C_in, g->C_out, g->ks, g->stride, g->pad, g->dil, g->Ho, g->Wo, g->eps};
    if (d.N <= 0 || d.H <= 0 || d.W <= 0 || d.C_in <= 0 || d.C_out <= 0 || d.ks <= 0 || d.stride <= 0 || d.dil <= 0 || d.pad < 0) {
        ghn3_set_error("dwpw: non-positive size in the descriptor");
        return GHN3_E_ARG;
    }
    if ((d.C_in & 3) || (d.C_out & 3) || d.C_in > 512 || d.C_out > 512 || d.ks > 7) {
        ghn3_set_error("dwpw: needs C_in, C_out multiples of 4 and <= 512, ks <= 7 (got %d -> %d, ks %d)", d.C_in, d.C_out, d.ks);
        return GHN3_E_LIMIT;
    }
    const int ho = (d.H + 2 * d.pad - d.dil * (d.ks - 1) - 1) / d.stride + 1, wo = (d.W + 2 * d.pad - d.dil * (d.ks - 1) - 1) / d.stride + 1;
    if (ho != d.Ho || wo != d.Wo || ho <= 0 || wo <= 0) {
        ghn3_set_error("dwpw: output size %d x %d does not match the convolution arithmetic (%d x %d)", d.Ho, d.Wo, ho, wo);
        return GHN3_E_ARG;
    }
    if ((int64_t)d.N * d.H * d.W * std::max(d.C_in, d.C_out) >= ((int64_t)1 << 31) ||
        (int64_t)d.N * d.Ho * d.Wo * std::max(d.C_in, d.C_out) >= ((int64_t)1 << 31)) {
        ghn3_set_error("dwpw: activation tensors of 2^31 elements or more are not supported");
        return GHN3_E_LIMIT;
    }
    return GHN3_OK;
}

template <typename K> int set_lds(K kern, size_t bytes) {
    // (once per kernel and process; keyed by the function's address -- every instantiation has the same pointer TYPE)
    static const void* done[32];
    static int n_done = 0;
    const void* key = (const void*)kern;
    bool seen = false;
    for (int i = 0; i < n_done; ++i) seen |= done[i] == key;
    if (bytes > 48 * 1024 && !seen) {
        if (n_done < 32) done[n_done++] = key;
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) { ghn3_set_error("hipFuncSetAttribute(dwpw): %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    }
    return GHN3_OK;
}

#define LAUNCH_CHECK(what) { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { ghn3_set_error(what ": %s", hipGetErrorString(e_)); return GHN3_E_HIP; } }

}  // namespace

extern "C" int64_t ghn3_dwpw_scratch_floats(const ghn3_dwpw_desc* g, int backward) {
    Desc d;
    if (check_desc(g, d)) return -1;
    const Plan pl = make_plan(d);
    if (!backward) return (int64_t)pl.n_tiles * 2 * d.C_out + 64;
    const int taps = d.ks * d.ks;
    return (int64_t)pl.n_tiles * 2 * d.C_out + 2 * d.C_out + (int64_t)pl.P * d.C_in +
           (int64_t)pl.pw_chunks * d.C_out * d.C_in + (int64_t)pl.dw_chunks * taps * d.C_in + 256;
}

extern "C" int ghn3_dwpw_bn_fwd(const ghn3_dwpw_desc* g, const float* x, const float* w_dw, const float* w_pw, const float* gamma,
                                const float* beta, float* z, float* out, float* stats, float* scratch, void* stream_) {
    Desc d;
    int rc = check_desc(g, d);
    if (rc) return rc;
    if (!x || !w_pw || !gamma || !beta || !z || !out || !stats || !scratch) { ghn3_set_error("dwpw fwd: null pointer"); return GHN3_E_ARG; }
    if (!w_dw && (d.ks != 1 || d.pad != 0)) { ghn3_set_error("dwpw: without depthwise weights the op is ReLU -> 1x1 conv -> norm: ks = 1, pad = 0"); return GHN3_E_ARG; }
    hipStream_t s = (hipStream_t)stream_;
    const Plan pl = make_plan(d);
    const int NT = nt_of(d.C_out);
    const size_t lds = fwd_lds(NT);
#define FWD_CASE(n, t) case 10 * n + t: rc = set_lds(tnet_dwpw_fwd_kernel<n, t>, lds); if (rc) return rc; \
        hipLaunchKernelGGL((tnet_dwpw_fwd_kernel<n, t>), dim3(pl.n_tiles), dim3(256), lds, s, x, w_dw, w_pw, z, scratch, d, pl.P); break;
    switch (10 * NT + terms_of(NT)) { FWD_CASE(4, 3) FWD_CASE(8, 3) FWD_CASE(16, 3) FWD_CASE(4, 2) FWD_CASE(8, 2) FWD_CASE(16, 2) FWD_CASE(32, 2) }
#undef FWD_CASE
    LAUNCH_CHECK("dwpw fwd")
    hipLaunchKernelGGL(tnet_bn_finalize_kernel, dim3((d.C_out + 15) / 16), dim3(256), 0, s, scratch, pl.n_tiles, pl.P, d.C_out, d.eps, stats);
    LAUNCH_CHECK("bn finalize")
    const int64_t total4 = (int64_t)pl.P * d.C_out / 4;
    hipLaunchKernelGGL(tnet_bn_apply_kernel, dim3((int)std::min<int64_t>((total4 + 255) / 256, 4096)), dim3(256), 0, s, z, stats, gamma,
                       beta, out, total4, d.C_out);
    LAUNCH_CHECK("bn apply")
    return GHN3_OK;
}

extern "C" int ghn3_dwpw_bn_bwd(const ghn3_dwpw_desc* g, const float* dout, const float* x, const float* z, const float* stats,
                                const float* w_dw, const float* w_pw, const float* gamma, float* dx, float* dw_dw, float* dw_pw,
                                float* dgamma, float* dbeta, float* scratch, void* stream_) {
    Desc d;
    int rc = check_desc(g, d);
    if (rc) return rc;
    if (!dout || !x || !z || !stats || !w_pw || !gamma || !dx || !dw_pw || !dgamma || !dbeta || !scratch || (w_dw && !dw_dw)) {
        ghn3_set_error("dwpw bwd: null pointer");
        return GHN3_E_ARG;
    }
    if (!w_dw && (d.ks != 1 || d.pad != 0)) { ghn3_set_error("dwpw: without depthwise weights ks = 1, pad = 0"); return GHN3_E_ARG; }
    hipStream_t s = (hipStream_t)stream_;
    const Plan pl = make_plan(d);
    const int taps = d.ks * d.ks;
    float* part12 = scratch;
    float* const s12_scratch = part12 + (int64_t)pl.n_tiles * 2 * d.C_out;
    // (dbeta directly followed by dgamma -- how target_ops.py lays them out -- IS the [sum dout | sum dout xhat] pair: no copies)
    const bool s12_in_place = dbeta && dgamma == dbeta + d.C_out;
    float* s12 = s12_in_place ? dbeta : s12_scratch;
    float* dy = s12_scratch + 2 * d.C_out;
    float* part_pw = dy + (int64_t)pl.P * d.C_in;
    float* part_dw = part_pw + (int64_t)pl.pw_chunks * d.C_out * d.C_in;
    // 1. dgamma / dbeta
    {
        const int nq = d.C_out / 4, ng = std::max(1, 256 / nq);
        hipLaunchKernelGGL(tnet_bn_bwd_partial_kernel, dim3(pl.n_tiles), dim3(256), (size_t)ng * 2 * d.C_out * 4, s, dout, z, stats, part12,
                           pl.P, d.C_out);
        LAUNCH_CHECK("bn bwd partial")
        hipLaunchKernelGGL(tnet_reduce_rows_kernel, dim3((2 * d.C_out / 4 + 15) / 16), dim3(256), 0, s, part12, pl.n_tiles,
                           (int64_t)2 * d.C_out, s12, 0, 0);
        LAUNCH_CHECK("bn bwd reduce")
        if (!s12_in_place) {
            hipMemcpyAsync(dbeta, s12, (size_t)d.C_out * 4, hipMemcpyDeviceToDevice, s);
            hipMemcpyAsync(dgamma, s12 + d.C_out, (size_t)d.C_out * 4, hipMemcpyDeviceToDevice, s);
        }
    }
    // 2. dy = dz W_pw
    {
        const int NT = nt_of(d.C_in);
        const size_t lds = bwd_lds(NT);
#define BWD_CASE(n, t) case 10 * n + t: rc = set_lds(tnet_dwpw_bwd_data_kernel<n, t>, lds); if (rc) return rc; \
        hipLaunchKernelGGL((tnet_dwpw_bwd_data_kernel<n, t>), dim3(pl.n_tiles), dim3(256), lds, s, dout, z, stats, gamma, s12, w_pw, dy, d, pl.P); break;
        switch (10 * NT + terms_of(NT)) { BWD_CASE(4, 3) BWD_CASE(8, 3) BWD_CASE(16, 3) BWD_CASE(4, 2) BWD_CASE(8, 2) BWD_CASE(16, 2) BWD_CASE(32, 2) }
#undef BWD_CASE
        LAUNCH_CHECK("dwpw bwd data")
    }
    // 3. dW_pw
    hipLaunchKernelGGL(tnet_pw_wgrad_kernel, dim3(pl.pw_chunks, (d.C_out + 63) / 64, (d.C_in + 63) / 64), dim3(256), 0, s, dout, z, stats,
                       gamma, s12, x, w_dw, part_pw, d, pl.P, pl.pw_chunk_px);
    LAUNCH_CHECK("pw wgrad")
    {
        const int64_t cols = (int64_t)d.C_out * d.C_in;
        hipLaunchKernelGGL(tnet_reduce_rows_kernel, dim3((int)((cols / 4 + 15) / 16)), dim3(256), 0, s, part_pw, pl.pw_chunks, cols, dw_pw, 0, 0);
        LAUNCH_CHECK("pw wgrad reduce")
    }
    // 4. dx and dW_dw
    {
        const int64_t total4 = (int64_t)d.N * d.H * d.W * d.C_in / 4;
        hipLaunchKernelGGL(tnet_dw_bwd_data_kernel, dim3((int)std::min<int64_t>((total4 + 255) / 256, 8192)), dim3(256), 0, s, dy, x, w_dw,
                           dx, d, total4);
        LAUNCH_CHECK("dw bwd data")
        if (w_dw) {
            const int nq = d.C_in / 4;
            hipLaunchKernelGGL(tnet_dw_wgrad_kernel, dim3(pl.dw_chunks), dim3(256), (size_t)std::max(1, 256 / nq) * 16 * nq * 16, s, dy, x,
                               part_dw, d, pl.P, pl.dw_chunk_px);
            LAUNCH_CHECK("dw wgrad")
            const int64_t cols = (int64_t)taps * d.C_in;
            hipLaunchKernelGGL(tnet_reduce_rows_kernel, dim3((int)((cols / 4 + 15) / 16)), dim3(256), 0, s, part_dw, pl.dw_chunks, cols,
                               dw_dw, d.C_in, taps);
            LAUNCH_CHECK("dw wgrad reduce")
        }
    }
    return GHN3_OK;
}

// ---- dense convolution family (round 6) ---------------------------------------------------------------------------------
namespace {

// ---------------------------------------------------------------------------------------------------------------------
// dense convolution, second version (round 6): the same implicit GEMM and the same three bf16 pieces per operand, but
//  * the weight pieces are cut ONCE per call (tnet_conv_w_pack_kernel: [piece][tap][row][k], k padded to the chunk) instead of by
//    every workgroup in every chunk, so the B tile is a plain 16-byte copy;
//  * the output columns are split over blockIdx.y (NT <= 8 fragments per workgroup): the small-image / wide-layer shapes of the
//    search space (4 x 4 x 256 channels: 16 pixel tiles) fill the chip;
//  * global loads of chunk i + 1 are in flight during the matrix products of chunk i (registers -> the other LDS buffer: one
//    barrier per chunk), the A pieces are stored as 16-byte vectors;
//  * the backward reads dz from a buffer written once (tnet_dz_kernel) instead of re-deriving it per tap.
// BWD = false: dst = z[P_dst = output pixels][R = C_out] from src = x;  BWD = true: dst = dx[input pixels][R = C_in] from src = dz.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tnet_conv_w_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int C_out,
                                                               int C_in, int taps, int transposed) {
    const int R = transposed ? C_in : C_out, K = transposed ? C_out : C_in, Kp = (K + KC - 1) / KC * KC;
    const int64_t plane = (int64_t)taps * R * Kp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % Kp);
        const int64_t tr = i / Kp;
        const int r = (int)(tr % R), t = (int)(tr / R);
        float v = 0.f;
        if (k < K) {
            const int co = transposed ? k : r, ci = transposed ? r : k;
            v = w[((int64_t)co * C_in + ci) * taps + t];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const unsigned short h = bf16_rn(v);
            wp[q * plane + i] = h;
            v -= __uint_as_float((unsigned)h << 16);
        }
    }
}

__global__ __launch_bounds__(256) void tnet_dz_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                      const float* __restrict__ stats, const float* __restrict__ gamma,
                                                      const float* __restrict__ s12, float* __restrict__ dz, int64_t total4, int C, int P) {
    const float invP = 1.f / (float)P;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i * 4) % C);
        const f32x4 g = *reinterpret_cast<const f32x4*>(dout + 4 * i), zz = *reinterpret_cast<const f32x4*>(z + 4 * i);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(stats + c), rs = *reinterpret_cast<const f32x4*>(stats + C + c);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(s12 + c), a2 = *reinterpret_cast<const f32x4*>(s12 + C + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (zz[e] - mu[e]) * rs[e];
            o[e] = ga[e] * rs[e] * (g[e] - a1[e] * invP - xh * a2[e] * invP);
        }
        *reinterpret_cast<f32x4*>(dz + 4 * i) = o;
    }
}

template <int NT, bool BWD>
__global__ __launch_bounds__(256) void tnet_conv2_kernel(const float* __restrict__ src, const unsigned short* __restrict__ wp,
                                                         float* __restrict__ dst, float* __restrict__ part,
                                                         const float* __restrict__ xmask, const CDesc d, const int P_dst, const int P_src) {
    constexpr int S = 3;
    constexpr int A_BUF = S * TP * LDK, B_BUF = S * 16 * NT * LDK;             // 16-bit elements per stage
    constexpr int NB = (S * 16 * NT * 4 + 255) / 256;                          // 16-byte units of the B tile per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* pix = reinterpret_cast<int*>(smem);                                   // [3][TP]
    unsigned short* As = reinterpret_cast<unsigned short*>(smem + 3 * TP * 4); // [2][S][TP][LDK]
    unsigned short* Bs = As + 2 * A_BUF;                                       // [2][S][16 NT][LDK]
    float* red = reinterpret_cast<float*>(As);                                 // epilogue (forward statistics): [5][16 NT], aliased
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, kc = lane >> 4;
    const int tile = blockIdx.x, p0 = tile * TP, taps = d.kh * d.kw, col0 = blockIdx.y * 16 * NT;
    const int R = BWD ? d.C_in : d.C_out, K = BWD ? d.C_out : d.C_in, Kp = (K + KC - 1) / KC * KC, nchunk = Kp / KC;
    const int64_t plane = (int64_t)taps * R * Kp;
    if (tid < TP) {
        const int p = p0 + tid;
        int n = -1, a = 0, b = 0;
        if (p < P_dst) {
            if (BWD) {
                const int hw = d.H * d.W;
                n = p / hw;
                const int r = p - n * hw, ih = r / d.W, iw = r - ih * d.W;
                a = ih + d.ph; b = iw + d.pw;
            } else {
                const int hw = d.Ho * d.Wo;
                n = p / hw;
                const int r = p - n * hw, oh = r / d.Wo, ow = r - oh * d.Wo;
                a = oh * d.sh - d.ph; b = ow * d.sw - d.pw;
            }
        }
        pix[tid] = n; pix[TP + tid] = a; pix[2 * TP + tid] = b;
    }
    __syncthreads();
    const int ai = tid >> 2, acc_ = (tid & 3) * 8;                             // this thread's A row (pixel) and k offset
    const int an = pix[ai], aa = pix[TP + ai], ab = pix[2 * TP + ai];
    f32x4 a0, a1;                                                              // prefetched A values (8 floats)
    u16x8 breg[NB];                                                            // prefetched B units
    auto fetch = [&](int it) {
        const int t = it / nchunk, c0 = (it - t * nchunk) * KC;
        const int dh = (t / d.kw) * d.dil, dw_ = (t % d.kw) * d.dil;
        int64_t ps = -1;
        if (an >= 0) {
            if (BWD) {
                const int th = aa - dh, tw = ab - dw_;
                if (th >= 0 && tw >= 0 && th % d.sh == 0 && tw % d.sw == 0) {
                    const int oh = th / d.sh, ow = tw / d.sw;
                    if (oh < d.Ho && ow < d.Wo) ps = ((int64_t)an * d.Ho + oh) * d.Wo + ow;
                }
            } else {
                const int ih = aa + dh, iw = ab + dw_;
                if (ih >= 0 && ih < d.H && iw >= 0 && iw < d.W) ps = ((int64_t)an * d.H + ih) * d.W + iw;
            }
        }
        const int c = c0 + acc_;
        a0 = f32x4{0.f, 0.f, 0.f, 0.f}; a1 = a0;
        if (ps >= 0 && c < K) {
            const float* px = src + ps * K + c;
            a0 = *reinterpret_cast<const f32x4*>(px);
            if (c + 4 < K) a1 = *reinterpret_cast<const f32x4*>(px + 4);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = tid + 256 * u;                                       // unit index: [piece][row][4 segments]
            const int q = i / (64 * NT), r = (i - q * 64 * NT) >> 2, seg = i & 3;
            u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (i < S * 64 * NT && col0 + r < R)
                v = *reinterpret_cast<const u16x8*>(wp + q * plane + ((int64_t)t * R + col0 + r) * Kp + c0 + 8 * seg);
            breg[u] = v;
        }
    };
    auto stage = [&](int buf) {                                                // registers -> LDS stage `buf`
        unsigned short* A = As + buf * A_BUF;
        unsigned short* B = Bs + buf * B_BUF;
        float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        if (!BWD && d.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
#pragma unroll
        for (int q = 0; q < S; ++q) {
            unsigned h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = bf16_pk(v[2 * e], v[2 * e + 1]);
                v[2 * e] -= __uint_as_float(h[e] << 16);
                v[2 * e + 1] -= __uint_as_float(h[e] & 0xffff0000u);
            }
            *reinterpret_cast<uint4*>(A + q * TP * LDK + ai * LDK + acc_) = uint4{h[0], h[1], h[2], h[3]};
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = tid + 256 * u;
            const int q = i / (64 * NT), r = (i - q * 64 * NT) >> 2, seg = i & 3;
            if (i < S * 64 * NT) *reinterpret_cast<u16x8*>(B + q * 16 * NT * LDK + r * LDK + 8 * seg) = breg[u];
        }
    };
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int n_it = taps * nchunk;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int it = 0; it < n_it; ++it) {
        const int buf = it & 1;
        if (it + 1 < n_it) fetch(it + 1);                                      // (in flight during the products below)
        {
            const unsigned short* A = As + buf * A_BUF;
            const unsigned short* B = Bs + buf * B_BUF;
            u16x8 af[S];
#pragma unroll
            for (int q = 0; q < S; ++q) af[q] = frag(A + q * TP * LDK, 16 * w + r16, kc);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                u16x8 bf[S];
#pragma unroll
                for (int q = 0; q < S; ++q) bf[q] = frag(B + q * 16 * NT * LDK, 16 * j + r16, kc);
                acc[j] = mma_terms<S>(af, bf, acc[j]);
            }
        }
        if (it + 1 < n_it) stage(buf ^ 1);
        __syncthreads();
    }
    const int prow = p0 + 16 * w + r16;
    const bool valid = prow < P_dst;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = col0 + 16 * j + 4 * kc;
        if (valid && col < R) {
            f32x4 v = acc[j];
            if (BWD && d.relu) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xmask + (int64_t)prow * R + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = xv[e] > 0.f ? v[e] : 0.f;
            }
            *reinterpret_cast<f32x4*>(dst + (int64_t)prow * R + col) = v;
        }
    }
    if (BWD) return;
    // ---- forward: per-tile (mean, M2) of every channel of this column group (as tnet_conv_fwd_kernel)
    const int cnt = min(TP, P_dst - p0);
    auto tile_sum = [&](bool centred) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = 16 * j + 4 * kc;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = 0.f;
                if (valid) { v = acc[j][e]; if (centred) { v -= red[4 * 16 * NT + col + e]; v *= v; } }
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                if (r16 == 0) red[w * 16 * NT + col + e] = v;
            }
        }
        __syncthreads();
    };
    tile_sum(false);
    for (int c = tid; c < 16 * NT; c += 256)
        red[4 * 16 * NT + c] = (red[c] + red[16 * NT + c] + red[2 * 16 * NT + c] + red[3 * 16 * NT + c]) / (float)cnt;
    tile_sum(true);
    for (int c = tid; c < 16 * NT; c += 256) {
        if (col0 + c < R) {
            part[((int64_t)tile * 2) * R + col0 + c] = red[4 * 16 * NT + c];
            part[((int64_t)tile * 2 + 1) * R + col0 + c] = red[c] + red[16 * NT + c] + red[2 * 16 * NT + c] + red[3 * 16 * NT + c];
        }
    }
}

struct CPlan { int P, P_in, n_tiles, n_tiles_in, w_chunks, w_chunk_px, taps; };

CPlan make_cplan(const CDesc& d) {
    CPlan pl;
    pl.taps = d.kh * d.kw;
    pl.P = d.N * d.Ho * d.Wo;
    pl.P_in = d.N * d.H * d.W;
    pl.n_tiles = (pl.P + TP - 1) / TP;
    pl.n_tiles_in = (pl.P_in + TP - 1) / TP;
    const int blocks = ((d.C_out + 63) / 64) * ((d.C_in + 63) / 64) * pl.taps;
    int chunks = (768 + blocks - 1) / blocks;
    chunks = std::max(1, std::min(chunks, (pl.P + KC - 1) / KC));
    pl.w_chunk_px = ((pl.P + chunks - 1) / chunks + KC - 1) / KC * KC;
    pl.w_chunks = (pl.P + pl.w_chunk_px - 1) / pl.w_chunk_px;
    return pl;
}

inline bool conv2_on() {
    static const bool on = !(getenv("GHN3_TNET_CONV2") && atoi(getenv("GHN3_TNET_CONV2")) == 0);
    return on;
}

int check_cdesc(const ghn3_conv_desc* g, CDesc& d) {
    if (!g) { ghn3_set_error("conv: null descriptor"); return GHN3_E_ARG; }
    d = CDesc{g->N, g->H, g->W, g->C_in, g->C_out, g->kh, g->kw, g->stride_h, g->stride_w, g->pad_h, g->pad_w, g->dil, g->Ho, g->Wo,
              (g->relu & 1) != 0, g->eps};
    if (d.N <= 0 || d.H <= 0 || d.W <= 0 || d.C_in <= 0 || d.C_out <= 0 || d.kh <= 0 || d.kw <= 0 || d.sh <= 0 || d.sw <= 0 ||
        d.dil <= 0 || d.ph < 0 || d.pw < 0) {
        ghn3_set_error("conv: non-positive size in the descriptor");
        return GHN3_E_ARG;
    }
    // (the second-version kernels walk C_in in chunks and split it over blockIdx.y in the backward: wide inputs are fine)
    if ((d.C_in & 3) || (d.C_out & 3) || d.C_in > (conv2_on() ? 4096 : 512) || d.C_out > 512 || d.kh > 7 || d.kw > 7) {
        ghn3_set_error("conv: needs C_in, C_out multiples of 4, C_out <= 512, C_in <= 4096 (512 with GHN3_TNET_CONV2=0), kernel <= 7 x 7 "
                       "(got %d -> %d, %d x %d)", d.C_in, d.C_out, d.kh, d.kw);
        return GHN3_E_LIMIT;
    }
    const int ho = (d.H + 2 * d.ph - d.dil * (d.kh - 1) - 1) / d.sh + 1, wo = (d.W + 2 * d.pw - d.dil * (d.kw - 1) - 1) / d.sw + 1;
    if (ho != d.Ho || wo != d.Wo || ho <= 0 || wo <= 0) {
        ghn3_set_error("conv: output size %d x %d does not match the convolution arithmetic (%d x %d)", d.Ho, d.Wo, ho, wo);
        return GHN3_E_ARG;
    }
    if ((int64_t)d.N * d.H * d.W * std::max(d.C_in, d.C_out) >= ((int64_t)1 << 31) ||
        (int64_t)d.N * d.Ho * d.Wo * std::max(d.C_in, d.C_out) >= ((int64_t)1 << 31)) {
        ghn3_set_error("conv: activation tensors of 2^31 elements or more are not supported");
        return GHN3_E_LIMIT;
    }
    return GHN3_OK;
}

inline size_t cfwd_lds(int NT) { const int S = terms_of(NT); return 3 * TP * 4 + S * TP * LDK * 2 + S * 16 * NT * LDK * 2 + 5 * 16 * NT * 4; }
inline size_t cbwd_lds(int NT) { const int S = terms_of(NT); return 3 * TP * 4 + S * TP * LDK * 2 + S * 16 * NT * LDK * 2; }

int conv_repack(const CDesc& d, const CPlan& pl, const float* w, float* w_r, hipStream_t s) {
    const int64_t total = (int64_t)d.C_out * d.C_in * pl.taps;
    hipLaunchKernelGGL(tnet_conv_w_repack_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 2048)), dim3(256), 0, s, w, w_r, d.C_out,
                       d.C_in, pl.taps);
    LAUNCH_CHECK("conv weight repack")
    return GHN3_OK;
}

}  // namespace

// 16-bit weight pieces of the second-version kernels, in floats: [3][taps][rows][k padded to the chunk]
inline int64_t pack_floats(int taps, int rows, int k) { return (((int64_t)3 * taps * rows * ((k + KC - 1) / KC * KC) + 1) / 2 + 3) / 4 * 4; }
// columns per workgroup of the second-version kernels: as wide as possible while the grid still covers the chip
inline int conv2_nt(int tiles, int cols) {
    int nt = cols <= 32 ? 2 : cols <= 64 ? 4 : 8;
    while (nt > 2 && (int64_t)tiles * ((cols + 16 * nt - 1) / (16 * nt)) < 256) nt >>= 1;
    return nt;
}
inline size_t conv2_lds(int NT) { return 3 * TP * 4 + 2 * (3 * TP * LDK + 3 * 16 * NT * LDK) * 2; }

extern "C" int64_t ghn3_conv_scratch_floats(const ghn3_conv_desc* g, int backward) {
    CDesc d;
    if (check_cdesc(g, d)) return -1;
    const CPlan pl = make_cplan(d);
    const int64_t wr = (int64_t)pl.taps * d.C_out * d.C_in;
    if (!backward) return (int64_t)pl.n_tiles * 2 * d.C_out + std::max(wr, pack_floats(pl.taps, d.C_out, d.C_in)) + 64;
    return (int64_t)pl.n_tiles * 2 * d.C_out + 2 * d.C_out + std::max(wr, pack_floats(pl.taps, d.C_in, d.C_out)) + (int64_t)pl.w_chunks * wr + 256 +
           3 * (int64_t)d.C_out + (int64_t)pl.P * d.C_out;
}

// mean 0 | 1 / std 1 | gamma 1: with these and zero sums dz8() passes the upstream gradient through (GHN3_CONV_NO_NORM)
__global__ __launch_bounds__(256) void tnet_identity_norm_kernel(float* __restrict__ p, int C) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 3 * C; i += gridDim.x * 256) p[i] = i < C ? 0.f : 1.f;
}

extern "C" int ghn3_conv_bn_fwd(const ghn3_conv_desc* g, const float* x, const float* w, const float* gamma, const float* beta, float* z,
                                float* out, float* stats, float* scratch, void* stream_) {
    CDesc d;
    int rc = check_cdesc(g, d);
    if (rc) return rc;
    const bool no_norm = (g->relu & GHN3_CONV_NO_NORM) != 0;       // z is the result: no statistics, no affine map
    if (!x || !w || !z || !scratch || (!no_norm && (!gamma || !beta || !out || !stats))) {
        ghn3_set_error("conv fwd: null pointer");
        return GHN3_E_ARG;
    }
    hipStream_t s = (hipStream_t)stream_;
    const CPlan pl = make_cplan(d);
    float* part = scratch;
    float* w_r = part + (int64_t)pl.n_tiles * 2 * d.C_out;
    if (conv2_on()) {
        unsigned short* wp = reinterpret_cast<unsigned short*>(w_r);
        const int64_t plane = (int64_t)pl.taps * d.C_out * ((d.C_in + KC - 1) / KC * KC);
        hipLaunchKernelGGL(tnet_conv_w_pack_kernel, dim3((int)std::min<int64_t>((plane + 255) / 256, 2048)), dim3(256), 0, s, w, wp, d.C_out,
                           d.C_in, pl.taps, 0);
        LAUNCH_CHECK("conv weight pack")
        const int NT = conv2_nt(pl.n_tiles, d.C_out);
        const dim3 grid(pl.n_tiles, (d.C_out + 16 * NT - 1) / (16 * NT));
        const size_t lds = conv2_lds(NT);
#define C2F_CASE(n) case n: rc = set_lds(tnet_conv2_kernel<n, false>, lds); if (rc) return rc; \
        hipLaunchKernelGGL((tnet_conv2_kernel<n, false>), grid, dim3(256), lds, s, x, wp, z, part, (const float*)nullptr, d, pl.P, pl.P_in); break;
        switch (NT) { C2F_CASE(2) C2F_CASE(4) C2F_CASE(8) }
#undef C2F_CASE
        LAUNCH_CHECK("conv fwd")
    } else {
    rc = conv_repack(d, pl, w, w_r, s);
    if (rc) return rc;
    const int NT = nt_of(d.C_out);
    const size_t lds = cfwd_lds(NT);
#define CFWD_CASE(n, t) case 10 * n + t: rc = set_lds(tnet_conv_fwd_kernel<n, t>, lds); if (rc) return rc; \
        hipLaunchKernelGGL((tnet_conv_fwd_kernel<n, t>), dim3(pl.n_tiles), dim3(256), lds, s, x, w_r, z, part, d, pl.P); break;
    switch (10 * NT + terms_of(NT)) { CFWD_CASE(4, 3) CFWD_CASE(8, 3) CFWD_CASE(16, 3) CFWD_CASE(4, 2) CFWD_CASE(8, 2) CFWD_CASE(16, 2) CFWD_CASE(32, 2) }
#undef CFWD_CASE
    LAUNCH_CHECK("conv fwd")
    }
    if (no_norm) return GHN3_OK;
    hipLaunchKernelGGL(tnet_bn_finalize_kernel, dim3((d.C_out + 15) / 16), dim3(256), 0, s, part, pl.n_tiles, pl.P, d.C_out, d.eps, stats);
    LAUNCH_CHECK("bn finalize")
    const int64_t total4 = (int64_t)pl.P * d.C_out / 4;
    hipLaunchKernelGGL(tnet_bn_apply_kernel, dim3((int)std::min<int64_t>((total4 + 255) / 256, 4096)), dim3(256), 0, s, z, stats, gamma,
                       beta, out, total4, d.C_out);
    LAUNCH_CHECK("bn apply")
    return GHN3_OK;
}

extern "C" int ghn3_conv_bn_bwd(const ghn3_conv_desc* g, const float* dout, const float* x, const float* z, const float* stats,
                                const float* w, const float* gamma, float* dx, float* dw, float* dgamma, float* dbeta, float* scratch,
                                void* stream_) {
    CDesc d;
    int rc = check_cdesc(g, d);
    if (rc) return rc;
    const bool no_norm = (g->relu & GHN3_CONV_NO_NORM) != 0;       // dout IS the gradient of the convolution's result
    if (!dout || !x || !w || !dx || !dw || !scratch || (!no_norm && (!z || !stats || !gamma || !dgamma || !dbeta))) {
        ghn3_set_error("conv bwd: null pointer");
        return GHN3_E_ARG;
    }
    hipStream_t s = (hipStream_t)stream_;
    const CPlan pl = make_cplan(d);
    const int64_t wr = (int64_t)pl.taps * d.C_out * d.C_in;
    float* part12 = scratch;
    float* const s12_scratch = part12 + (int64_t)pl.n_tiles * 2 * d.C_out;
    // (dbeta directly followed by dgamma -- how target_ops.py lays them out -- IS the [sum dout | sum dout xhat] pair: no copies)
    const bool s12_in_place = dbeta && dgamma == dbeta + d.C_out;
    float* s12 = s12_in_place ? dbeta : s12_scratch;
    float* w_r = s12_scratch + 2 * d.C_out;
    float* part_w = w_r + std::max(wr, pack_floats(pl.taps, d.C_in, d.C_out));
    float* ident = part_w + (int64_t)pl.w_chunks * wr;             // [mean 0 | 1 / std 1 | gamma 1]
    float* dzbuf = ident + 3 * d.C_out;                            // [P][C_out]: dz, written once (second-version kernels)
    const bool v2 = conv2_on();
    if (no_norm && v2) {
        // (dout is dz: nothing to prepare)
    } else if (no_norm) {
        hipLaunchKernelGGL(tnet_identity_norm_kernel, dim3(1), dim3(256), 0, s, ident, d.C_out);
        LAUNCH_CHECK("conv bwd identity")
        if (hipMemsetAsync(s12, 0, (size_t)2 * d.C_out * 4, s) != hipSuccess) { ghn3_set_error("conv bwd: memset failed"); return GHN3_E_HIP; }
        z = dout;                                                  // (read, multiplied by the zero sums)
        stats = ident;
        gamma = ident + 2 * d.C_out;
    } else
    // 1. dgamma / dbeta
    {
        const int nq = d.C_out / 4, ng = std::max(1, 256 / nq);
        hipLaunchKernelGGL(tnet_bn_bwd_partial_kernel, dim3(pl.n_tiles), dim3(256), (size_t)ng * 2 * d.C_out * 4, s, dout, z, stats, part12,
                           pl.P, d.C_out);
        LAUNCH_CHECK("bn bwd partial")
        hipLaunchKernelGGL(tnet_reduce_rows_kernel, dim3((2 * d.C_out / 4 + 15) / 16), dim3(256), 0, s, part12, pl.n_tiles,
                           (int64_t)2 * d.C_out, s12, 0, 0);
        LAUNCH_CHECK("bn bwd reduce")
        if (!s12_in_place) {
            hipMemcpyAsync(dbeta, s12, (size_t)d.C_out * 4, hipMemcpyDeviceToDevice, s);
            hipMemcpyAsync(dgamma, s12 + d.C_out, (size_t)d.C_out * 4, hipMemcpyDeviceToDevice, s);
        }
    }
    if (v2) {
        const float* dzp = dout;
        if (!no_norm) {
            const int64_t total4 = (int64_t)pl.P * d.C_out / 4;
            hipLaunchKernelGGL(tnet_dz_kernel, dim3((int)std::min<int64_t>((total4 + 255) / 256, 4096)), dim3(256), 0, s, dout, z, stats, gamma,
                               s12, dzbuf, total4, d.C_out, pl.P);
            LAUNCH_CHECK("conv bwd dz")
            dzp = dzbuf;
        }
        unsigned short* wp = reinterpret_cast<unsigned short*>(w_r);
        const int64_t plane = (int64_t)pl.taps * d.C_in * ((d.C_out + KC - 1) / KC * KC);
        hipLaunchKernelGGL(tnet_conv_w_pack_kernel, dim3((int)std::min<int64_t>((plane + 255) / 256, 2048)), dim3(256), 0, s, w, wp, d.C_out,
                           d.C_in, pl.taps, 1);
        LAUNCH_CHECK("conv weight pack (transposed)")
        const int NT = conv2_nt(pl.n_tiles_in, d.C_in);
        const dim3 grid(pl.n_tiles_in, (d.C_in + 16 * NT - 1) / (16 * NT));
        const size_t lds = conv2_lds(NT);
#define C2B_CASE(n) case n: rc = set_lds(tnet_conv2_kernel<n, true>, lds); if (rc) return rc; \
        hipLaunchKernelGGL((tnet_conv2_kernel<n, true>), grid, dim3(256), lds, s, dzp, wp, dx, (float*)nullptr, x, d, pl.P_in, pl.P); break;
        switch (NT) { C2B_CASE(2) C2B_CASE(4) C2B_CASE(8) }
#undef C2B_CASE
        LAUNCH_CHECK("conv bwd data")
        hipLaunchKernelGGL(tnet_conv_wgrad_kernel, dim3(pl.w_chunks, (d.C_out + 63) / 64, ((d.C_in + 63) / 64) * pl.taps), dim3(256), 0, s, dzp,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, x, part_w, d, pl.P,
                           pl.w_chunk_px);
        LAUNCH_CHECK("conv wgrad")
        hipLaunchKernelGGL(tnet_reduce_rows_kernel, dim3((int)((wr / 4 + 15) / 16)), dim3(256), 0, s, part_w, pl.w_chunks, wr, dw,
                           d.C_out * d.C_in, pl.taps);
        LAUNCH_CHECK("conv wgrad reduce")
        return GHN3_OK;
    }
    rc = conv_repack(d, pl, w, w_r, s);
    if (rc) return rc;
    // 2. dx
    {
        const int NT = nt_of(d.C_in);
        const size_t lds = cbwd_lds(NT);
#define CBWD_CASE(n, t) case 10 * n + t: rc = set_lds(tnet_conv_bwd_data_kernel<n, t>, lds); if (rc) return rc; \
        hipLaunchKernelGGL((tnet_conv_bwd_data_kernel<n, t>), dim3(pl.n_tiles_in), dim3(256), lds, s, dout, z, stats, gamma, s12, w_r, x, dx, d, pl.P, pl.P_in); break;
        switch (10 * NT + terms_of(NT)) { CBWD_CASE(4, 3) CBWD_CASE(8, 3) CBWD_CASE(16, 3) CBWD_CASE(4, 2) CBWD_CASE(8, 2) CBWD_CASE(16, 2) CBWD_CASE(32, 2) }
#undef CBWD_CASE
        LAUNCH_CHECK("conv bwd data")
    }
    // 3. dW (in the parameter's own [C_out][C_in][kh][kw] order)
    hipLaunchKernelGGL(tnet_conv_wgrad_kernel, dim3(pl.w_chunks, (d.C_out + 63) / 64, ((d.C_in + 63) / 64) * pl.taps), dim3(256), 0, s, dout, z,
                       stats, gamma, s12, x, part_w, d, pl.P, pl.w_chunk_px);
    LAUNCH_CHECK("conv wgrad")
    hipLaunchKernelGGL(tnet_reduce_rows_kernel, dim3((int)((wr / 4 + 15) / 16)), dim3(256), 0, s, part_w, pl.w_chunks, wr, dw,
                       d.C_out * d.C_in, pl.taps);
    LAUNCH_CHECK("conv wgrad reduce")
    return GHN3_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// squeeze-and-excitation with a hard-swish gate (round 6; ChannelSELayer, ops.py:239-274):
//   s = mean_hw(x);  h = relu(W1 s + b1);  a = W2 h + b2;  y = x * hardswish(a)
// One workgroup per sample: the reference runs mean, two hipBLASLt GEMMs of a 64-row batch (80 us of host dispatch each),
// ReLU, hard-swish and the product as ~8 launches forward and ~14 backward; here 1 and 2.  NHWC activations as the other ops.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float hswish(float a) { return a * fminf(fmaxf(a + 3.f, 0.f), 6.f) * (1.f / 6.f); }
// (torch's hardswish_backward: 0 below -3, x / 3 + 0.5 up to 3, 1 above)
__device__ __forceinline__ float hswish_grad(float a) { return a < -3.f ? 0.f : (a <= 3.f ? a * (1.f / 3.f) + 0.5f : 1.f); }

// v[c] = sum over the sample's pixels of a[p][c] (* b[p][c] when b != nullptr), c = 0 .. C - 1, into LDS `out` (floats);
// `red` = 256 float4 of LDS.  Threads = channel quad x pixel lane.
__device__ __forceinline__ void se_pixel_sum(const float* __restrict__ a, const float* __restrict__ b, int HW, int C, f32x4* red,
                                             float* out, float scale) {
    const int nq = C / 4, PL = max(1, 256 / nq), cq = threadIdx.x % nq, pl = threadIdx.x / nq;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (pl < PL) {
        for (int p = pl; p < HW; p += PL) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(a + (int64_t)p * C + 4 * cq);
            if (b) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(b + (int64_t)p * C + 4 * cq);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(u[e], v[e], acc[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += u[e];
            }
        }
        red[pl * nq + cq] = acc;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < nq; q += 256) {
        f32x4 sum = red[q];
        for (int l = 1; l < PL; ++l) {
            const f32x4 v = red[l * nq + q];
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) out[4 * q + e] = sum[e] * scale;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void tnet_se_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                          const float* __restrict__ b1, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, float* __restrict__ y, float* __restrict__ save,
                                                          const int HW, const int C, const int J) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);                               // [256]
    float* s = reinterpret_cast<float*>(smem + 4096);                          // [C]
    float* g = s + C;                                                          // [C]
    float* h = g + C;                                                          // [J]
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* xn = x + (int64_t)n * HW * C;
    float* sv = save + (int64_t)n * (2 * C + J);
    se_pixel_sum(xn, nullptr, HW, C, red, s, 1.f / (float)HW);
    for (int j = wv; j < J; j += 4) {                                          // h = relu(W1 s + b1): a wave per output
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc = fmaf(w1[(int64_t)j * C + c], s[c], acc);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
        if (lane == 0) h[j] = fmaxf(acc + b1[j], 0.f);
    }
    __syncthreads();
    for (int c = wv; c < C; c += 4) {                                          // a = W2 h + b2, g = hardswish(a)
        float acc = 0.f;
        for (int j = lane; j < J; j += 64) acc = fmaf(w2[(int64_t)c * J + j], h[j], acc);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
        if (lane == 0) {
            const float a = acc + b2[c];
            g[c] = hswish(a);
            sv[C + c] = a;
        }
    }
    for (int c = tid; c < C; c += 256) sv[c] = s[c];
    for (int j = tid; j < J; j += 256) sv[2 * C + j] = h[j];
    __syncthreads();
    const int nq = C / 4;
    float* yn = y + (int64_t)n * HW * C;
    for (int i = tid; i < HW * nq; i += 256) {
        const int c4 = (i % nq) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(xn + 4 * (int64_t)i);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= g[c4 + e];
        *reinterpret_cast<f32x4*>(yn + 4 * (int64_t)i) = v;
    }
}

// per sample: dx, and the vectors the parameter gradients are sums of: vec[n] = [da (C) | dz1 (J)]
__global__ __launch_bounds__(256) void tnet_se_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ w1, const float* __restrict__ w2,
                                                          const float* __restrict__ save, float* __restrict__ dx, float* __restrict__ vec,
                                                          const int HW, const int C, const int J) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    float* da = reinterpret_cast<float*>(smem + 4096);                         // [C]: dg, then da in place
    float* ds = da + C;                                                        // [C]
    float* dz = ds + C;                                                        // [J]
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* xn = x + (int64_t)n * HW * C;
    const float* dyn = dy + (int64_t)n * HW * C;
    const float* sv = save + (int64_t)n * (2 * C + J);
    float* vn = vec + (int64_t)n * (C + J);
    se_pixel_sum(dyn, xn, HW, C, red, da, 1.f);                                // dg[c] = sum_p dy x
    for (int c = tid; c < C; c += 256) {
        const float v = da[c] * hswish_grad(sv[C + c]);
        da[c] = v;
        vn[c] = v;
    }
    __syncthreads();
    for (int j = tid; j < J; j += 256) {                                       // dh = W2^T da, masked by the ReLU
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc = fmaf(w2[(int64_t)c * J + j], da[c], acc);
        const float v = sv[2 * C + j] > 0.f ? acc : 0.f;
        dz[j] = v;
        vn[C + j] = v;
    }
    __syncthreads();
    const float inv = 1.f / (float)HW;
    for (int c = tid; c < C; c += 256) {                                       // ds = W1^T dz1, spread over the pixels
        float acc = 0.f;
        for (int j = 0; j < J; ++j) acc = fmaf(w1[(int64_t)j * C + c], dz[j], acc);
        ds[c] = acc * inv;
    }
    __syncthreads();
    const int nq = C / 4;
    float* dxn = dx + (int64_t)n * HW * C;
    for (int i = tid; i < HW * nq; i += 256) {
        const int c4 = (i % nq) * 4;
        const f32x4 gy = *reinterpret_cast<const f32x4*>(dyn + 4 * (int64_t)i);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(gy[e], hswish(sv[C + c4 + e]), ds[c4 + e]);
        *reinterpret_cast<f32x4*>(dxn + 4 * (int64_t)i) = v;
    }
}

// dW2[c][j] = sum_n da[n][c] h[n][j];  dW1[j][c] = sum_n dz1[n][j] s[n][c];  db2 = sum_n da;  db1 = sum_n dz1  (fixed order over n)
__global__ __launch_bounds__(256) void tnet_se_wgrad_kernel(const float* __restrict__ vec, const float* __restrict__ save,
                                                            float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                            float* __restrict__ db2, const int N, const int C, const int J) {
    const int64_t cj = (int64_t)C * J, total = 2 * cj + C + J;
    const int sv_ld = 2 * C + J, v_ld = C + J;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float acc = 0.f;
        if (i < cj) {                                                          // dW2[c][j]
            const int c = (int)(i / J), j = (int)(i % J);
            for (int n = 0; n < N; ++n) acc = fmaf(vec[(int64_t)n * v_ld + c], save[(int64_t)n * sv_ld + 2 * C + j], acc);
            dw2[i] = acc;
        } else if (i < 2 * cj) {                                               // dW1[j][c]
            const int64_t r = i - cj;
            const int j = (int)(r / C), c = (int)(r % C);
            for (int n = 0; n < N; ++n) acc = fmaf(vec[(int64_t)n * v_ld + C + j], save[(int64_t)n * sv_ld + c], acc);
            dw1[r] = acc;
        } else if (i < 2 * cj + C) {
            const int c = (int)(i - 2 * cj);
            for (int n = 0; n < N; ++n) acc += vec[(int64_t)n * v_ld + c];
            db2[c] = acc;
        } else {
            const int j = (int)(i - 2 * cj - C);
            for (int n = 0; n < N; ++n) acc += vec[(int64_t)n * v_ld + C + j];
            db1[j] = acc;
        }
    }
}

int check_se(int N, int HW, int C, int J) {
    if (N <= 0 || HW <= 0 || C <= 0 || J <= 0) { ghn3_set_error("se: non-positive size"); return GHN3_E_ARG; }
    if ((C & 3) || C > 1024 || J > 1024 || (int64_t)N * HW * C >= ((int64_t)1 << 31)) {
        ghn3_set_error("se: needs C a multiple of 4, C, J <= 1024 and fewer than 2^31 activations (got C = %d, J = %d)", C, J);
        return GHN3_E_LIMIT;
    }
    return GHN3_OK;
}

}  // namespace

extern "C" int ghn3_se_fwd(int N, int HW, int C, int J, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                           float* y, float* save, void* stream_) {
    int rc = check_se(N, HW, C, J);
    if (rc) return rc;
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || !save) { ghn3_set_error("se fwd: null pointer"); return GHN3_E_ARG; }
    hipLaunchKernelGGL(tnet_se_fwd_kernel, dim3(N), dim3(256), (size_t)4096 + (2 * C + J) * 4, (hipStream_t)stream_, x, w1, b1, w2, b2, y, save,
                       HW, C, J);
    LAUNCH_CHECK("se fwd")
    return GHN3_OK;
}

extern "C" int ghn3_se_bwd(int N, int HW, int C, int J, const float* dy, const float* x, const float* w1, const float* w2, const float* save,
                           float* dx, float* dw1, float* db1, float* dw2, float* db2, float* scratch, void* stream_) {
    int rc = check_se(N, HW, C, J);
    if (rc) return rc;
    if (!dy || !x || !w1 || !w2 || !save || !dx || !dw1 || !db1 || !dw2 || !db2 || !scratch) {
        ghn3_set_error("se bwd: null pointer");
        return GHN3_E_ARG;
    }
    hipStream_t s = (hipStream_t)stream_;
    hipLaunchKernelGGL(tnet_se_bwd_kernel, dim3(N), dim3(256), (size_t)4096 + (2 * C + J) * 4, s, dy, x, w1, w2, save, dx, scratch, HW, C, J);
    LAUNCH_CHECK("se bwd")
    const int64_t total = (int64_t)2 * C * J + C + J;
    hipLaunchKernelGGL(tnet_se_wgrad_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, scratch, save, dw1, db1, dw2,
                       db2, N, C, J);
    LAUNCH_CHECK("se wgrad")
    return GHN3_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// k x k max / average pooling on NHWC activations (round 6; the `max_pool_3x3` / `avg_pool_3x3` ops of ops.py:289-291 and the
// stems' MaxPool2d(3, 2, 1)): ATen's pooling kernels return wrong input gradients for channels_last tensors on this ROCm build, so
// every pooling layer behind a fused layer paid two layout copies.  Average pooling counts valid taps only
// (count_include_pad = False, as the search space builds it); max pooling keeps the FIRST maximum in (kh, kw) scan order as
// torch does and stores its tap in one byte per output element.  Backward as a gather per input pixel: no atomics.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct PDesc { int N, H, W, C, k, stride, pad, Ho, Wo, mode; };

__global__ __launch_bounds__(256) void tnet_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            unsigned char* __restrict__ idx, const PDesc d, const int64_t total) {
    const int nq = d.C / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cq = (int)(i % nq);
        const int64_t p = i / nq;
        const int ow = (int)(p % d.Wo), oh = (int)((p / d.Wo) % d.Ho), n = (int)(p / ((int64_t)d.Wo * d.Ho));
        const int ih0 = oh * d.stride - d.pad, iw0 = ow * d.stride - d.pad;
        f32x4 acc = d.mode ? f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY} : f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned arg[4] = {0, 0, 0, 0};
        int cnt = 0;
        for (int kh = 0; kh < d.k; ++kh) {
            const int ih = ih0 + kh;
            if (ih < 0 || ih >= d.H) continue;
            for (int kw = 0; kw < d.k; ++kw) {
                const int iw = iw0 + kw;
                if (iw < 0 || iw >= d.W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((int64_t)n * d.H + ih) * d.W + iw) * d.C + 4 * cq);
                ++cnt;
                if (d.mode) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (v[e] > acc[e] || v[e] != v[e]) { acc[e] = v[e]; arg[e] = (unsigned)(kh * d.k + kw); }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += v[e];
                }
            }
        }
        if (!d.mode) {
            const float inv = 1.f / (float)max(cnt, 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] *= inv;
        } else {
            *reinterpret_cast<unsigned*>(idx + 4 * i) = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
        }
        *reinterpret_cast<f32x4*>(y + 4 * i) = acc;
    }
}

__device__ __forceinline__ int pool_count(const PDesc& d, int o, int size) {     // valid taps of output index o along one axis
    const int lo = max(o * d.stride - d.pad, 0), hi = min(o * d.stride - d.pad + d.k, size);
    return max(hi - lo, 0);
}

__global__ __launch_bounds__(256) void tnet_pool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                            float* __restrict__ dx, const PDesc d, const int64_t total) {
    const int nq = d.C / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cq = (int)(i % nq);
        const int64_t p = i / nq;
        const int iw = (int)(p % d.W), ih = (int)((p / d.W) % d.H), n = (int)(p / ((int64_t)d.W * d.H));
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // outputs whose window holds (ih, iw): oh * stride - pad <= ih < oh * stride - pad + k
        const int oh_lo = max(0, (ih + d.pad - d.k + d.stride) / d.stride), oh_hi = min(d.Ho - 1, (ih + d.pad) / d.stride);
        const int ow_lo = max(0, (iw + d.pad - d.k + d.stride) / d.stride), ow_hi = min(d.Wo - 1, (iw + d.pad) / d.stride);
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            const int kh = ih - (oh * d.stride - d.pad);
            if (kh < 0 || kh >= d.k) continue;
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const int kw = iw - (ow * d.stride - d.pad);
                if (kw < 0 || kw >= d.k) continue;
                const int64_t o = (((int64_t)n * d.Ho + oh) * d.Wo + ow) * nq + cq;
                const f32x4 g = *reinterpret_cast<const f32x4*>(dy + 4 * o);
                if (d.mode) {
                    const unsigned a = *reinterpret_cast<const unsigned*>(idx + 4 * o), t = (unsigned)(kh * d.k + kw);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (((a >> (8 * e)) & 255u) == t) acc[e] += g[e];
                } else {
                    const float inv = 1.f / (float)max(pool_count(d, oh, d.H) * pool_count(d, ow, d.W), 1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(g[e], inv, acc[e]);
                }
            }
        }
        *reinterpret_cast<f32x4*>(dx + 4 * i) = acc;
    }
}

int check_pdesc(const ghn3_pool_desc* g, PDesc& d) {
    if (!g) { ghn3_set_error("pool: null descriptor"); return GHN3_E_ARG; }
    d = PDesc{g->N, g->H, g->W, g->C, g->k, g->stride, g->pad, g->Ho, g->Wo, g->mode};
    if (d.N <= 0 || d.H <= 0 || d.W <= 0 || d.C <= 0 || d.k <= 0 || d.stride <= 0 || d.pad < 0 || (d.mode != 0 && d.mode != 1)) {
        ghn3_set_error("pool: bad descriptor");
        return GHN3_E_ARG;
    }
    if ((d.C & 3) || d.k > 15 || 2 * d.pad > d.k) {
        ghn3_set_error("pool: needs C a multiple of 4, k <= 15, pad <= k / 2 (got C = %d, k = %d, pad = %d)", d.C, d.k, d.pad);
        return GHN3_E_LIMIT;
    }
    const int ho = (d.H + 2 * d.pad - d.k) / d.stride + 1, wo = (d.W + 2 * d.pad - d.k) / d.stride + 1;
    if (ho != d.Ho || wo != d.Wo || ho <= 0 || wo <= 0) {
        ghn3_set_error("pool: output size %d x %d does not match the pooling arithmetic (%d x %d, floor mode)", d.Ho, d.Wo, ho, wo);
        return GHN3_E_ARG;
    }
    if ((int64_t)d.N * d.H * d.W * d.C >= ((int64_t)1 << 31)) { ghn3_set_error("pool: 2^31 activations or more"); return GHN3_E_LIMIT; }
    return GHN3_OK;
}

}  // namespace

extern "C" int ghn3_pool_fwd(const ghn3_pool_desc* g, const float* x, float* y, unsigned char* idx, void* stream_) {
    PDesc d;
    int rc = check_pdesc(g, d);
    if (rc) return rc;
    if (!x || !y || (d.mode == 1 && !idx)) { ghn3_set_error("pool fwd: null pointer"); return GHN3_E_ARG; }
    const int64_t total = (int64_t)d.N * d.Ho * d.Wo * (d.C / 4);
    hipLaunchKernelGGL(tnet_pool_fwd_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream_, x, y, idx,
                       d, total);
    LAUNCH_CHECK("pool fwd")
    return GHN3_OK;
}

extern "C" int ghn3_pool_bwd(const ghn3_pool_desc* g, const float* dy, const unsigned char* idx, float* dx, void* stream_) {
    PDesc d;
    int rc = check_pdesc(g, d);
    if (rc) return rc;
    if (!dy || !dx || (d.mode == 1 && !idx)) { ghn3_set_error("pool bwd: null pointer"); return GHN3_E_ARG; }
    const int64_t total = (int64_t)d.N * d.H * d.W * (d.C / 4);
    hipLaunchKernelGGL(tnet_pool_bwd_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream_, dy, idx,
                       dx, d, total);
    LAUNCH_CHECK("pool bwd")
    return GHN3_OK;
}
