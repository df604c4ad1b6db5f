// Split-bf16 GEMMs of the Graphormer chain without partial planes and without separate LayerNorm launches (gfx950).
//
// The chain of a 256-node graph is ~170 (forward) / ~220 (backward) dependent launches of a few microseconds each; what
// a launch costs is mostly its place in the chain (~3 us of dispatch + the round trip of its first loads), not its flops.
// Round 3 ran seven launches per layer forward (LayerNorm, to_qkv, attention, to_out, LayerNorm, ff.net.0, ff.net.3) and
// cut the narrow-output linears (N = C, K up to 4C) along K over several workgroup sets whose partial planes the next
// LayerNorm launch summed.  gemm_x3s_kernel ("staged", tile codes 44 / 45) removes both:
//
//   * the WEIGHT fragments come straight from global memory into registers: the persistent bf16 hi / lo copies are kept
//     in FRAGMENT-MAJOR order (GHN3_CAST_FRAG: the 1 KB operand of one v_mfma_f32_16x16x32_bf16 -- 16 output columns x 32 k --
//     is contiguous, lane L owns bytes [16 L, 16 L + 16)), so every load instruction is one fully coalesced kilobyte and
//     needs no LDS, no barrier and no conversion.  (Fragment loads from a row-major copy touch 16 cache lines per 16-lane
//     pass: measured 13.5 us for the ff.net.3 shape against 7.2 us for the plane kernel -- r04a.)  All weight loads of a
//     wave are issued before anything else.
//   * the ACTIVATION rows of the workgroup (fp32, 16 or 32 rows x the WHOLE reduction length) are loaded once with
//     coalesced 32-byte pieces, optionally pass through a LayerNorm forward (graphormer.py:239,241) or a LayerNorm
//     backward (+ residual gradient) computed on the whole rows (statistics by lane shuffles), are split into bf16 hi / lo
//     and left in LDS as swizzled images; the column-tile-0 workgroups write the by-products the backward needs
//     (normalised rows, mean, rstd / the propagated gradient).
//   * the waves of a workgroup form MT x NT output tiles x KP parts of K; every wave multiplies its K part from LDS
//     (activations) and registers (weights); the KP partial tiles are added through LDS in a fixed order: no partial
//     planes in HBM, hence no plane sums in the consumer and bit-identical reruns.
//
// Arithmetic = gemm_x3.hip: a = a_hi + a_lo (bf16 each), a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi on
// v_mfma_f32_16x16x32_bf16, fp32 accumulation, the cross terms in their own accumulator (8e-6 relative against fp64).
// Products are taken transposed (the weight fragment is the MFMA A operand): a lane owns 4 consecutive output columns.

#include "ghn3_internal.h"

#define GAS __attribute__((address_space(1)))
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// tools/x3s_probe.hip builds this file with GHN3_X3S_PROBE: thread 0 of workgroup 0 leaves shader-clock stamps
#ifdef GHN3_X3S_PROBE
__device__ long long g_x3s_stamps[16];
#define X3_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_x3s_stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define X3_STAMP(i) do { } while (0)
#endif

namespace {

__device__ __forceinline__ float d_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float d_gelu_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& h, bf16x8& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 ah = (__bf16)a[e], bh = (__bf16)b[e];
        h[e] = ah; h[4 + e] = bh;
        l[e] = (__bf16)(a[e] - (float)ah);
        l[4 + e] = (__bf16)(b[e] - (float)bh);
    }
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// the same split into F16 pieces (GHN3_GEMM_X3F16: 11 + 11 bits of mantissa, O(1) operands only), as raw 16-bit patterns
__device__ __forceinline__ void split8_f16(const f32x4& a, const f32x4& b, bf16x8& h, bf16x8& l) {
    // |x| > 65504 would become inf and the lo piece x - inf a NaN where the bf16-piece and fp32 paths (and the reference) stay
    // finite: both pieces saturate at the largest f16 (v_med3_f32); a NaN input stays NaN through the hi piece.
    f16x8 hh, ll;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ae = a[e] != a[e] ? a[e] : __builtin_amdgcn_fmed3f(a[e], -65504.f, 65504.f);
        const float be = b[e] != b[e] ? b[e] : __builtin_amdgcn_fmed3f(b[e], -65504.f, 65504.f);
        const _Float16 ah = (_Float16)ae, bh = (_Float16)be;
        hh[e] = ah; hh[4 + e] = bh;
        ll[e] = (_Float16)__builtin_amdgcn_fmed3f(a[e] - (float)ah, -65504.f, 65504.f);
        ll[4 + e] = (_Float16)__builtin_amdgcn_fmed3f(b[e] - (float)bh, -65504.f, 65504.f);
    }
    h = __builtin_bit_cast(bf16x8, hh);
    l = __builtin_bit_cast(bf16x8, ll);
}

__device__ __forceinline__ const GemmProbDev* find_problem(const GemmProbDev* probs, int n_probs) {
    int lo = 0, hi = n_probs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (probs[mid].tile_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    return probs + lo;
}

// epilogue of one 16 x 16 tile: the lane owns C[m][n .. n + 3]
__device__ __forceinline__ void x3_epilogue(const GemmProbDev* P, f32x4 v, int m, int n) {
    if (m >= P->M || n >= P->N) return;
    const int64_t ci = (int64_t)(P->c_gather ? P->c_gather[m] : m) * P->ldc + n;
    const float alpha = P->alpha;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= alpha;
    if (P->bias) v += *reinterpret_cast<const f32x4 GAS*>((const float GAS*)P->bias + n);
    if (P->aux_out) *reinterpret_cast<f32x4 GAS*>((float GAS*)P->aux_out + ci) = v;
    if (P->act == GHN3_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (P->act == GHN3_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = d_gelu(v[e]);
    }
    if (P->dact != GHN3_DACT_NONE) {
        const f32x4 a = *reinterpret_cast<const f32x4 GAS*>((const float GAS*)P->aux_in + ci);
        if (P->dact == GHN3_DACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = a[e] > 0.f ? v[e] : 0.f;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= d_gelu_grad(a[e]);
        }
    }
    if (P->residual) v += *reinterpret_cast<const f32x4 GAS*>((const float GAS*)P->residual + ci);
    *reinterpret_cast<f32x4 GAS*>((float GAS*)P->C + ci) = v;
}

// ---------------------------------------------------------------------------------------------------------------------
// staged kernel: K = 32 KS KP (a multiple of 64), BM = 16 MT rows and 16 NT columns per workgroup, MT NT KP waves.
//   PRO 0: A' = A (optional row gather a_gather)
//   PRO 1 (ghn3_gemm_problem::ln_kind 1): A' = (A - mean) rstd p0 + p1; p2 = mean out, p3 = rstd out, p4 = A' out
//   PRO 2 (ln_kind 2): A = dy, A' = rstd (dy g - s1 - xhat s2) + res, s1 = mean_k(dy g), s2 = mean_k(dy g xhat),
//         xhat = (x - mean) rstd; p0 = g, p1 = x, p2 = mean, p3 = rstd, p4 = res or absent, p5 = A' out or absent
// B / B2: fragment-major bf16 hi / lo copies of the weight W [N][K] (GHN3_CAST_FRAG): fragment (n / 16, k / 32) holds
//         512 elements, element (n, k) at ((k % 32) / 8 * 16 + n % 16) * 8 + k % 8.
// LDS: bf16 hi / lo images of A' [k-tile of 64][row][128 B], the 16-byte slot s of row r holds chunk s ^ ((r >> 1) & 7)
// (conflict-free ds_read_b128 for the 16-lane groups of the fragment reads), then the KP - 1 partial tiles.
// ---------------------------------------------------------------------------------------------------------------------
template <int MT, int NT, int KP, int KS, int PRO>
__device__ __forceinline__ void x3s_body(const GemmProbDev* __restrict__ P) {
    extern __shared__ __attribute__((aligned(16))) char x3s_smem[];
    constexpr int NTHR = 64 * MT * NT * KP, BM = 16 * MT, K = 32 * KS * KP, NKT = K / 64;
    constexpr int CHUNKS = K / 8;                         // 8-float chunks per row
    static_assert(K % 64 == 0, "whole 64-wide k-tiles");
    char* sAh = x3s_smem;
    char* sAl = sAh + NKT * BM * 128;
    f32x4* red = reinterpret_cast<f32x4*>(sAl + NKT * BM * 128);

    X3_STAMP(0);
    const int t_id = blockIdx.x - P->tile_start;
    const int n0 = (t_id % P->tiles_n) * (16 * NT), m0 = (t_id / P->tiles_n) * BM;
    const int M = P->M, N = P->N, lda = P->lda;
    const bool f16p = (P->flags & GHN3_GEMM_X3F16) != 0;          // f16 pieces (forward linears): wave-uniform
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int nt = wave % NT, mt = (wave / NT) % MT, kp = wave / (NT * MT);

    // ---- 0. requests, in the order the data is needed.  A wave's loads return in order (vmcnt): the rows of the workgroup --
    // small, L2-warm, consumed first by the prologue -- are requested FIRST, the weight fragments (cold, straight from HBM,
    // consumed last by the products) behind them, so that the prologue waits for the rows only.  (Round 4 first asked for the
    // weights first: tools/x3s_probe showed every wave's prologue waiting for its weights, 5-6k of a workgroup's 13-15k cycles
    // plus 2.5-3k at the barrier behind the slowest wave.)  All row loads are unconditional (threads outside the prologue
    // re-read the last row): a load inside an if-block is waited for inside it.
    constexpr int TPR0 = NTHR / BM;
    constexpr int TPR = (NTHR % BM == 0 && TPR0 <= 64 && (TPR0 & (TPR0 - 1)) == 0) ? TPR0 : 16;
    constexpr int NCH = (CHUNKS + TPR - 1) / TPR;                      // PRO >= 1: chunks per thread of a row
    constexpr int SLOTS = BM * CHUNKS, NSL = (SLOTS + NTHR - 1) / NTHR;   // PRO 0: chunk-linear mapping
    constexpr int NA = PRO == 0 ? NSL : NCH;
    // PRO >= 1: thread (pr, pc) owns chunks pc + TPR j of row pr: the TPR threads of a row are consecutive lanes of one wave (a
    // power of two <= 64; workgroups whose thread count is not a power-of-two multiple of BM -- 12 waves on 32 rows -- run the
    // prologue on their first 16 BM threads)
    const bool pro_on = tid < TPR * BM;
    const int pr = min(tid / TPR, BM - 1), pc = tid % TPR;
    const int prow = m0 + pr;
    const bool row_ok = prow < M;
    const int64_t ro = (int64_t)min(prow, M - 1) * lda;
    f32x4 a[NA][2];
    f32x4 q0[PRO == 0 ? 1 : NCH][2], q1[PRO == 0 ? 1 : NCH][2], q2[PRO == 2 ? NCH : 1][2];   // PRO 1: gamma, beta; PRO 2: x, res, gamma
    float mu_in = 0.f, rs_in = 0.f;
    const bool has_res = PRO == 2 && P->ln_p[4] != nullptr;
    if constexpr (PRO == 0) {
        // chunk-linear mapping: thread t stages chunks t, t + NTHR, ... of the BM x CHUNKS chunk grid (row major)
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            const int ci = min(tid + NTHR * j, SLOTS - 1);
            const int r_ = ci / CHUNKS, c = ci % CHUNKS;
            int row = min(m0 + r_, M - 1);
            if (P->a_gather) row = P->a_gather[row];
            const float GAS* src = (const float GAS*)P->A + (int64_t)row * lda + 8 * c;
            a[j][0] = *reinterpret_cast<const f32x4 GAS*>(src);
            a[j][1] = *reinterpret_cast<const f32x4 GAS*>(src + 4);
        }
    } else if constexpr (PRO == 1) {
        const float GAS* x = (const float GAS*)P->A + ro;
        const float GAS* g = (const float GAS*)P->ln_p[0];
        const float GAS* bt = (const float GAS*)P->ln_p[1];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = min(pc + TPR * j, CHUNKS - 1);
            a[j][0] = *reinterpret_cast<const f32x4 GAS*>(x + 8 * c);
            a[j][1] = *reinterpret_cast<const f32x4 GAS*>(x + 8 * c + 4);
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = min(pc + TPR * j, CHUNKS - 1);
            q0[j][0] = *reinterpret_cast<const f32x4 GAS*>(g + 8 * c);
            q0[j][1] = *reinterpret_cast<const f32x4 GAS*>(g + 8 * c + 4);
            q1[j][0] = *reinterpret_cast<const f32x4 GAS*>(bt + 8 * c);
            q1[j][1] = *reinterpret_cast<const f32x4 GAS*>(bt + 8 * c + 4);
        }
    } else {
        const float GAS* dy = (const float GAS*)P->A + ro;
        const float GAS* x = (const float GAS*)P->ln_p[1] + ro;
        const float GAS* g = (const float GAS*)P->ln_p[0];
        const float GAS* res = has_res ? (const float GAS*)P->ln_p[4] + ro : dy;      // (absent: re-reads dy, not used)
        mu_in = P->ln_p[2][min(prow, M - 1)];
        rs_in = P->ln_p[3][min(prow, M - 1)];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int co = 8 * min(pc + TPR * j, CHUNKS - 1);
            a[j][0] = *reinterpret_cast<const f32x4 GAS*>(dy + co);
            a[j][1] = *reinterpret_cast<const f32x4 GAS*>(dy + co + 4);
            q0[j][0] = *reinterpret_cast<const f32x4 GAS*>(x + co);
            q0[j][1] = *reinterpret_cast<const f32x4 GAS*>(x + co + 4);
            q1[j][0] = *reinterpret_cast<const f32x4 GAS*>(res + co);
            q1[j][1] = *reinterpret_cast<const f32x4 GAS*>(res + co + 4);
            q2[j][0] = *reinterpret_cast<const f32x4 GAS*>(g + co);
            q2[j][1] = *reinterpret_cast<const f32x4 GAS*>(g + co + 4);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // weight fragments of this wave (one coalesced kilobyte per instruction)
    const int ntile = min(n0 + 16 * nt, N - 16) >> 4;     // (N % 16 == 0; a column tile beyond N repeats the last one)
    const int64_t fo = ((int64_t)ntile * (K / 32) + kp * KS) * 512 + lane * 8;
    const unsigned short GAS* bhp = (const unsigned short GAS*)P->B + fo;
    const unsigned short GAS* blp = (const unsigned short GAS*)P->B2 + fo;
    u16x8 wh[KS], wl[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        wh[s] = *reinterpret_cast<const u16x8 GAS*>(bhp + 512 * s);
        wl[s] = *reinterpret_cast<const u16x8 GAS*>(blp + 512 * s);
    }
    __builtin_amdgcn_sched_barrier(0);
    X3_STAMP(1);

    // ---- 1. the rows of the workgroup -> bf16 hi / lo images in LDS
    if constexpr (PRO == 0) {
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            const int ci = tid + NTHR * j;
            if (ci < SLOTS) {
                const int r_ = ci / CHUNKS, c = ci % CHUNKS;
                bf16x8 h, l;
                if (f16p) split8_f16(a[j][0], a[j][1], h, l); else split8(a[j][0], a[j][1], h, l);
                const int off = (c >> 3) * BM * 128 + r_ * 128 + (((c & 7) ^ ((r_ >> 1) & 7)) << 4);
                *reinterpret_cast<bf16x8*>(sAh + off) = h;
                *reinterpret_cast<bf16x8*>(sAl + off) = l;
            }
        }
    } else if (pro_on) {
        const int row = prow;
        const bool side = n0 == 0 && row_ok;              // column-tile-0 workgroups write the by-products
        const float inv_k = 1.0f / (float)K;
        if constexpr (PRO == 1) {
            float s_ = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j)
                if (pc + TPR * j < CHUNKS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) s_ += a[j][0][e] + a[j][1][e];
                }
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) s_ += __shfl_xor(s_, o, 64);
            const float mu = s_ * inv_k;
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j)
                if (pc + TPR * j < CHUNKS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d0 = a[j][0][e] - mu, d1 = a[j][1][e] - mu;
                        v += d0 * d0 + d1 * d1;
                    }
                }
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) v += __shfl_xor(v, o, 64);
            const float rs = rsqrtf(v * inv_k + P->ln_eps);
            if (side && pc == 0) {
                if (P->ln_p[2]) const_cast<float*>(P->ln_p[2])[row] = mu;
                if (P->ln_p[3]) const_cast<float*>(P->ln_p[3])[row] = rs;
            }
            float GAS* yout = (side && P->ln_p[4]) ? (float GAS*)const_cast<float*>(P->ln_p[4]) + ro : nullptr;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int c = pc + TPR * j;
                if (c < CHUNKS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[j][0][e] = (a[j][0][e] - mu) * rs * q0[j][0][e] + q1[j][0][e];
                        a[j][1][e] = (a[j][1][e] - mu) * rs * q0[j][1][e] + q1[j][1][e];
                    }
                    if (yout) {
                        *reinterpret_cast<f32x4 GAS*>(yout + 8 * c) = a[j][0];
                        *reinterpret_cast<f32x4 GAS*>(yout + 8 * c + 4) = a[j][1];
                    }
                }
            }
        } else {
            const float mu = mu_in, rs = rs_in;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const bool in = pc + TPR * j < CHUNKS;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[j][0][e] = in ? a[j][0][e] * q2[j][0][e] : 0.f;
                    a[j][1][e] = in ? a[j][1][e] * q2[j][1][e] : 0.f;
                    q0[j][0][e] = in ? (q0[j][0][e] - mu) * rs : 0.f;
                    q0[j][1][e] = in ? (q0[j][1][e] - mu) * rs : 0.f;
                    s1 += a[j][0][e] + a[j][1][e];
                    s2 += a[j][0][e] * q0[j][0][e] + a[j][1][e] * q0[j][1][e];
                }
            }
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
            s1 *= inv_k; s2 *= inv_k;
            float GAS* yout = (side && P->ln_p[5]) ? (float GAS*)const_cast<float*>(P->ln_p[5]) + ro : nullptr;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int c = pc + TPR * j;
                if (c < CHUNKS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[j][0][e] = rs * (a[j][0][e] - s1 - q0[j][0][e] * s2) + (has_res ? q1[j][0][e] : 0.f);
                        a[j][1][e] = rs * (a[j][1][e] - s1 - q0[j][1][e] * s2) + (has_res ? q1[j][1][e] : 0.f);
                    }
                    if (yout) {
                        *reinterpret_cast<f32x4 GAS*>(yout + 8 * c) = a[j][0];
                        *reinterpret_cast<f32x4 GAS*>(yout + 8 * c + 4) = a[j][1];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = pc + TPR * j;
            if (c < CHUNKS) {
                bf16x8 h, l;
                if (f16p) split8_f16(a[j][0], a[j][1], h, l); else split8(a[j][0], a[j][1], h, l);
                const int off = (c >> 3) * BM * 128 + pr * 128 + (((c & 7) ^ ((pr >> 1) & 7)) << 4);
                *reinterpret_cast<bf16x8*>(sAh + off) = h;
                *reinterpret_cast<bf16x8*>(sAl + off) = l;
            }
        }
    }
    X3_STAMP(2);
    __syncthreads();
    X3_STAMP(3);

    // ---- 2. products
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int arow = 16 * mt + l15;
    if (f16p) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int kstep = kp * KS + s;
            const int ch = 4 * (kstep & 1) + lq;
            const int off = (kstep >> 1) * BM * 128 + arow * 128 + ((ch ^ ((arow >> 1) & 7)) << 4);
            const f16x8 xh = *reinterpret_cast<const f16x8*>(sAh + off);
            const f16x8 xl = *reinterpret_cast<const f16x8*>(sAl + off);
            const f16x8 h = __builtin_bit_cast(f16x8, wh[s]), l = __builtin_bit_cast(f16x8, wl[s]);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, xh, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, xl, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(l, xh, acc1, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int kstep = kp * KS + s;
            const int ch = 4 * (kstep & 1) + lq;
            const int off = (kstep >> 1) * BM * 128 + arow * 128 + ((ch ^ ((arow >> 1) & 7)) << 4);
            const bf16x8 xh = *reinterpret_cast<const bf16x8*>(sAh + off);
            const bf16x8 xl = *reinterpret_cast<const bf16x8*>(sAl + off);
            const bf16x8 h = __builtin_bit_cast(bf16x8, wh[s]), l = __builtin_bit_cast(bf16x8, wl[s]);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h, xh, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h, xl, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l, xh, acc1, 0, 0, 0);
        }
    }
    f32x4 v = acc0 + acc1;
    X3_STAMP(4);
    if (KP > 1) {
        if (kp > 0) red[((kp - 1) * MT * NT + mt * NT + nt) * 64 + lane] = v;
        __syncthreads();
        X3_STAMP(5);
        if (kp > 0) return;
#pragma unroll
        for (int p = 1; p < KP; ++p) v += red[((p - 1) * MT * NT + mt * NT + nt) * 64 + lane];
    }
    x3_epilogue(P, v, m0 + 16 * mt + l15, n0 + 16 * nt + 4 * lq);
    X3_STAMP(6);
}
template <int MT, int NT, int KP, int KS, int PRO>
__global__ __launch_bounds__(64 * MT * NT * KP) void gemm_x3s_kernel(const GemmProbDev* __restrict__ probs, int n_probs) {
    x3s_body<MT, NT, KP, KS, PRO>(find_problem(probs, n_probs));
}
// one problem per launch (every GEMM of the dependent chain): the problem travels BY VALUE in the kernel arguments -- its
// fields are scalar loads from the kernarg segment instead of a dependent round trip through the device-side problem table
// in front of the first request (tools/x3s_probe: 2.2-4.2k cycles passed before a workgroup had issued its loads)
template <int MT, int NT, int KP, int KS, int PRO>
__global__ __launch_bounds__(64 * MT * NT * KP) void gemm_x3s_kernel_v(const GemmProbDev Pv) {
    x3s_body<MT, NT, KP, KS, PRO>(&Pv);
}

typedef void (*kfn)(const GemmProbDev*, int);
typedef void (*kfnv)(const GemmProbDev);
struct Cfg { int code, K, mt, nt, kp, ks; kfn fn[3]; kfnv vfn[3]; };
#define X3S(CODE, MT, NT, KP, KS) {CODE, 32 * KS * KP, MT, NT, KP, KS, {gemm_x3s_kernel<MT, NT, KP, KS, 0>, \
                                   gemm_x3s_kernel<MT, NT, KP, KS, 1>, gemm_x3s_kernel<MT, NT, KP, KS, 2>}, \
                                   {gemm_x3s_kernel_v<MT, NT, KP, KS, 0>, gemm_x3s_kernel_v<MT, NT, KP, KS, 1>, \
                                    gemm_x3s_kernel_v<MT, NT, KP, KS, 2>}}
#define X3S0(CODE, MT, NT, KP, KS) {CODE, 32 * KS * KP, MT, NT, KP, KS, {gemm_x3s_kernel<MT, NT, KP, KS, 0>, nullptr, nullptr}, \
                                    {gemm_x3s_kernel_v<MT, NT, KP, KS, 0>, nullptr, nullptr}}
// 44: 32 x 48 tiles, two K halves, 12 waves (the wide outputs: to_qkv, ff.net.0, the ff.net.3 dgrad; K = C <= 384).  At
//     ghn3xlm16 / 256 rows: 192 (N = 3C) or 256 (N = 4C) workgroups, one per CU, one round.
// 45: 16 x 32 tiles, K in 2 .. 8 parts (the narrow outputs: to_out, ff.net.3 and the dgrads of to_qkv / ff.net.0 / to_out):
//     192 workgroups at ghn3xlm16 / 256 rows; the LayerNorm prologues need whole rows in a few lanes: K <= 384
const Cfg g_cfg[] = {
    X3S(44, 2, 3, 2, 1), X3S(44, 2, 3, 2, 2), X3S(44, 2, 3, 2, 3), X3S(44, 2, 3, 2, 4), X3S(44, 2, 3, 2, 6),
    X3S(45, 1, 2, 2, 1), X3S(45, 1, 2, 4, 1), X3S(45, 1, 2, 2, 3), X3S(45, 1, 2, 4, 2), X3S(45, 1, 2, 4, 3),
    X3S0(45, 1, 2, 8, 2), X3S0(45, 1, 2, 8, 3), X3S0(45, 1, 2, 8, 4), X3S0(45, 1, 2, 6, 6), X3S0(45, 1, 2, 8, 6),
};
constexpr int kNCfg = sizeof(g_cfg) / sizeof(g_cfg[0]);
bool g_ready = false;

const Cfg* find_cfg(int code, int K, int pro) {
    for (int i = 0; i < kNCfg; ++i)
        if (g_cfg[i].code == code && g_cfg[i].K == K && g_cfg[i].fn[pro]) return g_cfg + i;
    return nullptr;
}
int lds_bytes(const Cfg* c) { return c->K * 16 * c->mt * 4 + (c->kp - 1) * c->mt * c->nt * 64 * 16; }

}  // namespace

int ghn3_gemm_x3s_init() {
    if (g_ready) return GHN3_OK;
    for (int i = 0; i < kNCfg; ++i)
        for (int p = 0; p < 3; ++p) {
            if (!g_cfg[i].fn[p]) continue;
            hipError_t e = hipFuncSetAttribute((const void*)g_cfg[i].fn[p], hipFuncAttributeMaxDynamicSharedMemorySize,
                                               160 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)g_cfg[i].vfn[p], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) { ghn3_set_error("hipFuncSetAttribute(x3s): %s", hipGetErrorString(e)); return GHN3_E_HIP; }
        }
    g_ready = true;
    return GHN3_OK;
}

// tile edges of (tile code 44 | 45, K, ln_kind) or 0 when no kernel is instantiated for it
int ghn3_gemm_x3s_tile(int code, int K, int ln_kind, int* bm, int* bn) {
    const Cfg* c = (ln_kind >= 0 && ln_kind <= 2) ? find_cfg(code, K, ln_kind) : nullptr;
    if (!c) return 0;
    *bm = 16 * c->mt; *bn = 16 * c->nt;
    return 1;
}

int ghn3_gemm_x3s_launch(const GemmProbDev* d_probs, const GemmProbDev* h_probs, int n_probs, int total_tiles, int code, int K,
                         int ln_kind, hipStream_t stream) {
    if (n_probs <= 0 || total_tiles <= 0) return GHN3_OK;
    const Cfg* c = (ln_kind >= 0 && ln_kind <= 2) ? find_cfg(code, K, ln_kind) : nullptr;
    if (!c) {
        ghn3_set_error("x3 staged gemm: no kernel for tile code %d, K = %d, ln_kind %d", code, K, ln_kind);
        return GHN3_E_LIMIT;
    }
    // (h_probs: the host copy of the same table, when the caller has one)
    if (n_probs == 1 && h_probs)
        hipLaunchKernelGGL(c->vfn[ln_kind], dim3(total_tiles), dim3(64 * c->mt * c->nt * c->kp), lds_bytes(c), stream, h_probs[0]);
    else
        hipLaunchKernelGGL(c->fn[ln_kind], dim3(total_tiles), dim3(64 * c->mt * c->nt * c->kp), lds_bytes(c), stream, d_probs, n_probs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ghn3_set_error("x3 staged gemm launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    return GHN3_OK;
}
