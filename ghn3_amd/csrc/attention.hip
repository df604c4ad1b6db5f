// Fused multi-head self-attention with additive edge bias on the fp32 matrix cores (gfx950), N <= 4096, d <= 32
// (score rows held in registers up to N = 1024, streamed twice beyond).
//
// Replaces ghn3/graphormer.py:121-140:
//     attn = (q @ k^T) * d^-0.5 + edge_bias ; attn.masked_fill(~mask, -2**15) ; softmax ; attn @ v
// and its autograd backward.  Head dims on this path are tiny (d = 8..24) and N <= ~10^3; all four
// products of the forward and backward run on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).
//
// Layout trick.  Every score-shaped 32 x 32 tile is computed TRANSPOSED: S^T = K_tile Q_tile^T, so that in the MFMA
// C/D layout (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) a LANE is a query and its 16 accumulator
// REGISTERS are keys.  Then
//   * the row softmax is a per-lane reduction over registers (+ one xor-32 shuffle + a 4-wave LDS exchange);
//   * P^T (or dS^T) is already the B operand of the next product O^T = V^T P^T (dQ^T = K^T dS^T): MFMA step s
//     consumes accumulator register s, the matching A operand row is key (s & 3) + 8 (s >> 2) + 4 (lane >> 5)
//     -- no LDS round trip, no cross-lane movement between the two GEMMs of the fused op;
//   * the output tile O^T has e (head column) in registers as 4 groups of 4 consecutive e: float4 stores.
// The backward runs both reductions in ONE launch: "row" workgroups own 32 queries (dQ, dBias), "column"
// workgroups own 32 keys (dK, dV, tiles kept un-transposed so that a lane is a key); both recompute
// dP = dO V^T on the matrix cores, so nothing but P (written once by the forward) is exchanged through memory.
// A workgroup is 4 waves that split the tiles of the other dimension and combine their partial outputs in LDS.
//
// Mask semantics (quirks Q5/Q6): pair mask = valid(i) & valid(j); masked scores are set to -32768 (not -inf), so
// fully padded query rows produce a uniform distribution over all N_max keys, as the reference.

#include "ghn3_internal.h"

#define ATT_DMAX 32
#define ATT_NMAX 4096
// tools/attn_probe.hip builds this file with GHN3_ATTN_PROBE: wave 0 of block (0,0,0) leaves shader-clock stamps
#ifdef GHN3_ATTN_PROBE
__device__ long long g_attn_stamps[64];
#define ATT_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) \
    g_attn_stamps[stamp_base + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
// (backward: the first row-role block -> slots 16.., the first column-role block -> slots 32..)
#define ATT_BSTAMP(col, i) do { if ((int)blockIdx.x == ((col) ? NB : 0) && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) \
    g_attn_stamps[((col) ? 32 : 16) + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define ATT_STAMP(i) do { } while (0)
#define ATT_BSTAMP(col, i) do { } while (0)
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int acc_row(int r, int lhi) { return (r & 3) + 8 * (r >> 2) + 4 * lhi; }
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// Row operand of the MFMA (lane = matrix row l31, k index = 2 s + lhi): x[s] = row[2 s + lhi] for 2 s + lhi < d.
// vec: d % 4 == 0 and the row is 16-byte aligned -> d / 4 float4 loads (the two half-wave lanes of a row load the
// same vectors and keep alternate elements).
// Every load is UNCONDITIONAL (lanes without a row read `safe`, a row that always exists; offsets clamped into the row; the
// value selected afterwards): a load inside an if-block is waited for at the end of that block (s_waitcnt vmcnt(0) at the
// join), which turned the KS / 2 loads of a row into KS / 2 dependent round trips -- 12 of them in front of the first MFMA
// of the attention backward.
template <int KS>
__device__ __forceinline__ void load_row_operand(const float* __restrict__ row, const float* __restrict__ safe, int d,
                                                 int lhi, bool vec, float (&x)[KS]) {
    const bool ok = row != nullptr;
    const float* r = ok ? row : safe;
    if (vec) {                                       // (d >= 4)
        f32x4 f[KS / 2];
#pragma unroll
        for (int c = 0; c < KS / 2; ++c) f[c] = *reinterpret_cast<const f32x4*>(r + min(4 * c, d - 4));
#pragma unroll
        for (int c = 0; c < KS / 2; ++c) {
            const bool on = ok && 4 * c < d;
            x[2 * c] = on ? (lhi ? f[c].y : f[c].x) : 0.f;
            x[2 * c + 1] = on ? (lhi ? f[c].w : f[c].z) : 0.f;
        }
    } else {
        float v[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) v[s] = r[min(2 * s + lhi, d - 1)];
#pragma unroll
        for (int s = 0; s < KS; ++s) x[s] = (ok && 2 * s + lhi < d) ? v[s] : 0.f;
    }
}

// This lane's 16 values of a 32 x 32 fp32 tile (accumulator layout: register 4 g + c <-> column 8 g + 4 lhi + c of the row
// `rowp` points into at the tile's first column j0; nullptr: no row, zeros).  Interior tiles (uniform: 16-byte rows, all 32
// columns below N) load their four vectors unconditionally -- see load_row_operand -- from `safe` for lanes without a row.
__device__ __forceinline__ f32x16 load_tile_row(const float* __restrict__ rowp, const float* __restrict__ safe, int j0, int N,
                                                int lhi, bool pvec) {
    f32x16 t = zero16();
    if (pvec && j0 + 32 <= N) {
        const bool ok = rowp != nullptr;
        const float* r = ok ? rowp : safe;
        f32x4 f[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) f[g] = *reinterpret_cast<const f32x4*>(r + 8 * g + 4 * lhi);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            t[4 * g] = ok ? f[g].x : 0.f; t[4 * g + 1] = ok ? f[g].y : 0.f;
            t[4 * g + 2] = ok ? f[g].z : 0.f; t[4 * g + 3] = ok ? f[g].w : 0.f;
        }
        return t;
    }
    if (rowp) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int jj = 8 * g + 4 * lhi;
            if (pvec && j0 + jj + 3 < N) {
                const f32x4 f = *reinterpret_cast<const f32x4*>(rowp + jj);
                t[4 * g] = f.x; t[4 * g + 1] = f.y; t[4 * g + 2] = f.z; t[4 * g + 3] = f.w;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (j0 + jj + c < N) t[4 * g + c] = rowp[jj + c];
            }
        }
    }
    return t;
}

// Column operand (lane = head column e = l31, MFMA step s <-> matrix row acc_row(s, lhi) of the 32-row tile that
// starts at row0): x[s] = X[(row0 + acc_row(s, lhi)) * stride + e]
__device__ __forceinline__ void load_col_operand(const float* __restrict__ X, int row0, int n_rows, size_t stride,
                                                 int e, int d, int lhi, float (&x)[16]) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int row = row0 + acc_row(s, lhi);
        x[s] = (e < d && row < n_rows) ? X[(size_t)row * stride + e] : 0.f;
    }
}

// Two-phase operand loads of the backward kernel: *_issue only issues the (unconditional, clamped) loads, *_finish applies
// the validity selects.  A kernel issues the loads of ALL operands of a tile first and finishes them afterwards: the selects of
// one operand in front of the loads of the next made every operand its own dependent round trip (stamps of tools/attn_probe:
// 15k of the column role's 31k cycles went by before the first MFMA).
template <int KS> struct RowRaw { f32x4 f[KS / 2]; float s[KS]; };
template <int KS>
__device__ __forceinline__ void row_issue(RowRaw<KS>& R, const float* __restrict__ r, int d, int lhi, bool vec) {
    if (vec) {
#pragma unroll
        for (int c = 0; c < KS / 2; ++c) R.f[c] = *reinterpret_cast<const f32x4*>(r + min(4 * c, d - 4));
    } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) R.s[s] = r[min(2 * s + lhi, d - 1)];
    }
}
template <int KS>
__device__ __forceinline__ void row_finish(const RowRaw<KS>& R, bool ok, int d, int lhi, bool vec, float (&x)[KS]) {
    if (vec) {
#pragma unroll
        for (int c = 0; c < KS / 2; ++c) {
            const bool on = ok && 4 * c < d;
            x[2 * c] = on ? (lhi ? R.f[c].y : R.f[c].x) : 0.f;
            x[2 * c + 1] = on ? (lhi ? R.f[c].w : R.f[c].z) : 0.f;
        }
    } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) x[s] = (ok && 2 * s + lhi < d) ? R.s[s] : 0.f;
    }
}
struct ColRaw { float v[16]; };
__device__ __forceinline__ void col_issue(ColRaw& R, const float* __restrict__ X, int row0, int n_rows, size_t stride, int e,
                                          int d, int lhi) {
    const int ec = min(e, d - 1);
#pragma unroll
    for (int s = 0; s < 16; ++s) R.v[s] = X[(size_t)min(row0 + acc_row(s, lhi), n_rows - 1) * stride + ec];
}
__device__ __forceinline__ void col_finish(const ColRaw& R, int row0, int n_rows, int e, int d, int lhi, float (&x)[16]) {
#pragma unroll
    for (int s = 0; s < 16; ++s) x[s] = (e < d && row0 + acc_row(s, lhi) < n_rows) ? R.v[s] : 0.f;
}
// (interior tiles only -- see load_tile_row; other tiles are loaded by tile_finish itself)
struct TileRaw { f32x4 f[4]; };
__device__ __forceinline__ void tile_issue(TileRaw& R, const float* __restrict__ r, int lhi, bool interior) {
    if (interior) {
#pragma unroll
        for (int g = 0; g < 4; ++g) R.f[g] = *reinterpret_cast<const f32x4*>(r + 8 * g + 4 * lhi);
    }
}
__device__ __forceinline__ f32x16 tile_finish(const TileRaw& R, const float* __restrict__ rowp, const float* __restrict__ safe,
                                              bool interior, int j0, int N, int lhi, bool pvec) {
    if (!interior) return load_tile_row(rowp, safe, j0, N, lhi, pvec);
    const bool ok = rowp != nullptr;
    f32x16 t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        t[4 * g] = ok ? R.f[g].x : 0.f; t[4 * g + 1] = ok ? R.f[g].y : 0.f;
        t[4 * g + 2] = ok ? R.f[g].z : 0.f; t[4 * g + 3] = ok ? R.f[g].w : 0.f;
    }
    return t;
}

// Cooperative, coalesced copy of the block's head slices into LDS: 32 query rows (from row i0), then `nkeys` key rows and
// `nkeys` value rows (from row 0); rows beyond N are zero.  LDS rows of ldw floats, one lane per 16-byte chunk
// (consecutive lanes read consecutive chunks of a row, then the next row: 96-byte runs at d = 24 instead of one cache
// line per lane), and EVERY load is issued before the first LDS write: one memory round trip for the whole prologue
// (copying in batches measured one dependent round trip per batch: 6.5k of the kernel's 20k cycles).
// (Segment by segment, RPP = NT / 8 rows per pass -- at most 8 chunks per row: d <= 32 --, one pointer per thread advanced by a
// constant per pass: the former single loop over a virtual row index paid a branch chain, a 64-bit multiply and an integer
// division per load; its address arithmetic was most of the 6.7k cycles the forward spent in front of its first MFMA.)
template <int PASSES> struct SegRaw { f32x4 v[PASSES]; };
template <int PASSES, int RPP>
__device__ __forceinline__ void seg_issue(SegRaw<PASSES>& R, const float* __restrict__ src, size_t stride, int first, int n_rows,
                                          int N, int r0, int c4, bool on) {
    const float* p = src + (size_t)(first + r0) * stride + c4;
    const size_t step = (size_t)RPP * stride;
#pragma unroll
    for (int g = 0; g < PASSES; ++g) {
        const int row = g * RPP + r0;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        R.v[g] = z;
        if (on && row < n_rows && first + row < N) R.v[g] = *reinterpret_cast<const f32x4*>(p);
        p += step;
    }
}
template <int PASSES, int RPP>
__device__ __forceinline__ void seg_write(const SegRaw<PASSES>& R, float* __restrict__ dst, int ldw, int n_rows, int r0, int c4,
                                          bool on) {
#pragma unroll
    for (int g = 0; g < PASSES; ++g) {
        const int row = g * RPP + r0;
        if (on && row < n_rows) *reinterpret_cast<f32x4*>(dst + row * ldw + c4) = R.v[g];
    }
}

template <int NT, int MAXK>
__device__ __forceinline__ void stage_head_slices(float* __restrict__ dst, int ldw, const float* __restrict__ base,
                                                  int C, int i0, int nkeys, int N, int d, bool vec, int tid) {
    const size_t stride = (size_t)3 * C;
    if (vec) {
        constexpr int RPP = NT / 8;
        const int chunks = d >> 2;
        const int r0 = tid / chunks, c4 = 4 * (tid - r0 * chunks);
        const bool on = r0 < RPP;
        SegRaw<(32 + RPP - 1) / RPP> q;
        SegRaw<(MAXK + RPP - 1) / RPP> k, v;
        seg_issue<(32 + RPP - 1) / RPP, RPP>(q, base, stride, i0, 32, N, r0, c4, on);
        seg_issue<(MAXK + RPP - 1) / RPP, RPP>(k, base + C, stride, 0, nkeys, N, r0, c4, on);
        seg_issue<(MAXK + RPP - 1) / RPP, RPP>(v, base + 2 * C, stride, 0, nkeys, N, r0, c4, on);
        seg_write<(32 + RPP - 1) / RPP, RPP>(q, dst, ldw, 32, r0, c4, on);
        seg_write<(MAXK + RPP - 1) / RPP, RPP>(k, dst + 32 * ldw, ldw, nkeys, r0, c4, on);
        seg_write<(MAXK + RPP - 1) / RPP, RPP>(v, dst + (32 + nkeys) * ldw, ldw, nkeys, r0, c4, on);
    } else {
        const int rows = 32 + 2 * nkeys;
        auto source = [&](int r) -> const float* {      // nullptr: zero row
            if (r < 32) return i0 + r < N ? base + (size_t)(i0 + r) * stride : nullptr;
            r -= 32;
            if (r < nkeys) return r < N ? base + (size_t)r * stride + C : nullptr;
            r -= nkeys;
            return r < N ? base + (size_t)r * stride + 2 * C : nullptr;
        };
        for (int f = tid; f < rows * d; f += NT) {
            const int r = f / d, e = f - r * d;
            const float* p = source(r);
            dst[r * ldw + e] = p ? p[e] : 0.f;
        }
    }
}

template <int KS>
__device__ __forceinline__ f32x16 mfma_rows(const float (&a)[KS], const float (&b)[KS], f32x16 acc) {
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ f32x16 mfma_cols(const float (&a)[16], const f32x16& b, f32x16 acc) {
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    return acc;
}

// Sum the NW waves' 32 x 32 partial tiles through LDS; wave w < 4 returns register group w (4 consecutive head columns
// e = 8 w + 4 lhi + c of matrix column l31) of the total (waves 4 .. NW - 1 only contribute).
template <int NW = 4>
__device__ __forceinline__ f32x4 reduce_waves(float* red /* [NW][16][64] */, const f32x16& part, int w, int lane) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(w * 16 + r) * 64 + lane] = part[r];
    __syncthreads();
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (w < 4) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int r = 4 * w + c;
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < NW; ++q) t += red[(16 * q + r) * 64 + lane];
            o[c] = t;
        }
    }
    return o;
}

// dst[e0 .. e0 + 3] = v (only e < d), vectorised when possible
__device__ __forceinline__ void store4(float* __restrict__ dst, int e0, int d, bool vec, f32x4 v) {
    if (vec && e0 + 3 < d) {
        *reinterpret_cast<f32x4*>(dst + e0) = v;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (e0 + c < d) dst[e0 + c] = v[c];
    }
}

// ------------------------------------------------------------------------------------------------
// forward: grid (ceil(N / 32), H, B), 256 threads; wave w owns key tiles w, w + 4, ... (TPW of them)
// ------------------------------------------------------------------------------------------------
template <int KS, int TPW, int QB, int NW = 4>
__device__ __forceinline__ void attn_fwd_body(float* __restrict__ out, const float* __restrict__ qkv,
                                              const float* __restrict__ bias, float* __restrict__ Psave,
                                              const int* __restrict__ n_nodes, int N, int C, int H, float scale, int vec,
                                              int stamp_base) {
    __shared__ float red[NW * 16 * 64];
    __shared__ float red_m[NW][32], red_l[NW][32];
    extern __shared__ __attribute__((aligned(16))) float stage[];   // TPW <= 2: Q | K | V head slices (see below)
    const int d = C / H;
    const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * QB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int nb = n_nodes[b];
    const float* base = qkv + (size_t)b * N * 3 * C + h * d;
    const size_t bh = ((size_t)b * H + h) * N;
    // QB = 16: a block owns 16 queries (lanes 16-31 of the 32-row MFMA tiles idle) -- twice the workgroups, so that a
    // short graph's launch covers every CU: the kernel is bound by what ONE CU can fetch (~14 B/clk, the K / V slices
    // of all keys + the bias rows + the probabilities of its queries), not by the matrix cores
    const int qi = l31 < QB ? i0 + l31 : N;                     // this lane's query (N: none)
    const bool vq = (d & 3) == 0 && vec;

    ATT_STAMP(0);
    // PRE (N <= 256): the head slices of the block's 32 queries and of ALL keys / values are staged in LDS with
    // coalesced loads (one pass, every load in flight at once) and the MFMA operands are read from there; larger N
    // keeps only the score tiles in registers and loads operands per tile from memory.
    constexpr bool PRE = TPW <= 2;
    constexpr int TP = PRE ? TPW : 1;
    const int ldw = ((d + 3) & ~3) + 4;                          // padded LDS row (16-byte aligned for the float4 copy)
    const int nkeys = TPW * NW * 32;                             // keys covered by the block's NW waves
    float* Qs = stage;
    float* Ks = stage + 32 * ldw;
    float* Vs = Ks + nkeys * ldw;
    float qb[KS];
    float ka[TP][KS];
    // edge bias of this lane's query row: register r <-> key tile * 32 + acc_row(r, lhi) (4 runs of 4 keys); the loads of
    // interior tiles are issued in front of the staging loads (one round trip for both), see tile_issue
    f32x16 S[TPW];
    const bool pvec = (N & 3) == 0 && vec;
    TileRaw bR[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        const int j0 = (w + NW * k) * 32;
        if (bias && j0 < N)                          // (uniform)
            tile_issue(bR[k], (qi < N ? bias + (bh + qi) * N : bias + bh * N) + j0, lhi, pvec && j0 + 32 <= N);
    }
    if (PRE) {
        stage_head_slices<64 * NW, TPW * NW * 32>(stage, ldw, base, C, i0, nkeys, N, d, vq, tid);
    } else {
        load_row_operand<KS>(qi < N ? base + (size_t)qi * 3 * C : nullptr, base, d, lhi, vq, qb);
    }
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        S[k] = zero16();
        const int j0 = (w + NW * k) * 32;
        if (bias && j0 < N)                          // (uniform)
            S[k] = tile_finish(bR[k], qi < N ? bias + (bh + qi) * N + j0 : nullptr, bias + bh * N + j0, pvec && j0 + 32 <= N,
                               j0, N, lhi, pvec);
    }
    float va[TP][16];
    if (PRE) {
        __syncthreads();                                         // the staged slices are complete
#pragma unroll
        for (int sx = 0; sx < KS; ++sx) {
            const int kx = 2 * sx + lhi;
            qb[sx] = kx < d ? Qs[l31 * ldw + kx] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < TP; ++k) {
            const int j0 = (w + NW * k) * 32;
#pragma unroll
            for (int sx = 0; sx < KS; ++sx) {
                const int kx = 2 * sx + lhi;
                ka[k][sx] = kx < d ? Ks[(j0 + l31) * ldw + kx] : 0.f;
            }
#pragma unroll
            for (int sx = 0; sx < 16; ++sx)
                va[k][sx] = l31 < d ? Vs[(j0 + acc_row(sx, lhi)) * ldw + l31] : 0.f;
        }
    }

    ATT_STAMP(1);
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        const f32x16 bia = S[k];
        const int j0 = (w + NW * k) * 32;
        if (!PRE) {
            const int j = j0 + l31;
            load_row_operand<KS>(j < N ? base + (size_t)j * 3 * C + C : nullptr, base, d, lhi, vq, ka[0]);
        }
        f32x16 acc = mfma_rows<KS>(ka[PRE ? k : 0], qb, zero16());
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + acc_row(r, lhi);
            float s = acc[r] * scale + bia[r];
            if (!(qi < nb && j < nb)) s = -32768.f;
            if (j >= N) s = -INFINITY;
            acc[r] = s;
            mx = fmaxf(mx, s);
        }
        S[k] = acc;
    }
    ATT_STAMP(2);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (lhi == 0) red_m[w][l31] = mx;
    __syncthreads();
    ATT_STAMP(3);
    mx = red_m[0][l31];
#pragma unroll
    for (int q = 1; q < NW; ++q) mx = fmaxf(mx, red_m[q][l31]);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < TPW; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e_ = __expf(S[k][r] - mx);               // exp(-inf) = 0 for keys beyond N
            S[k][r] = e_;
            sum += e_;
        }
    sum += __shfl_xor(sum, 32, 64);
    if (lhi == 0) red_l[w][l31] = sum;
    __syncthreads();
    ATT_STAMP(4);
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) tot += red_l[q][l31];
    const float inv = 1.f / tot;
    f32x16 O = zero16();
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        const int j0 = (w + NW * k) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[k][r] *= inv;
        if (Psave && qi < N && j0 < N) {
            float* prow = Psave + (bh + qi) * N + j0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int jj = 8 * g + 4 * lhi;
                if (pvec && j0 + jj + 3 < N) {
                    f32x4 f = {S[k][4 * g], S[k][4 * g + 1], S[k][4 * g + 2], S[k][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(prow + jj) = f;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (j0 + jj + c < N) prow[jj + c] = S[k][4 * g + c];
                }
            }
        }
        if (!PRE && j0 < N) load_col_operand(base + 2 * C, j0, N, (size_t)3 * C, l31, d, lhi, va[0]);
        if (j0 < N) O = mfma_cols(va[PRE ? k : 0], S[k], O);       // O^T += V^T P^T
    }
    ATT_STAMP(5);
    const f32x4 o = reduce_waves<NW>(red, O, w, lane);
    ATT_STAMP(6);
    if (w < 4 && qi < N) store4(out + ((size_t)b * N + qi) * C + h * d, 8 * w + 4 * lhi, d, vq, o);
    ATT_STAMP(7);
}
template <int KS, int TPW, int QB, int NW = 4>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(float* __restrict__ out, const float* __restrict__ qkv,
                                                          const float* __restrict__ bias, float* __restrict__ Psave,
                                                          const int* __restrict__ n_nodes, int N, int C, int H,
                                                          float scale, int vec) {
    attn_fwd_body<KS, TPW, QB, NW>(out, qkv, bias, Psave, n_nodes, N, C, H, scale, vec, 0);
}

// ------------------------------------------------------------------------------------------------
// forward for N > 1024 (graphs of the largest torchvision networks): same tiles, but the score row no longer fits the
// register file, so the wave streams its key tiles twice -- pass 1 keeps a running (max, sum) per query (online
// softmax), pass 2 recomputes the scores, normalises, stores P and accumulates O^T.  The extra Q K^T product is ~d/N
// of the step; nothing but P is written.
// ------------------------------------------------------------------------------------------------
template <int KS>
__device__ __forceinline__ f32x16 score_tile(const float* __restrict__ base, const float* __restrict__ bias_row,
                                             const float (&qb)[KS], int j0, int N, int C, int d, int nb, int qi,
                                             int l31, int lhi, bool vq, float scale) {
    float ka[KS];
    const int j = j0 + l31;
    load_row_operand<KS>(j < N ? base + (size_t)j * 3 * C + C : nullptr, base, d, lhi, vq, ka);
    f32x16 acc = mfma_rows<KS>(ka, qb, zero16());
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int jr = j0 + acc_row(r, lhi);
        float s_ = acc[r] * scale + ((bias_row && qi < N && jr < N) ? bias_row[jr] : 0.f);
        if (!(qi < nb && jr < nb)) s_ = -32768.f;
        if (jr >= N) s_ = -INFINITY;
        acc[r] = s_;
    }
    return acc;
}

template <int KS>
__global__ __launch_bounds__(256) void attn_fwd_stream_kernel(float* __restrict__ out, const float* __restrict__ qkv,
                                                              const float* __restrict__ bias,
                                                              float* __restrict__ Psave,
                                                              const int* __restrict__ n_nodes, int N, int C, int H,
                                                              float scale, int vec) {
    __shared__ float red[4 * 16 * 64];
    __shared__ float red_m[4][32], red_l[4][32];
    const int d = C / H;
    const int NB = (N + 31) >> 5;
    const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int nb = n_nodes[b];
    const float* base = qkv + (size_t)b * N * 3 * C + h * d;
    const size_t bh = ((size_t)b * H + h) * N;
    const int qi = i0 + l31;
    const bool vq = (d & 3) == 0 && vec;
    const float* brow = (bias && qi < N) ? bias + (bh + qi) * N : nullptr;
    float qb[KS];
    load_row_operand<KS>(qi < N ? base + (size_t)qi * 3 * C : nullptr, base, d, lhi, vq, qb);

    float mx = -INFINITY, sum = 0.f;
    for (int t = w; t < NB; t += 4) {
        const f32x16 sc = score_tile<KS>(base, brow, qb, t * 32, N, C, d, nb, qi, l31, lhi, vq, scale);
        float tm = mx;
#pragma unroll
        for (int r = 0; r < 16; ++r) tm = fmaxf(tm, sc[r]);
        if (tm > -INFINITY) {
            float acc = sum * __expf(mx - tm);                   // (mx = -inf on the first tile: exp(-inf) = 0)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc += __expf(sc[r] - tm);
            sum = acc;
            mx = tm;
        }
    }
    {   // the two half-waves of a query hold different keys
        const float m2 = __shfl_xor(mx, 32, 64), s2 = __shfl_xor(sum, 32, 64);
        const float mn = fmaxf(mx, m2);
        sum = (mx > -INFINITY ? sum * __expf(mx - mn) : 0.f) + (m2 > -INFINITY ? s2 * __expf(m2 - mn) : 0.f);
        mx = mn;
    }
    if (lhi == 0) { red_m[w][l31] = mx; red_l[w][l31] = sum; }
    __syncthreads();
    const float gm = fmaxf(fmaxf(red_m[0][l31], red_m[1][l31]), fmaxf(red_m[2][l31], red_m[3][l31]));
    float gl = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (red_m[k][l31] > -INFINITY) gl += red_l[k][l31] * __expf(red_m[k][l31] - gm);
    const float inv = 1.f / gl;

    f32x16 O = zero16();
    for (int t = w; t < NB; t += 4) {
        const int j0 = t * 32;
        f32x16 sc = score_tile<KS>(base, brow, qb, j0, N, C, d, nb, qi, l31, lhi, vq, scale);
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = __expf(sc[r] - gm) * inv;
        if (Psave && qi < N) {
            float* prow = Psave + (bh + qi) * N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jr = j0 + acc_row(r, lhi);
                if (jr < N) prow[jr] = sc[r];
            }
        }
        float va[16];
        load_col_operand(base + 2 * C, j0, N, (size_t)3 * C, l31, d, lhi, va);
        O = mfma_cols(va, sc, O);
    }
    const f32x4 o = reduce_waves(red, O, w, lane);
    if (qi < N) store4(out + ((size_t)b * N + qi) * C + h * d, 8 * w + 4 * lhi, d, vq, o);
}

// ------------------------------------------------------------------------------------------------
// backward: grid (2 * ceil(N / 32), H, B); blockIdx.x < NB: row role (32 queries: dQ, dBias += dS),
// otherwise column role (32 keys: dK, dV).  delta_i = sum_e dO[i][e] O[i][e] (= rowsum(P * dP)).
// ------------------------------------------------------------------------------------------------
// NW waves per workgroup split the tiles of the other dimension.  NW = 8 for graphs of up to 256 nodes (one tile per wave:
// no dependent second round of loads, twice the fetch rate of a 4-wave workgroup): 16.3 -> 14.8 us alone (round 3), adopted
// in round 4 when the W2 weight gradient left the side stream and the chain runs undisturbed.
template <int KS, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void attn_bwd_kernel(float* __restrict__ dqkv, const float* __restrict__ dO,
                                                       const float* __restrict__ qkv, const float* __restrict__ P,
                                                       const float* __restrict__ Oin, float* __restrict__ dBias,
                                                       const int* __restrict__ n_nodes, int N, int C, int H,
                                                       float scale, int vec, float* __restrict__ amax_out) {
    __shared__ float red[NW * 16 * 64];
    __shared__ float dl[NW][32];
    float bmx = 0.f;                                     // max |dBias written| (the last layer's launch: amax_out)
    const int d = C / H;
    const int NB = (N + 31) >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int nb = n_nodes[b];
    const float* base = qkv + (size_t)b * N * 3 * C + h * d;
    const float* dOb = dO + (size_t)b * N * C + h * d;
    const float* Ob = Oin + (size_t)b * N * C + h * d;
    const size_t bh = ((size_t)b * H + h) * N;
    const bool vq = (d & 3) == 0 && vec;
    const bool pvec = (N & 3) == 0 && vec;

    if ((int)blockIdx.x < NB) {
        // ---------------- row role: lane = query qi, accumulator registers = keys ----------------
        const int qi = blockIdx.x * 32 + l31;
        ATT_BSTAMP(0, 0);
        const bool qok = qi < N;
        const float* prow0 = P + bh * N;                 // (row 0 of this head: a row that always exists)
        RowRaw<KS> gbR, obR;
        row_issue(gbR, qok ? dOb + (size_t)qi * C : dOb, d, lhi, vq);
        row_issue(obR, qok ? Ob + (size_t)qi * C : Ob, d, lhi, vq);
        float gb[KS], ob[KS];
        float delta = 0.f;
        bool first = true;
        f32x16 dQ = zero16();
        for (int t = w; t < NB; t += NW) {
            const int j0 = t * 32;
            const int j = j0 + l31;
            const bool interior = pvec && j0 + 32 <= N;
            // every load of the tile is in flight (the first tile's together with the dO / O rows) before anything is used
            RowRaw<KS> vaR;
            ColRaw kcR;
            TileRaw pR, dbR;
            row_issue(vaR, j < N ? base + (size_t)j * 3 * C + 2 * C : base, d, lhi, vq);
            col_issue(kcR, base + C, j0, N, (size_t)3 * C, l31, d, lhi);
            tile_issue(pR, (qok ? P + (bh + qi) * N : prow0) + j0, lhi, interior);
            if (dBias) tile_issue(dbR, (qok ? dBias + (bh + qi) * N : dBias + bh * N) + j0, lhi, interior);
            if (first) {
                row_finish(gbR, qok, d, lhi, vq, gb);
                row_finish(obR, qok, d, lhi, vq, ob);
#pragma unroll
                for (int s = 0; s < KS; ++s) delta += gb[s] * ob[s];
                delta += __shfl_xor(delta, 32, 64);
                first = false;
                ATT_BSTAMP(0, 1);
            }
            float va[KS], kc[16];
            row_finish(vaR, j < N, d, lhi, vq, va);
            col_finish(kcR, j0, N, l31, d, lhi, kc);
            const f32x16 p = tile_finish(pR, qok ? P + (bh + qi) * N + j0 : nullptr, prow0 + j0, interior, j0, N, lhi, pvec);
            f32x16 db = zero16();
            if (dBias)                               // (uniform)
                db = tile_finish(dbR, qok ? dBias + (bh + qi) * N + j0 : nullptr, dBias + bh * N + j0, interior, j0, N, lhi, pvec);
            ATT_BSTAMP(0, 2);
            f32x16 ds = mfma_rows<KS>(va, gb, zero16());           // dP^T = V dO^T
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jr = j0 + acc_row(r, lhi);
                float x = p[r] * (ds[r] - delta);
                if (!(qi < nb && jr < nb)) x = 0.f;               // masked_fill blocks the gradient
                ds[r] = x;
            }
            if (dBias && qi < N) {
                float* brow = dBias + (bh + qi) * N + j0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int jj = 8 * g + 4 * lhi;
                    if (pvec && j0 + jj + 3 < N) {
                        f32x4 f = {db[4 * g] + ds[4 * g], db[4 * g + 1] + ds[4 * g + 1], db[4 * g + 2] + ds[4 * g + 2],
                                   db[4 * g + 3] + ds[4 * g + 3]};
                        *reinterpret_cast<f32x4*>(brow + jj) = f;
                        bmx = fmaxf(fmaxf(bmx, fmaxf(fabsf(f.x), fabsf(f.y))), fmaxf(fabsf(f.z), fabsf(f.w)));
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (j0 + jj + c < N) { const float f = db[4 * g + c] + ds[4 * g + c]; brow[jj + c] = f; bmx = fmaxf(bmx, fabsf(f)); }
                    }
                }
            }
            ATT_BSTAMP(0, 3);
            dQ = mfma_cols(kc, ds, dQ);                            // dQ^T += K^T dS^T
        }
        ATT_BSTAMP(0, 4);
        if (amax_out) ghn3_atomic_amax(amax_out, bmx);        // (uniform branch: every lane of the wave takes part)
        f32x4 o = reduce_waves<NW>(red, dQ, w, lane);
        ATT_BSTAMP(0, 5);
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] *= scale;
        if (w < 4 && qi < N) store4(dqkv + ((size_t)b * N + qi) * 3 * C + h * d, 8 * w + 4 * lhi, d, vq, o);
        ATT_BSTAMP(0, 6);
    } else {
        // ---------------- column role: lane = key kj, accumulator registers = queries ----------------
        const int kj = (blockIdx.x - NB) * 32 + l31;
        ATT_BSTAMP(1, 0);
        const bool kok = kj < N;
        RowRaw<KS> vbR;
        row_issue(vbR, kok ? base + (size_t)kj * 3 * C + 2 * C : base, d, lhi, vq);
        float vb[KS];
        bool first = true;
        f32x16 dV = zero16(), dK = zero16();
        for (int t = w; t < NB; t += NW) {
            const int q0 = t * 32;
            const int qrow = q0 + l31;
            // every load of the tile in flight (the first tile's together with the V row) before anything is used
            RowRaw<KS> gaR, oaR;
            ColRaw gcR, qcR;
            float pr[16];
            row_issue(gaR, qrow < N ? dOb + (size_t)qrow * C : dOb, d, lhi, vq);
            row_issue(oaR, qrow < N ? Ob + (size_t)qrow * C : Ob, d, lhi, vq);
            col_issue(gcR, dOb, q0, N, (size_t)C, l31, d, lhi);
            col_issue(qcR, base, q0, N, (size_t)3 * C, l31, d, lhi);
            {
                const float* pc = P + bh * N + min(kj, N - 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) pr[r] = pc[(size_t)min(q0 + acc_row(r, lhi), N - 1) * N];
            }
            if (first) { row_finish(vbR, kok, d, lhi, vq, vb); first = false; }
            float ga[KS], oa[KS], gc[16], qc[16];
            row_finish(gaR, qrow < N, d, lhi, vq, ga);
            row_finish(oaR, qrow < N, d, lhi, vq, oa);
            col_finish(gcR, q0, N, l31, d, lhi, gc);
            col_finish(qcR, q0, N, l31, d, lhi, qc);
            f32x16 p;
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = (kok && q0 + acc_row(r, lhi) < N) ? pr[r] : 0.f;
            ATT_BSTAMP(1, 1);
            float dpart = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) dpart += ga[s] * oa[s];
            dpart += __shfl_xor(dpart, 32, 64);
            if (lhi == 0) dl[w][l31] = dpart;                      // wave-private hand-off: lane -> register index
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ATT_BSTAMP(1, 2);
            f32x16 ds = mfma_rows<KS>(ga, vb, zero16());           // dP = dO V^T   (rows = queries, lane = key)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ir = acc_row(r, lhi);
                float x = p[r] * (ds[r] - dl[w][ir]);
                if (!(q0 + ir < nb && kj < nb)) x = 0.f;
                ds[r] = x;
            }
            __builtin_amdgcn_wave_barrier();
            ATT_BSTAMP(1, 3);
            dV = mfma_cols(gc, p, dV);                             // dV^T += dO^T P
            dK = mfma_cols(qc, ds, dK);                            // dK^T += Q^T dS
        }
        ATT_BSTAMP(1, 4);
        const f32x4 ov = reduce_waves<NW>(red, dV, w, lane);
        __syncthreads();                                      // (one exchange buffer for both reductions)
        f32x4 ok = reduce_waves<NW>(red, dK, w, lane);
#pragma unroll
        for (int c = 0; c < 4; ++c) ok[c] *= scale;
        if (w < 4 && kj < N) {
            float* row = dqkv + ((size_t)b * N + kj) * 3 * C + h * d;
            store4(row + 2 * C, 8 * w + 4 * lhi, d, vq, ov);
            store4(row + C, 8 * w + 4 * lhi, d, vq, ok);
        }
        ATT_BSTAMP(1, 5);
    }
}

// ------------------------------------------------------------------------------------------------
// backward for graphs of up to 256 nodes (one tile of the other dimension per wave, eight waves), operands STAGED IN LDS.
// The kernel above loads MFMA operands "lane = matrix row": every wave-level load touches 32 different cache lines, ~1300 line
// requests per wave and tile -- tools/attn_probe: 19k of the row role's 29k cycles pass before the first MFMA, whatever the
// order of the loads.  Here the workgroup copies the head slices it needs (row role: its 32 dO / O rows and all V / K rows;
// column role: its 32 V rows and all dO / O / Q rows) with coalesced 16-byte loads -- consecutive lanes read consecutive
// chunks of a row, every load in flight before the first LDS write, the P / dBias tile loads in front of them -- and reads the
// operands from LDS.  Same MFMAs, same summation order: bit-identical to the kernel above.
// ------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(512, 1) void attn_bwd_staged_kernel(float* __restrict__ dqkv, const float* __restrict__ dO,
                                                                 const float* __restrict__ qkv, const float* __restrict__ P,
                                                                 const float* __restrict__ Oin, float* __restrict__ dBias,
                                                                 const int* __restrict__ n_nodes, int N, int C, int H,
                                                                 float scale, int vec, float* __restrict__ amax_out) {
    constexpr int NW = 8, RPP = 64;                  // (64 rows per staging pass: 8 chunks of 16 bytes per row at most)
    extern __shared__ float bsm[];
    float* red = bsm;                                // [NW][16][64]
    float* dl = red + NW * 16 * 64;                  // [NW][32]
    float* stage = dl + NW * 32;
    float bmx = 0.f;
    const int d = C / H;                             // (d % 4 == 0, 16-byte aligned rows: checked by the host)
    const int NB = (N + 31) >> 5, Np = NB * 32;      // (NB <= 8)
    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int nb = n_nodes[b];
    const float* base = qkv + (size_t)b * N * 3 * C + h * d;
    const float* dOb = dO + (size_t)b * N * C + h * d;
    const float* Ob = Oin + (size_t)b * N * C + h * d;
    const size_t bh = ((size_t)b * H + h) * N;
    const bool pvec = (N & 3) == 0 && vec;
    const int ldw = d + 4, chunks = d >> 2;
    const size_t s3 = (size_t)3 * C;
    const bool has = w < NB;                         // this wave's tile of the other dimension
    const int t0 = w * 32;
    const int sr0 = tid / chunks, sc4 = 4 * (tid - sr0 * chunks);      // staging: row within a pass, chunk of the row
    const bool son = sr0 < RPP;

    if ((int)blockIdx.x < NB) {
        // ---------------- row role: lane = query qi, accumulator registers = keys ----------------
        const int i0 = blockIdx.x * 32, qi = i0 + l31;
        const bool qok = qi < N;
        const int j0 = t0;
        ATT_BSTAMP(0, 0);
        const bool interior = pvec && j0 + 32 <= N;
        const float* prow0 = P + bh * N;
        TileRaw pR, dbR;
        if (has) {
            tile_issue(pR, (qok ? P + (bh + qi) * N : prow0) + j0, lhi, interior);
            if (dBias) tile_issue(dbR, (qok ? dBias + (bh + qi) * N : dBias + bh * N) + j0, lhi, interior);
        }
        // LDS rows: [0, 32) dO, [32, 64) O of the block's queries; [64, 64 + Np) V, [64 + Np, 64 + 2 Np) K of all keys
        {
            SegRaw<1> a, o_;
            SegRaw<4> v_, k_;
            seg_issue<1, RPP>(a, dOb, (size_t)C, i0, 32, N, sr0, sc4, son);
            seg_issue<1, RPP>(o_, Ob, (size_t)C, i0, 32, N, sr0, sc4, son);
            seg_issue<4, RPP>(v_, base + 2 * C, s3, 0, Np, N, sr0, sc4, son);
            seg_issue<4, RPP>(k_, base + C, s3, 0, Np, N, sr0, sc4, son);
            seg_write<1, RPP>(a, stage, ldw, 32, sr0, sc4, son);
            seg_write<1, RPP>(o_, stage + 32 * ldw, ldw, 32, sr0, sc4, son);
            seg_write<4, RPP>(v_, stage + 64 * ldw, ldw, Np, sr0, sc4, son);
            seg_write<4, RPP>(k_, stage + (64 + Np) * ldw, ldw, Np, sr0, sc4, son);
        }
        ATT_BSTAMP(0, 1);
        __syncthreads();
        ATT_BSTAMP(0, 2);
        const float* dOs = stage;
        const float* Os = stage + 32 * ldw;
        const float* Vs = stage + 64 * ldw;
        const float* Ks = Vs + Np * ldw;
        float gb[KS], ob[KS];
        float delta = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int kx = 2 * s + lhi;
            gb[s] = kx < d ? dOs[l31 * ldw + kx] : 0.f;
            ob[s] = kx < d ? Os[l31 * ldw + kx] : 0.f;
            delta += gb[s] * ob[s];
        }
        delta += __shfl_xor(delta, 32, 64);
        f32x16 dQ = zero16();
        if (has) {
            float va[KS], kc[16];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int kx = 2 * s + lhi;
                va[s] = kx < d ? Vs[(j0 + l31) * ldw + kx] : 0.f;
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) kc[s] = l31 < d ? Ks[(j0 + acc_row(s, lhi)) * ldw + l31] : 0.f;
            const f32x16 p = tile_finish(pR, qok ? P + (bh + qi) * N + j0 : nullptr, prow0 + j0, interior, j0, N, lhi, pvec);
            f32x16 db = zero16();
            if (dBias)
                db = tile_finish(dbR, qok ? dBias + (bh + qi) * N + j0 : nullptr, dBias + bh * N + j0, interior, j0, N, lhi, pvec);
            f32x16 ds = mfma_rows<KS>(va, gb, zero16());           // dP^T = V dO^T
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jr = j0 + acc_row(r, lhi);
                float x = p[r] * (ds[r] - delta);
                if (!(qi < nb && jr < nb)) x = 0.f;               // masked_fill blocks the gradient
                ds[r] = x;
            }
            if (dBias && qok) {
                float* brow = dBias + (bh + qi) * N + j0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int jj = 8 * g + 4 * lhi;
                    if (pvec && j0 + jj + 3 < N) {
                        f32x4 f = {db[4 * g] + ds[4 * g], db[4 * g + 1] + ds[4 * g + 1], db[4 * g + 2] + ds[4 * g + 2],
                                   db[4 * g + 3] + ds[4 * g + 3]};
                        *reinterpret_cast<f32x4*>(brow + jj) = f;
                        bmx = fmaxf(fmaxf(bmx, fmaxf(fabsf(f.x), fabsf(f.y))), fmaxf(fabsf(f.z), fabsf(f.w)));
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (j0 + jj + c < N) { const float f = db[4 * g + c] + ds[4 * g + c]; brow[jj + c] = f; bmx = fmaxf(bmx, fabsf(f)); }
                    }
                }
            }
            dQ = mfma_cols(kc, ds, dQ);                            // dQ^T += K^T dS^T
        }
        ATT_BSTAMP(0, 3);
        if (amax_out) ghn3_atomic_amax(amax_out, bmx);
        f32x4 o = reduce_waves<NW>(red, dQ, w, lane);
        ATT_BSTAMP(0, 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] *= scale;
        if (w < 4 && qok) store4(dqkv + ((size_t)b * N + qi) * 3 * C + h * d, 8 * w + 4 * lhi, d, true, o);
        ATT_BSTAMP(0, 5);
    } else {
        // ---------------- column role: lane = key kj, accumulator registers = queries ----------------
        const int k0 = (blockIdx.x - NB) * 32, kj = k0 + l31;
        const bool kok = kj < N;
        const int q0 = t0;
        ATT_BSTAMP(1, 0);
        float pr[16];
        if (has) {
            const float* pc = P + bh * N + min(kj, N - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) pr[r] = pc[(size_t)min(q0 + acc_row(r, lhi), N - 1) * N];
        }
        // LDS rows: [0, 32) V of the block's keys; [32, 32 + Np) dO, then O, then Q of all queries
        {
            SegRaw<1> v_;
            SegRaw<4> a, o_, q_;
            seg_issue<1, RPP>(v_, base + 2 * C, s3, k0, 32, N, sr0, sc4, son);
            seg_issue<4, RPP>(a, dOb, (size_t)C, 0, Np, N, sr0, sc4, son);
            seg_issue<4, RPP>(o_, Ob, (size_t)C, 0, Np, N, sr0, sc4, son);
            seg_issue<4, RPP>(q_, base, s3, 0, Np, N, sr0, sc4, son);
            seg_write<1, RPP>(v_, stage, ldw, 32, sr0, sc4, son);
            seg_write<4, RPP>(a, stage + 32 * ldw, ldw, Np, sr0, sc4, son);
            seg_write<4, RPP>(o_, stage + (32 + Np) * ldw, ldw, Np, sr0, sc4, son);
            seg_write<4, RPP>(q_, stage + (32 + 2 * Np) * ldw, ldw, Np, sr0, sc4, son);
        }
        ATT_BSTAMP(1, 1);
        __syncthreads();
        ATT_BSTAMP(1, 2);
        const float* Vs = stage;
        const float* dOs = stage + 32 * ldw;
        const float* Os = dOs + Np * ldw;
        const float* Qs = Os + Np * ldw;
        f32x16 dV = zero16(), dK = zero16();
        if (has) {
            float vb[KS], ga[KS], oa[KS], gc[16], qc[16];
            float dpart = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int kx = 2 * s + lhi;
                vb[s] = kx < d ? Vs[l31 * ldw + kx] : 0.f;
                ga[s] = kx < d ? dOs[(q0 + l31) * ldw + kx] : 0.f;
                oa[s] = kx < d ? Os[(q0 + l31) * ldw + kx] : 0.f;
                dpart += ga[s] * oa[s];
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int row = q0 + acc_row(s, lhi);
                gc[s] = l31 < d ? dOs[row * ldw + l31] : 0.f;
                qc[s] = l31 < d ? Qs[row * ldw + l31] : 0.f;
            }
            f32x16 p;
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = (kok && q0 + acc_row(r, lhi) < N) ? pr[r] : 0.f;
            dpart += __shfl_xor(dpart, 32, 64);
            if (lhi == 0) dl[w * 32 + l31] = dpart;                // wave-private hand-off: lane -> register index
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            f32x16 ds = mfma_rows<KS>(ga, vb, zero16());           // dP = dO V^T   (rows = queries, lane = key)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ir = acc_row(r, lhi);
                float x = p[r] * (ds[r] - dl[w * 32 + ir]);
                if (!(q0 + ir < nb && kj < nb)) x = 0.f;
                ds[r] = x;
            }
            __builtin_amdgcn_wave_barrier();
            dV = mfma_cols(gc, p, dV);                             // dV^T += dO^T P
            dK = mfma_cols(qc, ds, dK);                            // dK^T += Q^T dS
        }
        ATT_BSTAMP(1, 3);
        const f32x4 ov = reduce_waves<NW>(red, dV, w, lane);
        __syncthreads();                                      // (one exchange buffer for both reductions)
        f32x4 okk = reduce_waves<NW>(red, dK, w, lane);
#pragma unroll
        for (int c = 0; c < 4; ++c) okk[c] *= scale;
        if (w < 4 && kok) {
            float* row = dqkv + ((size_t)b * N + kj) * 3 * C + h * d;
            store4(row + 2 * C, 8 * w + 4 * lhi, d, true, ov);
            store4(row + C, 8 * w + 4 * lhi, d, true, okk);
        }
        ATT_BSTAMP(1, 4);
    }
}

// ------------------------------------------------------------------------------------------------
typedef void (*attn_fwd_fn)(float*, const float*, const float*, float*, const int*, int, int, int, float, int);
typedef void (*attn_bwd_fn)(float*, const float*, const float*, const float*, const float*, float*, const int*, int,
                            int, int, float, int, float*);

static int g_attn_fwd_waves = 8;
template <int KS> static attn_fwd_fn fwd_for(int tpw) {
    if (tpw > 8) return (attn_fwd_fn)attn_fwd_stream_kernel<KS>;          // N > 1024
    // N <= 256: eight waves with one key tile each (a workgroup fetches ~5 B/clk per wave: tools/fetch_rate_probe.hip)
    if (tpw <= 2) return g_attn_fwd_waves == 8 ? (attn_fwd_fn)attn_fwd_kernel<KS, 1, 16, 8> : (attn_fwd_fn)attn_fwd_kernel<KS, 2, 16, 4>;
    return (attn_fwd_fn)attn_fwd_kernel<KS, 8, 32, 4>;
}
static attn_fwd_fn pick_fwd(int d, int tpw) {
    if (d <= 4) return fwd_for<2>(tpw);
    if (d <= 8) return fwd_for<4>(tpw);                 // released GHN-3 head dims: 8 (T, S), 16 (L), 24 (XL)
    if (d <= 16) return fwd_for<8>(tpw);
    if (d <= 24) return fwd_for<12>(tpw);
    return fwd_for<16>(tpw);
}
static int g_attn_bwd_waves = 8;
static int g_attn_bwd_staged = 1;                    // GHN3_ATTN_BWD_STAGED=0: operands straight from memory (round 3)
template <int NW> static attn_bwd_fn pick_bwd_nw(int d) {
    if (d <= 4) return attn_bwd_kernel<2, NW>;
    if (d <= 8) return attn_bwd_kernel<4, NW>;
    if (d <= 16) return attn_bwd_kernel<8, NW>;
    if (d <= 24) return attn_bwd_kernel<12, NW>;
    return attn_bwd_kernel<16, NW>;
}
// eight waves when every wave gets at most one tile of the other dimension (N <= 256), four otherwise
static attn_bwd_fn pick_bwd(int d, int N, int* nw) {
    *nw = (g_attn_bwd_waves == 8 && N <= 256) ? 8 : 4;
    return *nw == 8 ? pick_bwd_nw<8>(d) : pick_bwd_nw<4>(d);
}

int ghn3_attn_init() {
    if (getenv("GHN3_ATTN_FWD_WAVES")) g_attn_fwd_waves = atoi(getenv("GHN3_ATTN_FWD_WAVES")) == 4 ? 4 : 8;
    if (getenv("GHN3_ATTN_BWD_WAVES")) g_attn_bwd_waves = atoi(getenv("GHN3_ATTN_BWD_WAVES")) == 4 ? 4 : 8;
    if (getenv("GHN3_ATTN_BWD_STAGED")) g_attn_bwd_staged = atoi(getenv("GHN3_ATTN_BWD_STAGED")) != 0;
    return GHN3_OK;
}

static int check_dims(int N, int C, int H) {
    if (H <= 0 || C % H != 0 || C / H > ATT_DMAX) {
        ghn3_set_error("attention: head dim %d/%d unsupported (max %d)", C, H, ATT_DMAX);
        return GHN3_E_LIMIT;
    }
    if (N > ATT_NMAX || N <= 0) { ghn3_set_error("attention: N=%d outside [1,%d]", N, ATT_NMAX); return GHN3_E_LIMIT; }
    return GHN3_OK;
}
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int ghn3_attn_fwd(float* out, const float* qkv, const float* bias, float* P, const int* n_nodes, int B, int N, int C,
                  int H, hipStream_t s) {
    int rc = check_dims(N, C, H);
    if (rc) return rc;
    const int d = C / H;
    const float scale = 1.0f / sqrtf((float)d);
    const int nb = (N + 31) / 32, tpw = (nb + 3) / 4;
    const int vec = (C % 4 == 0) && aligned16(out) && aligned16(qkv) && aligned16(bias) && aligned16(P);
    // TPW <= 2 variants stage Q (32 rows) + K + V (tpw * 128 rows each) head slices in LDS
    const size_t lds = tpw <= 2 ? (size_t)(32 + 2 * 2 * 128) * (((d + 3) & ~3) + 4) * sizeof(float) : 0;
    attn_fwd_fn fn = pick_fwd(d, tpw);
    if (lds > 32 * 1024) {
        // (once per kernel function and size: see ghn3_attn_bwd)
        static const void* done_fn[16];
        static size_t done_lds[16];
        static int n_done = 0;
        bool seen = false;
        for (int i = 0; i < n_done; ++i) seen = seen || (done_fn[i] == (const void*)fn && done_lds[i] >= lds);
        if (!seen) {
            hipError_t ea = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (ea != hipSuccess) { ghn3_set_error("attn fwd: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(ea)); return GHN3_E_HIP; }
            if (n_done < 16) { done_fn[n_done] = (const void*)fn; done_lds[n_done] = lds; ++n_done; }
        }
    }
    // (the staged variants own 16 queries per block)
    hipLaunchKernelGGL(fn, dim3(tpw <= 2 ? (N + 15) / 16 : nb, H, B), dim3(tpw <= 2 ? 64 * g_attn_fwd_waves : 256), lds, s, out,
                       qkv, bias, P, n_nodes, N, C, H, scale, vec);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ghn3_set_error("attn fwd launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    return GHN3_OK;
}

int ghn3_attn_bwd(float* dqkv, const float* dO, const float* qkv, const float* P, const float* O, float* amax_out,
                  float* dBias, const int* n_nodes, int B, int N, int C, int H, int general, hipStream_t s) {
    if (amax_out && !dBias) { ghn3_set_error("attention bwd: r5 (max |dBias|) needs r6"); return GHN3_E_ARG; }
    int rc = check_dims(N, C, H);
    if (rc) return rc;
    if (!P || !O) { ghn3_set_error("attention bwd: needs the saved probabilities and outputs"); return GHN3_E_ARG; }
    const int d = C / H;
    const float scale = 1.0f / sqrtf((float)d);
    const int nb = (N + 31) / 32;
    const int vec = (C % 4 == 0) && aligned16(dqkv) && aligned16(dO) && aligned16(qkv) && aligned16(P) &&
                    aligned16(O) && aligned16(dBias);
    int nw = 4;
    attn_bwd_fn fn = pick_bwd(d, N, &nw);
    size_t lds = 0;
    if (nw == 8 && vec && (d & 3) == 0 && g_attn_bwd_staged && !general) {
        // operands staged in LDS: exchange buffers + (32 + 3 * 32 * nb) rows of d + 4 floats (123 KB at d = 24, N = 256)
        fn = d <= 4 ? attn_bwd_staged_kernel<2> : d <= 8 ? attn_bwd_staged_kernel<4> : d <= 16 ? attn_bwd_staged_kernel<8>
             : d <= 24 ? attn_bwd_staged_kernel<12> : attn_bwd_staged_kernel<16>;
        lds = (size_t)(8 * 16 * 64 + 8 * 32 + (32 + 3 * 32 * nb) * (d + 4)) * sizeof(float);
        // (raised once per instantiation to the most any launch needs -- 8 row tiles at d + 4 = 36 floats: the call costs
        // host time on every launch of a chain that is host-bound when a new architecture arrives each step)
        static bool raised[5] = {false, false, false, false, false};
        const int slot = d <= 4 ? 0 : d <= 8 ? 1 : d <= 16 ? 2 : d <= 24 ? 3 : 4;
        if (!raised[slot]) {
            const size_t most = (size_t)(8 * 16 * 64 + 8 * 32 + (32 + 3 * 256) * 36) * sizeof(float);
            hipError_t ea = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)most);
            if (ea != hipSuccess) { ghn3_set_error("attn bwd: hipFuncSetAttribute(%zu): %s", most, hipGetErrorString(ea)); return GHN3_E_HIP; }
            raised[slot] = true;
        }
    }
    hipLaunchKernelGGL(fn, dim3(2 * nb, H, B), dim3(64 * nw), lds, s, dqkv, dO, qkv, P, O, dBias, n_nodes, N, C, H,
                       scale, vec, amax_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ghn3_set_error("attn bwd launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    return GHN3_OK;
}
