// Grouped MFMA GEMM for gfx950 (MI355X).  See include/ghn3_hip.h for the operand / epilogue contract.
//
// Replaces every F.linear / nn.Linear on the GHN-3 path (ghn3/graphormer.py:38-44,97-99,121,141;
// ghn3/nn.py:283-299,738-758) and their autograd backward (dgrad / wgrad), including the row-subset
// decoder GEMMs that only touch the W2 / Wfc rows a parameter group consumes.
//
// Structure (both compute types): 256 threads = 4 waves (2 x 2), block tile BM x BN, each wave owns
// (BM/2) x (BN/2) as a grid of 32 x 32 MFMA tiles accumulated in fp32.  Operands stay fp32 in HBM; tiles
// are staged global -> registers -> LDS (double buffered, one barrier per K-tile) so that the gather /
// strided addressing and the optional fp32 -> f16/bf16 conversion happen in the staging pass.
//   * F32 : v_mfma_f32_32x32x2_f32, exact fp32 (157 TF peak)            -- parity mode, BK = 32
//   * F16 / BF16 : v_mfma_f32_32x32x16_{f16,bf16} (2.5 PF peak)         -- throughput mode, BK = 64
// LDS images: ROW-mode operand  [rows][BK + pad]  (pad 1 float / 8 halfs -> conflict-free fragment reads)
//             COL-mode operand  f32: [BK][rows + 4] ; 16-bit: register-transposed into [rows][BK + 8].

#include "ghn3_internal.h"

// Pointers read from the problem table are generic ("flat") to the compiler; flat loads tick BOTH vmcnt and
// lgkmcnt, so every LDS wait would also drain the global prefetch.  All global accesses therefore go through
// explicit address-space-1 pointers (global_load / global_store).
#define GAS __attribute__((address_space(1)))
typedef const float GAS* gcf;
typedef float GAS* gf;
typedef const int GAS* gci;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 GAS* gcf4;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

#define ROWM GHN3_MODE_ROW
#define COLM GHN3_MODE_COL

__device__ __forceinline__ int map_row(int r, gci gather, int q, int s) {
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

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

__device__ __forceinline__ const GemmProbDev* find_problem(const GemmProbDev* probs, int n_probs, int bid) {
    int lo = 0, hi = n_probs - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (probs[mid].tile_start <= bid) lo = mid; else hi = mid - 1;
    }
    return probs + lo;
}

// Tile order.  Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, each with a private
// L2).  All m-tiles of one n-tile read the same B rows (the streamed weight rows), so they are given block
// ids that are congruent mod 8: n_tile = ((t / 8) / tiles_m) * 8 + t % 8, m_tile = (t / 8) % tiles_m.
// tile_start of every problem is a multiple of 8 and the inner tile count is padded to a multiple of 8 (surplus
// blocks exit), so the streamed panel is fetched into one L2 instead of up to eight.  With split-K
// (ksplit > 1) the tile grid is replicated per K chunk and partial sums are added to C atomically.
template <int BM, int BN>
__device__ __forceinline__ bool tile_origin(const GemmProbDev* P, int t, int& m0, int& n0, int& kz) {
    // order 0: B is the streamed operand (M <= N side small): m-tiles of one n-tile share an XCD;
    // order 1: A is the streamed operand: n-tiles of one m-tile share an XCD.
    const int tiles_m = P->tiles_m, tiles_n = P->tiles_n;
    const int per_split = P->order ? ((tiles_m + 7) / 8 * 8) * tiles_n : tiles_m * ((tiles_n + 7) / 8 * 8);
    kz = t / per_split;
    t -= kz * per_split;
    const int grp = t >> 3;
    int mt, nt;
    if (P->order) { mt = (grp / tiles_n) * 8 + (t & 7); nt = grp % tiles_n; }
    else { nt = (grp / tiles_m) * 8 + (t & 7); mt = grp % tiles_m; }
    m0 = mt * BM;
    n0 = nt * BN;
    return m0 < P->M && n0 < P->N;
}

// Epilogue shared by all variants.  acc tile layout (32x32 MFMA C/D): col = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// The element math is branch free for everything but GELU (uniform selects instead of per-element scalar
// branches: the unrolled branchy version was tens of KB of code per kernel and instruction-fetch bound); the GELU
// variant (erff) is a separate instantiation entered through ONE uniform branch.
template <int TM, int TN, bool GELU>
__device__ __forceinline__ void epilogue_impl(const GemmProbDev* P, f32x16 (&acc)[TM][TN], int m_base, int n_base,
                                              int lane) {
    const int M = P->M, N = P->N, ldc = P->ldc;
    gf C = (gf)P->C;
    gcf bias = (gcf)P->bias;
    gcf residual = (gcf)P->residual;
    gcf aux_in = (gcf)P->aux_in;
    gf aux_out = (gf)P->aux_out;
    gci cg = (gci)P->c_gather;
    const int cq = P->c_q, cs = P->c_s;
    const int act = P->act, dact = P->dact;
    const bool accum = (P->flags & GHN3_GEMM_ACCUM) != 0;
    const bool split = P->ksplit > 1;
    const float alpha = P->alpha;
    const bool act_relu = act == GHN3_ACT_RELU, dact_relu = dact == GHN3_DACT_RELU;
    const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = n_base + tn * 32 + l31;
        const bool col_ok = col < N;
        float bv = 0.f;
        if (bias && col_ok && !(P->flags & GHN3_GEMM_BIASGRAD)) {
            int bi = col;
            if (P->bias_q > 0) bi = (col / P->bias_q) * P->bias_s + (col % P->bias_q);
            bv = bias[(int64_t)bi * P->bias_stride];
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            // Pass 1: addresses + every read of this 32x32 sub-tile (old C, aux_in, residual) issued back to
            // back, so the 16 elements cost one memory round trip instead of 16 dependent ones.
            int64_t ci[16];
            bool ok[16];
            float cold[16], auxv[16], resv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m_base + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                ok[r] = row < M && col_ok;
                ci[r] = (int64_t)map_row(ok[r] ? row : 0, cg, cq, cs) * ldc + (col_ok ? col : 0);
            }
            if (split) {      // partial sum of one K chunk: C was zeroed by the host program
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (ok[r])
                        __hip_atomic_fetch_add(C + ci[r], acc[tm][tn][r] * alpha, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                cold[r] = (accum && ok[r]) ? C[ci[r]] : 0.f;
                auxv[r] = (dact != GHN3_DACT_NONE && ok[r]) ? aux_in[ci[r]] : 0.f;
                resv[r] = (residual && ok[r]) ? residual[ci[r]] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (ok[r]) {
                    float v = acc[tm][tn][r] * alpha + bv;
                    if (aux_out) aux_out[ci[r]] = v;
                    if (GELU) {
                        if (act == GHN3_ACT_RELU) v = fmaxf(v, 0.f);
                        else if (act == GHN3_ACT_GELU) v = gelu_f(v);
                        if (dact == GHN3_DACT_RELU) v = (auxv[r] > 0.f) ? v : 0.f;
                        else if (dact == GHN3_DACT_GELU) v *= gelu_grad_f(auxv[r]);
                    } else {
                        v = act_relu ? fmaxf(v, 0.f) : v;
                        v = (dact_relu && !(auxv[r] > 0.f)) ? 0.f : v;
                    }
                    v += resv[r] + cold[r];
                    C[ci[r]] = v;
                }
            }
        }
    }
}

template <int TM, int TN>
__device__ __forceinline__ void epilogue(const GemmProbDev* P, f32x16 (&acc)[TM][TN], int m_base, int n_base, int lane) {
    if (P->act == GHN3_ACT_GELU || P->dact == GHN3_DACT_GELU) epilogue_impl<TM, TN, true>(P, acc, m_base, n_base, lane);
    else epilogue_impl<TM, TN, false>(P, acc, m_base, n_base, lane);
}

// Row-vector epilogue: the accumulators of a wave's (TM*32) x 64 sub-tile are transposed through a wave-private LDS
// region (32 x 68 floats) so that every lane owns 4 CONSECUTIVE columns of a row: all reads (old C, aux_in,
// residual) and writes (C, aux_out) become 16-byte accesses of 256-byte row segments instead of one dword per lane
// (dword stores are store-issue bound: the 64 KB output of a 128 x 128 tile cost more than its whole K loop).
// Same epilogue semantics as epilogue<>; needs ldc % 4 == 0 and 16-byte aligned C / aux / residual (the caller
// checks and falls back).  `stage` = this wave's LDS region of 32 * 68 floats.
typedef f32x4 GAS* gf4;
struct RowsEpi {
    int M, N, ldc, cq, cs, ncol, col, c4, rsub, l31, lhi;
    bool act_relu, dact_relu;
    gf C; gcf residual, aux_in; gf aux_out; gci cg;
    bool accum; float alpha; f32x4 bv;
    float* stage;
};
// one 32 x 64 row block (accumulators a[0..1]) whose first row is `row0`; JB = row groups per read batch
template <int JB>
__device__ __forceinline__ void epilogue_rows_block(const RowsEpi& E, const f32x16 (&a)[2], int row0) {
    constexpr int LD = 68;
    float* stage = E.stage;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            stage[((r & 3) + 8 * (r >> 2) + 4 * E.lhi) * LD + tn * 32 + E.l31] = a[tn][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const bool full = E.ncol == 4;
    // batches of JB row groups: every read of a batch is issued before its first use (one memory round trip per
    // batch) while the live registers stay at 4 x JB float4 next to the accumulators
#pragma unroll
    for (int jb = 0; jb < 8; jb += JB) {
        f32x4 v[JB], cold[JB], auxv[JB], resv[JB];
        int64_t ci[JB];
        bool ok[JB];
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const int rl = E.rsub + 4 * (jb + j);
            v[j] = *reinterpret_cast<const f32x4*>(stage + rl * LD + E.c4);
            const int row = row0 + rl;
            ok[j] = row < E.M && E.ncol > 0;
            ci[j] = (int64_t)map_row(ok[j] ? row : 0, E.cg, E.cq, E.cs) * E.ldc + (E.ncol > 0 ? E.col : 0);
        }
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            cold[j] = z; auxv[j] = z; resv[j] = z;
            if (ok[j] && full) {
                if (E.accum) cold[j] = *reinterpret_cast<gcf4>(E.C + ci[j]);
                if (E.dact_relu) auxv[j] = *reinterpret_cast<gcf4>(E.aux_in + ci[j]);
                if (E.residual) resv[j] = *reinterpret_cast<gcf4>(E.residual + ci[j]);
            } else if (ok[j]) {
                for (int e = 0; e < E.ncol; ++e) {
                    if (E.accum) cold[j][e] = E.C[ci[j] + e];
                    if (E.dact_relu) auxv[j][e] = E.aux_in[ci[j] + e];
                    if (E.residual) resv[j][e] = E.residual[ci[j] + e];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            if (!ok[j]) continue;
            f32x4 pre, o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = v[j][e] * E.alpha + E.bv[e];
                pre[e] = x;
                x = E.act_relu ? fmaxf(x, 0.f) : x;
                x = (E.dact_relu && !(auxv[j][e] > 0.f)) ? 0.f : x;
                o[e] = x + resv[j][e] + cold[j][e];
            }
            if (full) {
                if (E.aux_out) *reinterpret_cast<gf4>(E.aux_out + ci[j]) = pre;
                *reinterpret_cast<gf4>(E.C + ci[j]) = o;
            } else {
                for (int e = 0; e < E.ncol; ++e) {
                    if (E.aux_out) E.aux_out[ci[j] + e] = pre[e];
                    E.C[ci[j] + e] = o[e];
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();                           // the next row block overwrites the stage
}

template <int TM>
__device__ __forceinline__ void epilogue_rows(const GemmProbDev* P, f32x16 (&acc)[TM][2], int m_base, int n_base,
                                              int lane, float* stage) {
    RowsEpi E;
    E.M = P->M; E.N = P->N; E.ldc = P->ldc; E.cq = P->c_q; E.cs = P->c_s;
    E.act_relu = P->act == GHN3_ACT_RELU; E.dact_relu = P->dact == GHN3_DACT_RELU;      // (GELU: rejected by the host)
    E.C = (gf)P->C; E.residual = (gcf)P->residual; E.aux_in = (gcf)P->aux_in; E.aux_out = (gf)P->aux_out;
    E.cg = (gci)P->c_gather;
    E.accum = (P->flags & GHN3_GEMM_ACCUM) != 0;
    E.alpha = P->alpha_amax ? P->alpha * ghn3_pow2_inv_scale(*P->alpha_amax) : P->alpha;
    E.l31 = lane & 31; E.lhi = lane >> 5;
    E.c4 = (lane & 15) * 4; E.rsub = lane >> 4;               // this lane's 4 columns / row inside a 4-row group
    E.col = n_base + E.c4;
    E.ncol = min(4, E.N - E.col);                              // valid columns of this lane (<= 0: none)
    E.stage = stage;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    E.bv = z;
    gcf bias = (gcf)P->bias;
    if (bias && !(P->flags & GHN3_GEMM_BIASGRAD)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e < E.ncol) {
                int bi = E.col + e;
                if (P->bias_q > 0) bi = (bi / P->bias_q) * P->bias_s + (bi % P->bias_q);
                E.bv[e] = bias[(int64_t)bi * P->bias_stride];
            }
        }
    }
    // static indices only: a dynamically indexed accumulator array would live in scratch memory
    constexpr int JB = TM > 2 ? 2 : 4;
    if constexpr (TM >= 1) epilogue_rows_block<JB>(E, acc[0], m_base);
    if constexpr (TM >= 2) epilogue_rows_block<JB>(E, acc[1], m_base + 32);
    if constexpr (TM >= 3) epilogue_rows_block<JB>(E, acc[2], m_base + 64);
    if constexpr (TM >= 4) epilogue_rows_block<JB>(E, acc[3], m_base + 96);
}

// split-K partial sums: plain dword atomics straight from the accumulator layout (32 lanes on one 128-byte line)
__device__ __forceinline__ void epilogue_split_block(const GemmProbDev* P, const f32x16& a, int row0, int col, int lhi) {
    if (col >= P->N) return;
    gf C = (gf)P->C;
    const float alpha = P->alpha_amax ? P->alpha * ghn3_pow2_inv_scale(*P->alpha_amax) : P->alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < P->M)
            __hip_atomic_fetch_add(C + (int64_t)map_row(row, (gci)P->c_gather, P->c_q, P->c_s) * P->ldc + col,
                                   a[r] * alpha, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int TM>
__device__ __forceinline__ void epilogue_split(const GemmProbDev* P, f32x16 (&acc)[TM][2], int m_base, int n_base,
                                               int lane) {
    const int l31 = lane & 31, lhi = lane >> 5;
#define GHN3_SPLIT_ROW(I)                                                                   \
    if constexpr (TM > I) {                                                                 \
        epilogue_split_block(P, acc[I][0], m_base + 32 * I, n_base + l31, lhi);             \
        epilogue_split_block(P, acc[I][1], m_base + 32 * I, n_base + 32 + l31, lhi);        \
    }
    GHN3_SPLIT_ROW(0) GHN3_SPLIT_ROW(1) GHN3_SPLIT_ROW(2) GHN3_SPLIT_ROW(3)
#undef GHN3_SPLIT_ROW
}
__device__ __forceinline__ bool rows_epilogue_ok(const GemmProbDev* P) {
    const uintptr_t bits = (uintptr_t)P->C | (uintptr_t)P->aux_in | (uintptr_t)P->aux_out | (uintptr_t)P->residual;
    return (P->ldc & 3) == 0 && (bits & 15) == 0;
}

// ------------------------------------------------------------------------------------------------
// fp32 operands: v_mfma_f32_32x32x2_f32
// ------------------------------------------------------------------------------------------------
template <int ROWS, int MODE> struct F32Tile {
    static constexpr int BK = 32;
    static constexpr int LD = (MODE == ROWM) ? (BK + 1) : (ROWS + 4);
    static constexpr int SIZE = (MODE == ROWM) ? ROWS * (BK + 1) : BK * (ROWS + 4);
    static constexpr int NV = ROWS / 32;                 // float4 per thread per K-tile
    static constexpr int F4_PER_ROW = ROWS / 4;          // COL mode
    static constexpr int RPP = 256 / F4_PER_ROW;         // COL mode: k rows per pass
};

template <int ROWS, int MODE>
struct F32Loader {
    using T = F32Tile<ROWS, MODE>;
    gcf base; gci gather; int q, s, ld;
    int lim_rows, lim_k;          // logical extents (rows = M or N ; k = K)
    int r0;                       // tile origin along rows
    // ROW mode state
    gcf ptr[T::NV]; bool ok[T::NV]; int kc, rr;
    // COL mode state
    int mc, kr; bool ok_m;

    __device__ __forceinline__ void init(const float* b, const int* g, int q_, int s_, int ld_, int rows, int K,
                                         int origin, int tid) {
        base = (gcf)b; gather = (gci)g; q = q_; s = s_; ld = ld_; lim_rows = rows; lim_k = K; r0 = origin;
        if (MODE == ROWM) {
            kc = tid & 7; rr = tid >> 3;
#pragma unroll
            for (int i = 0; i < T::NV; ++i) {
                int m = r0 + rr + 32 * i;
                ok[i] = m < lim_rows;
                int r = map_row(ok[i] ? m : 0, gather, q, s);
                ptr[i] = base + (int64_t)r * ld + kc * 4;
            }
        } else {
            mc = tid % T::F4_PER_ROW; kr = tid / T::F4_PER_ROW;
            ok_m = (r0 + mc * 4) < lim_rows;
        }
    }
    // load() only ISSUES the global loads (no use of the results: the tail masking that would force an
    // s_waitcnt right behind every load is applied in store(), after the MFMA loop of the current tile).
    __device__ __forceinline__ void load(int kt, f32x4 (&v)[T::NV]) const {
        if (MODE == ROWM) {
            const int k = kt * T::BK + kc * 4;
#pragma unroll
            for (int i = 0; i < T::NV; ++i) {
                f32x4 x = {0.f, 0.f, 0.f, 0.f};
                if (ok[i] && k < lim_k) x = *reinterpret_cast<gcf4>(ptr[i] + kt * T::BK);
                v[i] = x;
            }
        } else {
            const int m = r0 + mc * 4;
#pragma unroll
            for (int i = 0; i < T::NV; ++i) {
                const int k = kt * T::BK + kr + T::RPP * i;
                f32x4 x = {0.f, 0.f, 0.f, 0.f};
                if (ok_m && k < lim_k) {
                    const int r = map_row(k, gather, q, s);
                    x = *reinterpret_cast<gcf4>(base + (int64_t)r * ld + m);
                }
                v[i] = x;
            }
        }
    }
    __device__ __forceinline__ void mask_tail(int kt, f32x4& x) const {
        // elements beyond the logical extent along the contiguous dimension read as zero
        const int c0 = (MODE == ROWM) ? kt * T::BK + kc * 4 : r0 + mc * 4;
        const int lim = (MODE == ROWM) ? lim_k : lim_rows;
        if (c0 + 3 >= lim) {
            if (c0 + 1 >= lim) x.y = 0.f;
            if (c0 + 2 >= lim) x.z = 0.f;
            if (c0 + 3 >= lim) x.w = 0.f;
        }
    }
    __device__ __forceinline__ void store(float* lds, f32x4 (&v)[T::NV], int kt) const {
        if (MODE == ROWM) {
#pragma unroll
            for (int i = 0; i < T::NV; ++i) {
                mask_tail(kt, v[i]);
                float* p = lds + (rr + 32 * i) * T::LD + kc * 4;
                p[0] = v[i].x; p[1] = v[i].y; p[2] = v[i].z; p[3] = v[i].w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < T::NV; ++i) {
                mask_tail(kt, v[i]);
                *reinterpret_cast<f32x4*>(lds + (kr + T::RPP * i) * T::LD + mc * 4) = v[i];
            }
        }
    }
};

template <int BM, int BN, int AM, int BMD>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmProbDev* __restrict__ probs, int n_probs) {
    using TA = F32Tile<BM, AM>;
    using TB = F32Tile<BN, BMD>;
    constexpr int BK = 32;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = TA::SIZE + TB::SIZE;

    const GemmProbDev* P = find_problem(probs, n_probs, blockIdx.x);
    const int t = blockIdx.x - P->tile_start;
    int m0, n0, kz;
    if (!tile_origin<BM, BN>(P, t, m0, n0, kz)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int kt0 = kz * (P->k_chunk / BK);
    const int K = (P->ksplit > 1) ? min(P->K, (kz + 1) * P->k_chunk) : P->K;
    if (kt0 * BK >= K && P->ksplit > 1) return;

    F32Loader<BM, AM> la; F32Loader<BN, BMD> lb;
    la.init(P->A, P->a_gather, P->a_q, P->a_s, P->lda, P->M, K, m0, tid);
    lb.init(P->B, P->b_gather, P->b_q, P->b_s, P->ldb, P->N, K, n0, tid);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (K + BK - 1) / BK;
    // fused bias gradient (wgrad problems): row sums of the COL-mode A tile, done by the n-tile-0 blocks
    const bool do_bg = (AM == COLM) && (P->flags & GHN3_GEMM_BIASGRAD) && n0 == 0;
    float bg = 0.f;
    f32x4 ra[TA::NV], rb[TB::NV];
    la.load(kt0, ra); lb.load(kt0, rb);
    la.store(smem, ra, kt0); lb.store(smem + TA::SIZE, rb, kt0);
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        const int cur = (kt - kt0) & 1;
        const bool more = (kt + 1) < nk;
        if (more) { la.load(kt + 1, ra); lb.load(kt + 1, rb); }
        const float* a_s = smem + cur * STAGE;
        const float* b_s = a_s + TA::SIZE;
        if (AM == COLM) {
            if (do_bg && tid < BM) {
#pragma unroll 8
                for (int kk = 0; kk < BK; ++kk) bg += a_s[kk * TA::LD + tid];
            }
        }
#pragma unroll 4
        for (int kk = 0; kk < BK; kk += 2) {
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = (AM == ROWM) ? a_s[(wm0 + i * 32 + l31) * TA::LD + kk + lhi]
                                     : a_s[(kk + lhi) * TA::LD + wm0 + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = (BMD == ROWM) ? b_s[(wn0 + j * 32 + l31) * TB::LD + kk + lhi]
                                      : b_s[(kk + lhi) * TB::LD + wn0 + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            float* nx = smem + (cur ^ 1) * STAGE;
            la.store(nx, ra, kt + 1); lb.store(nx + TA::SIZE, rb, kt + 1);
        }
        __syncthreads();
    }
    if (do_bg && tid < BM && m0 + tid < P->M) {
        gf dbias = (gf)P->bias;
        const int64_t bi = (int64_t)map_row(m0 + tid, (gci)P->c_gather, P->c_q, P->c_s) * P->bias_stride;
        if (P->ksplit > 1) __hip_atomic_fetch_add(dbias + bi, bg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else dbias[bi] += bg;
    }
    epilogue<TM, TN>(P, acc, m0 + wm0, n0 + wn0, lane);
}

// ------------------------------------------------------------------------------------------------
// 16-bit operands (converted from fp32 while staging): v_mfma_f32_32x32x16_{f16,bf16}
// LDS image for both modes: [rows][BK + 8] 16-bit elements (BK = 64).
// Fragment: lane l holds 8 consecutive k at k = kk + (l >> 5) * 8 for row (l & 31).
// ------------------------------------------------------------------------------------------------
template <int CT> __device__ __forceinline__ unsigned short cvt16(float x);
template <> __device__ __forceinline__ unsigned short cvt16<GHN3_CT_F16>(float x) {
    _Float16 h = (_Float16)x;
    return __builtin_bit_cast(unsigned short, h);
}
template <> __device__ __forceinline__ unsigned short cvt16<GHN3_CT_BF16>(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7fffu + ((u >> 16) & 1u);          // round to nearest even (inputs are finite)
    return (unsigned short)(u >> 16);
}

template <int CT> __device__ __forceinline__ float cvt_back(unsigned short h) {
    if (CT == GHN3_CT_F16) return (float)__builtin_bit_cast(_Float16, h);
    return __builtin_bit_cast(float, ((unsigned)h) << 16);
}

template <int ROWS, int MODE, int CT>
struct H16Loader {
    static constexpr int BK = 64;
    static constexpr int LD = BK + 8;                     // in 16-bit elements (144 B rows)
    static constexpr int SIZE = ROWS * LD;                // 16-bit elements
    // ROW: thread -> (row = tid/8 + 32 i, 8 k at kc*8), i < ROWS/32, 2 float4 each
    // COL: thread -> micro tile 4 rows x 8 k: kg = tid & 7, mg = tid >> 3 (mg < ROWS/4), 8 float4
    static constexpr int NV = (MODE == ROWM) ? (ROWS / 32) * 2 : 8;
    gcf base; gci gather; int q, s, ld, lim_rows, lim_k, r0;
    gcf ptr[ROWS / 32]; bool ok[ROWS / 32]; int kc, rr;
    int kg, mg; bool ok_m, active;

    __device__ __forceinline__ void init(const float* b, const int* g, int q_, int s_, int ld_, int rows, int K,
                                         int origin, int tid) {
        base = (gcf)b; gather = (gci)g; q = q_; s = s_; ld = ld_; lim_rows = rows; lim_k = K; r0 = origin;
        if (MODE == ROWM) {
            kc = tid & 7; rr = tid >> 3;
#pragma unroll
            for (int i = 0; i < ROWS / 32; ++i) {
                int m = r0 + rr + 32 * i;
                ok[i] = m < lim_rows;
                int r = map_row(ok[i] ? m : 0, gather, q, s);
                ptr[i] = base + (int64_t)r * ld + kc * 8;
            }
        } else {
            kg = tid & 7; mg = tid >> 3;
            active = mg < ROWS / 4;
            ok_m = active && (r0 + mg * 4) < lim_rows;
        }
    }
    __device__ __forceinline__ void load(int kt, f32x4 (&v)[NV]) const {
        if (MODE == ROWM) {
            const int k = kt * BK + kc * 8;
#pragma unroll
            for (int i = 0; i < ROWS / 32; ++i) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
                    if (ok[i] && k + 4 * h < lim_k) x = *reinterpret_cast<gcf4>(ptr[i] + kt * BK + 4 * h);
                    v[i * 2 + h] = x;
                }
            }
        } else {
            const int m = r0 + mg * 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = kt * BK + kg * 8 + j;
                f32x4 x = {0.f, 0.f, 0.f, 0.f};
                if (ok_m && k < lim_k) {
                    const int r = map_row(k, gather, q, s);
                    x = *reinterpret_cast<gcf4>(base + (int64_t)r * ld + m);
                }
                v[j] = x;
            }
        }
    }
    __device__ __forceinline__ void mask_tail(int c0, int lim, f32x4& x) const {
        if (c0 + 3 >= lim) {
            if (c0 + 1 >= lim) x.y = 0.f;
            if (c0 + 2 >= lim) x.z = 0.f;
            if (c0 + 3 >= lim) x.w = 0.f;
        }
    }
    __device__ __forceinline__ void store(unsigned short* lds, f32x4 (&v)[NV], int kt) const {
        if (MODE == ROWM) {
#pragma unroll
            for (int i = 0; i < ROWS / 32; ++i) {
                mask_tail(kt * BK + kc * 8, lim_k, v[2 * i]);
                mask_tail(kt * BK + kc * 8 + 4, lim_k, v[2 * i + 1]);
                u16x8 h;
                h[0] = cvt16<CT>(v[2 * i].x); h[1] = cvt16<CT>(v[2 * i].y);
                h[2] = cvt16<CT>(v[2 * i].z); h[3] = cvt16<CT>(v[2 * i].w);
                h[4] = cvt16<CT>(v[2 * i + 1].x); h[5] = cvt16<CT>(v[2 * i + 1].y);
                h[6] = cvt16<CT>(v[2 * i + 1].z); h[7] = cvt16<CT>(v[2 * i + 1].w);
                *reinterpret_cast<u16x8*>(lds + (rr + 32 * i) * LD + kc * 8) = h;
            }
        } else if (active) {
            u16x8 h0, h1, h2, h3;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                mask_tail(r0 + mg * 4, lim_rows, v[j]);
                h0[j] = cvt16<CT>(v[j].x); h1[j] = cvt16<CT>(v[j].y);
                h2[j] = cvt16<CT>(v[j].z); h3[j] = cvt16<CT>(v[j].w);
            }
            unsigned short* p = lds + (mg * 4) * LD + kg * 8;
            *reinterpret_cast<u16x8*>(p) = h0;
            *reinterpret_cast<u16x8*>(p + LD) = h1;
            *reinterpret_cast<u16x8*>(p + 2 * LD) = h2;
            *reinterpret_cast<u16x8*>(p + 3 * LD) = h3;
        }
    }
};

template <int CT>
__device__ __forceinline__ f32x16 mfma16(u16x8 a, u16x8 b, f32x16 c) {
    if (CT == GHN3_CT_F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int BM, int BN, int AM, int BMD, int CT>
__global__ __launch_bounds__(256) void gemm_h16_kernel(const GemmProbDev* __restrict__ probs, int n_probs) {
    using LA = H16Loader<BM, AM, CT>;
    using LB = H16Loader<BN, BMD, CT>;
    constexpr int BK = 64, LD = LA::LD;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* sm16 = reinterpret_cast<unsigned short*>(smem);
    constexpr int STAGE = LA::SIZE + LB::SIZE;

    const GemmProbDev* P = find_problem(probs, n_probs, blockIdx.x);
    const int t = blockIdx.x - P->tile_start;
    int m0, n0, kz;
    if (!tile_origin<BM, BN>(P, t, m0, n0, kz)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int kt0 = kz * (P->k_chunk / BK);
    const int K = (P->ksplit > 1) ? min(P->K, (kz + 1) * P->k_chunk) : P->K;
    if (kt0 * BK >= K && P->ksplit > 1) return;

    LA la; LB lb;
    la.init(P->A, P->a_gather, P->a_q, P->a_s, P->lda, P->M, K, m0, tid);
    lb.init(P->B, P->b_gather, P->b_q, P->b_s, P->ldb, P->N, K, n0, tid);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (K + BK - 1) / BK;
    const bool do_bg = (AM == COLM) && (P->flags & GHN3_GEMM_BIASGRAD) && n0 == 0;
    float bg = 0.f;
    f32x4 ra[LA::NV], rb[LB::NV];
    la.load(kt0, ra); lb.load(kt0, rb);
    la.store(sm16, ra, kt0); lb.store(sm16 + LA::SIZE, rb, kt0);
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        const int cur = (kt - kt0) & 1;
        const bool more = (kt + 1) < nk;
        if (more) { la.load(kt + 1, ra); lb.load(kt + 1, rb); }
        const unsigned short* a_s = sm16 + cur * STAGE;
        const unsigned short* b_s = a_s + LA::SIZE;
        if (AM == COLM) {
            if (do_bg && tid < BM) {
#pragma unroll
                for (int kk = 0; kk < BK; kk += 8) {
                    const u16x8 h = *reinterpret_cast<const u16x8*>(a_s + tid * LD + kk);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bg += cvt_back<CT>(h[e]);
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            u16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const u16x8*>(a_s + (wm0 + i * 32 + l31) * LD + kk + lhi * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const u16x8*>(b_s + (wn0 + j * 32 + l31) * LD + kk + lhi * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = mfma16<CT>(af[i], bf[j], acc[i][j]);
        }
        if (more) {
            unsigned short* nx = sm16 + (cur ^ 1) * STAGE;
            la.store(nx, ra, kt + 1); lb.store(nx + LA::SIZE, rb, kt + 1);
        }
        __syncthreads();
    }
    if (do_bg && tid < BM && m0 + tid < P->M) {
        gf dbias = (gf)P->bias;
        const int64_t bi = (int64_t)map_row(m0 + tid, (gci)P->c_gather, P->c_q, P->c_s) * P->bias_stride;
        if (P->ksplit > 1) __hip_atomic_fetch_add(dbias + bi, bg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else dbias[bi] += bg;
    }
    epilogue<TM, TN>(P, acc, m0 + wm0, n0 + wn0, lane);
}

// ------------------------------------------------------------------------------------------------
// 16-bit operands ALREADY in HBM (GHN3_GEMM_OP16): both operands k-contiguous (ROW mode), staged with the
// LDS-DMA path (global_load_lds_dwordx4: 16 B per lane, no VGPR round trip, no conversion in the loop).
//   block tile 128 x 128 x 64, 4 waves (2 x 2) of 64 x 64, v_mfma_f32_32x32x16_{f16,bf16}, fp32 accumulate.
//   LDS image per operand and stage: 128 rows x 128 B, written lane-linearly by the DMA; the 16-byte slot s of
//   row r holds global k-chunk (s ^ (r & 7)) -- the swizzle is applied to the per-lane SOURCE address and to the
//   fragment reads (rule 21 of the CDNA guide), which makes the ds_read_b128 fragment reads conflict free.
//   Rows beyond M/N are clamped to the last valid row (their results are never stored); the K range must be
//   zero padded to a multiple of 64 in the 16-bit copies (the host allocates them that way).
//   The B operand may carry a k-map (kq, ks): physical k = (k / kq) * ks + k % kq with kq % 64 == 0 -- the
//   row-subset structure of the decoder W2 weights seen from the reduction side (dgrad).
// ------------------------------------------------------------------------------------------------
#define LAS __attribute__((address_space(3)))
typedef const unsigned short GAS* gch;

// WGM x WGN waves, each owning a (BM / WGM) x 64 sub-tile; NS LDS stages form a ring that keeps NS - 1 k-tiles in
// flight (counted s_waitcnt vmcnt + a raw s_barrier per k-tile: __syncthreads would drain the DMA queue).
// The loop is bound by the L2 -> LDS rate (~17 B/clk/CU measured), so the flop rate scales with the tile's
// arithmetic intensity BM BN / (BM + BN): 64 flop/B for 128 x 128, 128 flop/B for 256 x 256.
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int CT, int BM, int BN, int WGM, int WGN, int NS>
__device__ __forceinline__ void h16d_tile(const GemmProbDev* __restrict__ probs, int n_probs, int tile_id, float* smem) {
    constexpr int BK = 64;
    constexpr int NT = 64 * WGM * WGN;               // threads
    constexpr int TM = BM / WGM / 32, TN = 2;
    static_assert(BN / WGN == 64, "a wave owns 64 columns (epilogue_rows)");
    constexpr int OPA = BM * BK * 2, OPB = BN * BK * 2;   // bytes per operand image
    constexpr int STAGE = OPA + OPB;
    constexpr int PA = BM * 8 / NT, PB = BN * 8 / NT;     // 16-byte pieces per thread
    static_assert((NS - 2) * (PA + PB) < 64, "vmcnt is a 6-bit counter");
    char* sm = reinterpret_cast<char*>(smem);

    const GemmProbDev* P;
    int m0, n0, kz;
    const int pin_end = probs[0].pin_end;
    if (tile_id < pin_end) {
        // XCD-pinned problems (ghn3_gemm_problem::xcd_pin): id -> (XCD, local index); the problems of an XCD are found
        // through the directory in entry x of the array; their tiles run m fastest (the row tiles of a column tile share
        // its B rows, consecutive column tiles the A rows of the chunk)
        const int x = tile_id & 7, local = tile_id >> 3;
        const int cnt = probs[x].pin_count;
        if (cnt == 0) return;
        int lo = probs[x].pin_first, hi = lo + cnt - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (probs[mid].tile_start <= local) lo = mid; else hi = mid - 1;
        }
        P = probs + lo;
        const int t = local - P->tile_start;
        if (t >= P->tiles_m * P->tiles_n) return;
        m0 = (t % P->tiles_m) * BM;
        n0 = (t / P->tiles_m) * BN;
        kz = 0;
    } else {
        const int n_pin = probs[0].pin_total;
        P = find_problem(probs + n_pin, n_probs - n_pin, tile_id);
        if (!tile_origin<BM, BN>(P, tile_id - P->tile_start, m0, n0, kz)) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WGN) * (BM / WGM), wn0 = (wave % WGN) * 64;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int kt0 = kz * (P->k_chunk / BK);
    int Kp = P->K;
    if (P->lim) {                                    // ragged extents of the stacked rows (one entry per 128 rows)
        const int* lim = P->lim;
        int ext = lim[m0 >> 7];
        if (BM > 128 && m0 + 128 < P->M) ext = max(ext, lim[(m0 >> 7) + 1]);
        if (P->lim_kind == 1) { if (n0 >= ext) return; }
        else Kp = min(Kp, ext);
    }
    const int K = (P->ksplit > 1) ? min(Kp, (kz + 1) * P->k_chunk) : Kp;
    if (kt0 * BK >= K && P->ksplit > 1) return;
    const int nkt = (K + BK - 1) / BK - kt0;         // k-tiles of this block

    // per-lane source pointers: piece p = tid + NT i -> LDS row p / 8, slot p % 8, which holds the global 16-byte
    // chunk slot ^ ((row >> 1) & 7): with that swizzle the 16 lanes of every ds_read_b128 lane group hit 16
    // different 4-bank columns (MI355X_MICROARCH.md, LDS table)
    gch pa[PA], pb[PB];
    int ck;                                          // k offset of this lane's chunk: the same for all its pieces
    {
        gch A = (gch)P->A; gch B = (gch)P->B;
        gci ag = (gci)P->a_gather; gci bg = (gci)P->b_gather;
        const int slot = tid & 7, rbase = tid >> 3;  // NT / 8 rows per pass (a multiple of 16)
        ck = (slot ^ ((rbase >> 1) & 7)) * 8;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int ra = min(m0 + rbase + (NT / 8) * i, P->M - 1);
            pa[i] = A + (int64_t)map_row(ra, ag, P->a_q, P->a_s) * P->lda + ck;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int rb = min(n0 + rbase + (NT / 8) * i, P->N - 1);
            pb[i] = B + (int64_t)map_row(rb, bg, P->b_q, P->b_s) * P->ldb;
        }
    }
    const int kq = P->kq, ks = P->ks;
    const float inv_kq = kq > 0 ? 1.0f / (float)kq : 0.f;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto issue = [&](int i) {                          // k-tile i of this block -> stage i % NS
        const int k0 = (kt0 + i) * BK;
        int kb = k0 + ck;
        if (kq > 0) {                                  // k-map of B: (k / kq) * ks + k % kq   (k < 2^24)
            int qd = (int)((float)kb * inv_kq);
            int rm = kb - qd * kq;
            if (rm < 0) { rm += kq; --qd; } else if (rm >= kq) { rm -= kq; ++qd; }
            kb = qd * ks + rm;
        }
        LAS char* la = (LAS char*)(sm + (i % NS) * STAGE);
        // wave-uniform LDS base: the 64 lanes of a wave write 64 consecutive 16-byte pieces
#pragma unroll
        for (int j = 0; j < PA; ++j)
            __builtin_amdgcn_global_load_lds((const void GAS*)(pa[j] + k0), (LAS void*)(la + (wave * 64 + NT * j) * 16),
                                             16, 0, 0);
#pragma unroll
        for (int j = 0; j < PB; ++j)
            __builtin_amdgcn_global_load_lds((const void GAS*)(pb[j] + kb),
                                             (LAS void*)(la + OPA + (wave * 64 + NT * j) * 16), 16, 0, 0);
    };

#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
        if (i < nkt) issue(i);
    for (int i = 0; i < nkt; ++i) {
        // tile i has landed once at most the NS - 2 younger tiles are outstanding (fewer were issued near the end)
        if (i + NS - 1 <= nkt) wait_vmcnt<(NS - 2) * (PA + PB)>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                  // every wave's share of tile i landed; stage (i-1) % NS is free
        const char* a_s = sm + (i % NS) * STAGE;
        const char* b_s = a_s + OPA;
        // fragments of k-step kk + 1 are read from LDS while the MFMAs of step kk run (two register sets)
        u16x8 af[2][TM], bf[2][TN];
        auto load_frags = [&](int kk, int buf) {
            const int slot = kk * 2 + lhi;             // 16-byte k chunk wanted by this lane
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) {
                const int row = wm0 + ii * 32 + l31;
                af[buf][ii] = *reinterpret_cast<const u16x8*>(a_s + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn0 + j * 32 + l31;
                bf[buf][j] = *reinterpret_cast<const u16x8*>(b_s + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
            }
        };
        load_frags(0, 0);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {         // 4 MFMA k-steps of 16
            if (kk + 1 < BK / 16) load_frags(kk + 1, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);         // keep the reads of step kk + 1 ahead of the MFMAs of step kk
#pragma unroll
            for (int ii = 0; ii < TM; ++ii)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[ii][j] = mfma16<CT>(af[kk & 1][ii], bf[kk & 1][j], acc[ii][j]);
            __builtin_amdgcn_sched_barrier(0);
            // the DMA of the next tile is issued behind the first MFMA batch: its address arithmetic runs while the
            // matrix pipe is busy instead of in front of the first fragment reads
            if (kk == 0 && i + NS - 1 < nkt) { issue(i + NS - 1); __builtin_amdgcn_sched_barrier(0); }
        }
    }
    // (the host only routes problems with ldc % 4 == 0 and 16-byte aligned C / aux / residual to this kernel)
    if (P->ksplit > 1) {
        epilogue_split<TM>(P, acc, m0 + wm0, n0 + wn0, lane);
    } else {
        __syncthreads();                               // every wave is done with the operand stages
        epilogue_rows<TM>(P, acc, m0 + wm0, n0 + wn0, lane, smem + wave * (32 * 68));
    }
}

// One workgroup per tile, or (grid capped by the host) a persistent workgroup that strides over the tiles: a
// side-stream launch limited to a fraction of the CUs leaves the others to the latency-bound chain it runs beside.
template <int CT, int BM, int BN, int WGM, int WGN, int NS>
__global__ __launch_bounds__(64 * WGM * WGN, (NS * (BM + BN) * 128 <= 80 * 1024 ? 2 : 1) * WGM * WGN / 4)
void gemm_h16d_kernel(const GemmProbDev* __restrict__ probs, int n_probs, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        h16d_tile<CT, BM, BN, WGM, WGN, NS>(probs, n_probs, tile, smem);
        __syncthreads();                               // LDS (stages / epilogue staging) is reused by the next tile
    }
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 PERSISTENT variant for output-heavy problems (tile code 25): the decoder's W2 weight gradient is
// dW2 [o i x 8C] = d_tiles^T u with K = the family's rows (<= 768: twelve k-tiles) -- every tile computes for ~5 us and
// then writes 256 KB.  In the kernel above a workgroup owns the CU (128 KB of LDS), so nothing overlaps its prologue
// (first DMA round trip), its LDS-staged epilogue and its stores: the matrix cores idle half the time (measured 534 TF
// standalone against 864-921 TF for K = 3072).  Here
//   * a workgroup walks its tiles (tile += gridDim.x) and issues the FIRST k-tile of the next tile in the last
//     iteration of the current one -- the stage it lands in was released by the barrier of that iteration;
//   * the products are computed TRANSPOSED (operands swapped: D^T = B_tile A_tile^T), so that in the 32 x 32 C/D layout a
//     lane is an output ROW and its registers 4 g .. 4 g + 3 are 4 consecutive COLUMNS: the epilogue is 32 float4
//     stores per lane straight from the accumulators -- no LDS staging, no barrier, and it runs while the next tile's
//     first k-tile is in flight (the store acknowledgements and that DMA share one s_waitcnt vmcnt(0)).
// Plain problems only (the host checks): C = alpha * A B^T with an optional row map of C; no bias / activation /
// residual / accumulate, no gathers, no k-map, no ragged extents, no split-K.
// ------------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(512)
void gemm_h16w_kernel(const GemmProbDev* __restrict__ probs, int n_probs, int total_tiles) {
    constexpr int BM = 256, BN = 256, BK = 64, WGN = 4, NT = 512, TM = 4, TN = 2;
    constexpr int OPA = BM * BK * 2, OPB = BN * BK * 2, STAGE = OPA + OPB;
    constexpr int PA = BM * 8 / NT, PB = BN * 8 / NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sm = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WGN) * (BM / 2), wn0 = (wave % WGN) * 64;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int slot = tid & 7, rbase = tid >> 3;
    const int ck = (slot ^ ((rbase >> 1) & 7)) * 8;          // k offset of this lane's 16-byte chunk (swizzled source)

    gch pa[PA], pb[PB];
    const GemmProbDev* P = nullptr;
    int m0 = 0, n0 = 0, nkt = 0;
    // first valid tile at or behind `t` (surplus ids of the XCD-aware order are skipped); sets P, m0, n0, nkt, pa, pb
    auto setup = [&](int t) -> int {
        for (; t < total_tiles; t += gridDim.x) {
            const GemmProbDev* Q = find_problem(probs, n_probs, t);
            {
                // XCD-blocked order (workgroup b runs on XCD b % 8; tile ids keep their residue: gridDim.x % 8 == 0).
                // XCD x = (row class x / G, column group x % G): consecutive tiles of an XCD walk the column tiles of its
                // group for one row tile (they share that A tile while it streams), then the next row tile of its class;
                // the group's slice of B (<= 1.5 MB) stays in the XCD's L2.  With every XCD cycling through ALL of B
                // (4.7 MB for the W2 weight gradient, more than one L2) each tile re-fetched its B tile: 2.7 GB per step.
                const int tl = t - Q->tile_start, x = tl & 7, grp = tl >> 3;
                const int G = Q->xcd_cols, Gm = 8 / G;
                const int npg = (Q->tiles_n + G - 1) / G;
                const int nt = (x % G) * npg + grp % npg, mt = (grp / npg) * Gm + x / G;
                if (nt >= Q->tiles_n || mt >= Q->tiles_m) continue;
                m0 = mt * BM;
                n0 = nt * BN;
            }
            P = Q;
            nkt = (Q->K + BK - 1) / BK;
            gch A = (gch)Q->A; gch B = (gch)Q->B;
#pragma unroll
            for (int i = 0; i < PA; ++i)
                pa[i] = A + (int64_t)min(m0 + rbase + (NT / 8) * i, Q->M - 1) * Q->lda + ck;
#pragma unroll
            for (int i = 0; i < PB; ++i)
                pb[i] = B + (int64_t)min(n0 + rbase + (NT / 8) * i, Q->N - 1) * Q->ldb + ck;
            return t;
        }
        return total_tiles;
    };
    auto issue = [&](int kt, int stage) {                     // k-tile kt of the tile described by pa / pb
        LAS char* la = (LAS char*)(sm + stage * STAGE);
#pragma unroll
        for (int j = 0; j < PA; ++j)
            __builtin_amdgcn_global_load_lds((const void GAS*)(pa[j] + kt * BK), (LAS void*)(la + (wave * 64 + NT * j) * 16),
                                             16, 0, 0);
#pragma unroll
        for (int j = 0; j < PB; ++j)
            __builtin_amdgcn_global_load_lds((const void GAS*)(pb[j] + kt * BK),
                                             (LAS void*)(la + OPA + (wave * 64 + NT * j) * 16), 16, 0, 0);
    };

    int tile = setup(blockIdx.x);
    int st = 0;                                               // stage that holds k-tile 0 of the current tile
    if (tile < total_tiles) issue(0, 0);
    while (tile < total_tiles) {
        // what the epilogue needs of the current tile (pa / pb / P are re-used for the next one inside the loop)
        const GemmProbDev* Pc = P;
        const int cm0 = m0, cn0 = n0, cnkt = nkt;
        int next_tile = total_tiles;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int i = 0; i < cnkt; ++i) {
            wait_vmcnt<0>();                                  // k-tile i landed (and the previous tile's stores are out)
            __builtin_amdgcn_s_barrier();
            const int s_cur = (st + i) & 1;
            const char* a_s = sm + s_cur * STAGE;
            const char* b_s = a_s + OPA;
            u16x8 af[2][TM], bf[2][TN];
            auto load_frags = [&](int kk, int buf) {
                const int sl = kk * 2 + lhi;
#pragma unroll
                for (int ii = 0; ii < TM; ++ii) {
                    const int row = wm0 + ii * 32 + l31;
                    af[buf][ii] = *reinterpret_cast<const u16x8*>(a_s + row * 128 + ((sl ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = wn0 + j * 32 + l31;
                    bf[buf][j] = *reinterpret_cast<const u16x8*>(b_s + row * 128 + ((sl ^ ((row >> 1) & 7)) << 4));
                }
            };
            load_frags(0, 0);
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                if (kk + 1 < BK / 16) load_frags(kk + 1, (kk + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ii = 0; ii < TM; ++ii)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[ii][j] = mfma16<CT>(bf[kk & 1][j], af[kk & 1][ii], acc[ii][j]);   // transposed block
                __builtin_amdgcn_sched_barrier(0);
                if (kk == 0) {
                    if (i + 1 < cnkt) {
                        issue(i + 1, s_cur ^ 1);
                    } else {
                        next_tile = setup(tile + gridDim.x);  // pa / pb of the current tile are dead from here on
                        if (next_tile < total_tiles) issue(0, s_cur ^ 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // epilogue: lane = output row (l31 of a 32-row block), registers 4 g .. 4 g + 3 = columns 8 g + 4 lhi .. + 3
        {
            gf C = (gf)Pc->C;
            const float alpha = Pc->alpha_amax ? Pc->alpha * ghn3_pow2_inv_scale(*Pc->alpha_amax) : Pc->alpha;
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) {
                const int row = cm0 + wm0 + ii * 32 + l31;
                if (row >= Pc->M) continue;
                float GAS* crow = C + (int64_t)map_row(row, nullptr, Pc->c_q, Pc->c_s) * Pc->ldc;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = cn0 + wn0 + j * 32 + 8 * g + 4 * lhi;
                        if (col + 3 < Pc->N) {
                            f32x4 v = {acc[ii][j][4 * g] * alpha, acc[ii][j][4 * g + 1] * alpha,
                                       acc[ii][j][4 * g + 2] * alpha, acc[ii][j][4 * g + 3] * alpha};
                            *reinterpret_cast<gf4>(crow + col) = v;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (col + e < Pc->N) crow[col + e] = acc[ii][j][4 * g + e] * alpha;
                        }
                    }
            }
        }
        st = (st + cnkt) & 1;
        tile = next_tile;
    }
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 "ping-pong" variant of the 16-bit-operand kernel (tile code 26).
//
// The two-stage kernel above runs all 8 waves of a workgroup in the same phase: everybody reads fragments, then everybody
// multiplies, and the LDS-DMA of the next k-tile is issued in one burst -- measured 921-958 TF at 4096^3 against a
// ceiling of ~1.34 PF set by what one CU pulls through L2 -> LDS (~17 B/clk: 64 KB per 256 x 256 x 64 k-tile).  Here the
// workgroup is two wave GROUPS (wave rows wr = 0 / 1: one wave of each on every SIMD) that run HALF A PHASE APART: while
// one group multiplies, the other reads fragments, issues its share of the DMA and waits for it -- the matrix pipe of a
// SIMD always has one wave feeding it and one wave hiding memory work behind it.
//   * a k-tile (64 k) is four PHASES of 8 MFMAs (32 x 32 x 16): the wave's 128 x 64 output as quadrants Q(a, b) of
//     64 x 32 in the order Q00, Q01, Q11, Q10: phase 1 reads A sub-tile a = 0 (8 ds_read_b128) and B sub-tile b = 0 (4),
//     phase 2 B b = 1 (4), phase 3 A a = 1 (8), phase 4 nothing (B b = 0 is still in registers);
//   * ONE s_barrier per phase: in the interval behind it group 0 multiplies the phase first and reads the fragments of
//     the next one afterwards, group 1 reads first and multiplies afterwards -- a load segment = fragment reads, ONE
//     half-tile of DMA (2 pieces per thread) and a counted vmcnt;
//   * LDS: 2 k-tile buffers x 4 half-tiles of 16 KB.  A half-tile is defined by QUADRANT PARITY, not by contiguous rows:
//     A_h = the rows of sub-tile a = h of both wave rows, B_h = the columns of sub-tile b = h of the four wave columns,
//     so that phase 1 needs exactly {A_0, B_0}, phase 2 {B_1}, phase 3 {A_1}: the DMA is issued in that consumption
//     order, one half-tile per phase, each into its buffer two phases after the buffer's last read (the staggered
//     group's reads retire one barrier later) -- six half-tiles (96 KB) ahead of the phase that issues them, four
//     (64 KB) in flight at every wait;
//   * a half-tile is waited for (vmcnt(8): this thread's pieces of the four younger half-tiles may stay in flight) in
//     the phase BEFORE the one that reads it; the barriers in between make every wave's pieces visible.
// Same operand / epilogue contract as gemm_h16d_kernel (k-map, ragged extents, split-K, row-vector epilogue).
//
// MEASURED (round 2, MI355X, f16, tools/diag/gemm_bench.py shapes) and therefore OFF by default (GHN3_PINGPONG=1 selects it):
//   4096^3 847 TF (two-stage kernel 934), 8192^3 874 (854), W2 forward 768 x 147456 x 3072 854 (870), wgrad band 512 (546).
// Ablations of this kernel (GHN3_PP_DEBUG): without the DMA 1150-1390 TF-equivalent, without the MFMAs 1160-1710, without
// the fragment reads 902-933 -- the DMA stream and the matrix work each take ~60 % of the time of the full kernel and
// do NOT overlap the way the structure intends: per k-tile the load segments (12 ds_read_b128 + 2 DMA pieces per phase)
// cost ~360 cycles per phase, the vmcnt waits almost nothing -- the DMA pieces block at ISSUE (the L2 -> LDS path
// saturates at ~17-19 B/clk/CU, 64 KB per k-tile = ~3400 cycles against 2048 cycles of MFMA), and a wave blocked in a
// DMA issue does not release its SIMD's issue slots to the partner's MFMAs as cleanly as assumed.  The tile would have
// to move fewer bytes per flop (bigger than 256 x 256 does not fit the accumulators) for this structure to pay.
// ------------------------------------------------------------------------------------------------
template <int CT, int DBG = 0>      // DBG (ablation runs only): 1 = no DMA, 2 = no MFMA, 3 = no fragment reads
__device__ __forceinline__ void h16p_tile(const GemmProbDev* __restrict__ probs, int n_probs, int tile_id, float* smem) {
    constexpr int BM = 256, BN = 256, BK = 64, NT = 512;
    constexpr int HALF = 128 * 128;                  // bytes per half-tile image: 128 rows x 128 B
    char* sm = reinterpret_cast<char*>(smem);        // [k-tile parity][A0, B0, B1, A1][128][128 B]

    const GemmProbDev* P = find_problem(probs, n_probs, tile_id);
    const int t = tile_id - P->tile_start;
    int m0, n0, kz;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    bool live = tile_origin<BM, BN>(P, t, m0, n0, kz);
    const int l31 = lane & 31, lhi = lane >> 5;
    int Kp = P->K;
    if (live && P->lim) {
        const int* lim = P->lim;
        int ext = lim[m0 >> 7];
        if (m0 + 128 < P->M) ext = max(ext, lim[(m0 >> 7) + 1]);
        if (P->lim_kind == 1) { if (n0 >= ext) live = false; }
        else Kp = min(Kp, ext);
    }
    const int kt0 = live ? kz * (P->k_chunk / BK) : 0;
    const int K = (P->ksplit > 1) ? min(Kp, (kz + 1) * P->k_chunk) : Kp;
    if (live && kt0 * BK >= K && P->ksplit > 1) live = false;
    if (!live) return;                               // (uniform over the workgroup: nobody is left at a barrier)
    const int nkt = (K + BK - 1) / BK - kt0;

    // DMA: piece p = tid + 512 i (i = 0, 1) of a half-tile -> buffer row rho = p >> 3, slot p & 7 holding k-chunk
    // slot ^ ((rho >> 1) & 7).  A_h row rho <-> tile row 128 (rho >> 6) + 64 h + (rho & 63); B_h row rho <-> tile column
    // 64 (rho >> 5) + 32 h + (rho & 31).
    gch pa[2][2], pb[2][2];                          // [half][piece]
    int ck;
    {
        gch A = (gch)P->A; gch B = (gch)P->B;
        gci ag = (gci)P->a_gather; gci bg = (gci)P->b_gather;
        const int slot = tid & 7;
        ck = (slot ^ ((tid >> 4) & 7)) * 8;          // ((rho >> 1) & 7 is the same for both pieces: 64 rows apart)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rho = (tid >> 3) + 64 * i;
                const int ra = min(m0 + 128 * (rho >> 6) + 64 * h + (rho & 63), P->M - 1);
                const int rb = min(n0 + 64 * (rho >> 5) + 32 * h + (rho & 31), P->N - 1);
                pa[h][i] = A + (int64_t)map_row(ra, ag, P->a_q, P->a_s) * P->lda + ck;
                pb[h][i] = B + (int64_t)map_row(rb, bg, P->b_q, P->b_s) * P->ldb;
            }
    }
    const int kq = P->kq, ks = P->ks;
    const float inv_kq = kq > 0 ? 1.0f / (float)kq : 0.f;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // half-tile sequence index q = 4 T + j, j = 0..3 <-> (A0, B0, B1, A1) of k-tile T, stored at buffer (T & 1, j)
    auto issue = [&](int q) {
        const int T = q >> 2, j = q & 3;
        if (T >= nkt || DBG == 1) return;            // (past the last k-tile: nothing to fetch)
        const int k0 = (kt0 + T) * BK;
        LAS char* dst = (LAS char*)(sm + ((T & 1) * 4 + j) * HALF);
        if (j == 0 || j == 3) {
            const int h = j == 3;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((const void GAS*)(pa[h][i] + k0), (LAS void*)(dst + (wave * 64 + NT * i) * 16),
                                                 16, 0, 0);
        } else {
            const int h = j == 2;
            int kb = k0 + ck;
            if (kq > 0) {
                int qd = (int)((float)kb * inv_kq);
                int rm = kb - qd * kq;
                if (rm < 0) { rm += kq; --qd; } else if (rm >= kq) { rm -= kq; ++qd; }
                kb = qd * ks + rm;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((const void GAS*)(pb[h][i] + kb), (LAS void*)(dst + (wave * 64 + NT * i) * 16),
                                                 16, 0, 0);
        }
    };
    // pieces of this thread still allowed in flight so that half-tile `q` has landed, given `last` = newest issued index
    auto wait_for = [&](int q, int last) {
        const int total = 4 * nkt - 1;               // newest half-tile that exists
        const int younger = min(last, total) - q;    // issued after q (each 2 pieces per thread)
        if (younger >= 4) wait_vmcnt<8>();
        else if (younger == 3) wait_vmcnt<6>();
        else if (younger == 2) wait_vmcnt<4>();
        else if (younger == 1) wait_vmcnt<2>();
        else wait_vmcnt<0>();
    };

    // prologue: the first seven half-tiles; A0(0), B0(0) must have landed for phase 0
#pragma unroll
    for (int q = 0; q < 7; ++q) issue(q);
    wait_for(2, 6);                                  // (phase 0 reads A0, B0 and phase 1 B1 of k-tile 0)
    __builtin_amdgcn_s_barrier();

    u16x8 fa[2][4], fb[2][4];                        // A sub-tile fragments [32-row tile][k-step]; B fragments [b][k-step]
    // fragment reads of phase (T, ph): quadrant order Q00, Q01, Q11, Q10
    auto load_phase = [&](int T, int ph) {
        if (DBG == 3 && T > 0) return;
        const char* buf = sm + (T & 1) * 4 * HALF;
        if (ph == 0 || ph == 2) {                    // A sub-tile a = (ph == 2): buffer row 64 wr + 32 i + l31
            const char* a_s = buf + (ph == 0 ? 0 : 3) * HALF;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 64 * wr + 32 * i + l31;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    fa[i][kk] = *reinterpret_cast<const u16x8*>(a_s + row * 128 + (((kk * 2 + lhi) ^ ((row >> 1) & 7)) << 4));
            }
        }
        if (ph == 0 || ph == 1) {                    // B sub-tile b = ph: buffer row 32 wc + l31
            const char* b_s = buf + (1 + ph) * HALF;
            const int row = 32 * wc + l31;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                fb[ph][kk] = *reinterpret_cast<const u16x8*>(b_s + row * 128 + (((kk * 2 + lhi) ^ ((row >> 1) & 7)) << 4));
        }
    };
    auto mfma_phase = [&](int T, int ph) {
        if (DBG == 2 && T > 0) return;
        const int a = ph >> 1, b = (ph == 1 || ph == 2) ? 1 : 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[2 * a + i][b] = mfma16<CT>(fa[i][kk], fb[b][kk], acc[2 * a + i][b]);
        __builtin_amdgcn_s_setprio(0);
    };
    // ONE barrier per phase.  Interval k (after barrier k) computes phase k - 1: group 0 multiplies first and then reads the
    // fragments of phase k, group 1 reads the fragments of phase k - 1 first and multiplies then -- on every SIMD one
    // wave multiplies while the other one reads, issues its DMA pieces (half-tile k + 6) and waits for what phase k + 1
    // needs.  A buffer is refilled two intervals after its last read (group 1 reads a phase one interval later).
    if (wr == 0) load_phase(0, 0);
    for (int T = 0; T < nkt; ++T) {
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int k = 4 * T + ph + 1;            // interval index; phase k - 1 = (T, ph)
            __builtin_amdgcn_s_barrier();
            if (wr == 0) {
                mfma_phase(T, ph);
                __builtin_amdgcn_sched_barrier(0);
                if (ph < 3) load_phase(T, ph + 1); else if (T + 1 < nkt) load_phase(T + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                issue(k + 6);
                if (ph != 1) wait_for(k + 2, k + 6);   // (phase k + 1 = (.., 3) reads nothing new)
            } else {
                load_phase(T, ph);
                __builtin_amdgcn_sched_barrier(0);
                issue(k + 6);
                if (ph != 1) wait_for(k + 2, k + 6);
                __builtin_amdgcn_sched_barrier(0);
                mfma_phase(T, ph);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const int wm0 = 128 * wr, wn0 = 64 * wc;
    if (P->ksplit > 1) {
        epilogue_split<4>(P, acc, m0 + wm0, n0 + wn0, lane);
    } else {
        __syncthreads();                             // every wave is done with the operand buffers
        epilogue_rows<4>(P, acc, m0 + wm0, n0 + wn0, lane, smem + wave * (32 * 68));
    }
}

template <int CT, int DBG = 0>
__global__ __launch_bounds__(512, 2)
void gemm_h16p_kernel(const GemmProbDev* __restrict__ probs, int n_probs, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        h16p_tile<CT, DBG>(probs, n_probs, tile, smem);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
typedef void (*gemm_fn)(const GemmProbDev*, int);
// 16-bit-operand kernel variants -- tile code 16 / 24 picks the row
typedef void (*h16d_fn)(const GemmProbDev*, int, int);
struct H16dVariant { h16d_fn fn[2]; int threads, lds, edge; };
#define H16D(BMN, WGM, WGN, NS)                                                                          \
    {{gemm_h16d_kernel<GHN3_CT_F16, BMN, BMN, WGM, WGN, NS>, gemm_h16d_kernel<GHN3_CT_BF16, BMN, BMN, WGM, WGN, NS>}, \
     64 * WGM * WGN, NS * 2 * BMN * 64 * 2, BMN}
#define H16D_RECT(BM, BN, WGM, WGN, NS)                                                                  \
    {{gemm_h16d_kernel<GHN3_CT_F16, BM, BN, WGM, WGN, NS>, gemm_h16d_kernel<GHN3_CT_BF16, BM, BN, WGM, WGN, NS>}, \
     64 * WGM * WGN, NS * (BM + BN) * 64 * 2, BM}
// (deeper rings -- NS = 3 / 4 with one 128 x 128 workgroup per CU -- measured 25-35 % slower than two workgroups
// per CU with two stages: occupancy beats prefetch depth here)
static H16dVariant g_h16d[] = {
    H16D(128, 2, 2, 2),      // 0: 64 KB, 2 workgroups / CU
    H16D(256, 2, 4, 2),      // 1: 128 KB, 1 workgroup / CU
    H16D_RECT(256, 128, 4, 2, 3),   // 2 (tile code 20): 256 x 128, three stages (two k-tiles in flight), 144 KB
};
static int g_h16d_small = 0, g_h16d_big = 1;
static h16d_fn g_h16p[2] = {gemm_h16p_kernel<GHN3_CT_F16>, gemm_h16p_kernel<GHN3_CT_BF16>};
static h16d_fn g_h16w[2] = {gemm_h16w_kernel<GHN3_CT_F16>, gemm_h16w_kernel<GHN3_CT_BF16>};
static int g_n_cu = 256;
static h16d_fn g_h16p_dbg[4] = {nullptr, gemm_h16p_kernel<GHN3_CT_F16, 1>, gemm_h16p_kernel<GHN3_CT_F16, 2>,
                                gemm_h16p_kernel<GHN3_CT_F16, 3>};
static int g_pp_dbg = 0;
static int g_pingpong = 0;      // GHN3_PINGPONG=1: 256 x 256 problems on the ping-pong kernel (measured no faster, see above)

template <int BM, int BN, int AM, int BMD> static size_t f32_lds() {
    return 2 * (size_t)(F32Tile<BM, AM>::SIZE + F32Tile<BN, BMD>::SIZE) * sizeof(float);
}
template <int BM, int BN> static size_t h16_lds() {
    return 2 * (size_t)(BM + BN) * (64 + 8) * sizeof(unsigned short);
}

struct GemmVariant { gemm_fn fn; size_t lds; };

template <int BM, int BN, int AM, int BMD> static GemmVariant f32_variant() {
    return GemmVariant{(gemm_fn)gemm_f32_kernel<BM, BN, AM, BMD>, f32_lds<BM, BN, AM, BMD>()};
}
template <int BM, int BN, int AM, int BMD, int CT> static GemmVariant h16_variant() {
    return GemmVariant{(gemm_fn)gemm_h16_kernel<BM, BN, AM, BMD, CT>, h16_lds<BM, BN>()};
}

// [ctype][tile(0:64,1:128)][a_mode][b_mode]
static GemmVariant g_variants[3][2][2][2];
static bool g_gemm_ready = false;

#define FILL_F32(T, TI)                                              \
    g_variants[GHN3_CT_F32][TI][0][0] = f32_variant<T, T, ROWM, ROWM>(); \
    g_variants[GHN3_CT_F32][TI][0][1] = f32_variant<T, T, ROWM, COLM>(); \
    g_variants[GHN3_CT_F32][TI][1][0] = f32_variant<T, T, COLM, ROWM>(); \
    g_variants[GHN3_CT_F32][TI][1][1] = f32_variant<T, T, COLM, COLM>();
#define FILL_H16(T, TI, CT)                                              \
    g_variants[CT][TI][0][0] = h16_variant<T, T, ROWM, ROWM, CT>(); \
    g_variants[CT][TI][0][1] = h16_variant<T, T, ROWM, COLM, CT>(); \
    g_variants[CT][TI][1][0] = h16_variant<T, T, COLM, ROWM, CT>(); \
    g_variants[CT][TI][1][1] = h16_variant<T, T, COLM, COLM, CT>();

int ghn3_gemm_init() {
    if (g_gemm_ready) return GHN3_OK;
    FILL_F32(64, 0)
    FILL_F32(128, 1)
    FILL_H16(64, 0, GHN3_CT_F16)
    FILL_H16(128, 1, GHN3_CT_F16)
    FILL_H16(64, 0, GHN3_CT_BF16)
    FILL_H16(128, 1, GHN3_CT_BF16)
    for (int c = 0; c < 3; ++c)
        for (int t = 0; t < 2; ++t)
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    GemmVariant& v = g_variants[c][t][a][b];
                    if (v.lds > 48 * 1024) {
                        hipError_t e = hipFuncSetAttribute((const void*)v.fn,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds);
                        if (e != hipSuccess) {
                            ghn3_set_error("hipFuncSetAttribute(gemm lds=%zu): %s", v.lds, hipGetErrorString(e));
                            return GHN3_E_HIP;
                        }
                    }
                }
    for (const H16dVariant& v : g_h16d)
        for (int ct = 0; ct < 2; ++ct) {
            hipError_t e = hipFuncSetAttribute((const void*)v.fn[ct], hipFuncAttributeMaxDynamicSharedMemorySize, v.lds);
            if (e != hipSuccess) { ghn3_set_error("hipFuncSetAttribute(h16d): %s", hipGetErrorString(e)); return GHN3_E_HIP; }
        }
    for (int ct = 0; ct < 2; ++ct) {
        hipError_t e = hipFuncSetAttribute((const void*)g_h16p[ct], hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) { ghn3_set_error("hipFuncSetAttribute(h16p): %s", hipGetErrorString(e)); return GHN3_E_HIP; }
        e = hipFuncSetAttribute((const void*)g_h16w[ct], hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) { ghn3_set_error("hipFuncSetAttribute(h16w): %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    }
    {
        int dev = 0, n_cu = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n_cu > 0)
            g_n_cu = n_cu;
    }
    if (getenv("GHN3_PINGPONG")) g_pingpong = atoi(getenv("GHN3_PINGPONG"));
    if (getenv("GHN3_PP_DEBUG")) g_pp_dbg = atoi(getenv("GHN3_PP_DEBUG")) & 3;
    for (int d = 1; d < 4; ++d)
        (void)hipFuncSetAttribute((const void*)g_h16p_dbg[d], hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    g_gemm_ready = true;
    return GHN3_OK;
}

int ghn3_gemm_h16d_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int tile, int ctype, int grid_cap,
                          hipStream_t stream) {
    if (n_probs <= 0 || total_tiles <= 0) return GHN3_OK;
    if ((ctype != GHN3_CT_F16 && ctype != GHN3_CT_BF16) || (tile != 128 && tile != 256 && tile != 20 && tile != 25)) {
        ghn3_set_error("16-bit-operand GEMM needs compute type f16 or bf16 (got %d) and tile 128 / 256 (got %d)", ctype,
                       tile);
        return GHN3_E_ARG;
    }
    if (tile == 25) {
        // persistent by construction: one workgroup per CU (or per CU the caller leaves to this launch)
        const int want = grid_cap > 0 ? grid_cap : g_n_cu;
        hipLaunchKernelGGL(g_h16w[ctype == GHN3_CT_BF16], dim3(want < total_tiles ? want : total_tiles), dim3(512),
                           128 * 1024, stream, d_probs, n_probs, total_tiles);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { ghn3_set_error("h16w gemm launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
        return GHN3_OK;
    }
    // grid_cap counts CUs' worth of workgroups: the 128 x 128 variant runs two workgroups per CU
    if (tile == 128) grid_cap *= 2;
    const int grid = grid_cap > 0 && grid_cap < total_tiles ? grid_cap : total_tiles;
    if (tile == 256 && g_pingpong) {
        hipLaunchKernelGGL(g_pp_dbg ? g_h16p_dbg[g_pp_dbg] : g_h16p[ctype == GHN3_CT_BF16], dim3(grid), dim3(512),
                           128 * 1024, stream, d_probs, n_probs, total_tiles);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { ghn3_set_error("h16p gemm launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
        return GHN3_OK;
    }
    const H16dVariant& v = g_h16d[tile == 20 ? 2 : tile == 256 ? g_h16d_big : g_h16d_small];
    hipLaunchKernelGGL(v.fn[ctype == GHN3_CT_BF16], dim3(grid), dim3(v.threads), v.lds, stream, d_probs, n_probs,
                       total_tiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ghn3_set_error("h16d gemm launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    return GHN3_OK;
}

int ghn3_gemm_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int a_mode, int b_mode, int tile,
                     int ctype, hipStream_t stream) {
    if (n_probs <= 0 || total_tiles <= 0) return GHN3_OK;
    if (ctype < 0 || ctype > 2 || (tile != 64 && tile != 128)) {
        ghn3_set_error("gemm: bad ctype/tile %d/%d", ctype, tile);
        return GHN3_E_ARG;
    }
    const GemmVariant& v = g_variants[ctype][tile == 128][a_mode][b_mode];
    hipLaunchKernelGGL(v.fn, dim3(total_tiles), dim3(256), v.lds, stream, d_probs, n_probs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ghn3_set_error("gemm launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    return GHN3_OK;
}
