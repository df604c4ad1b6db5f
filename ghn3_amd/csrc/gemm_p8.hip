// 8-phase 16-bit-operand GEMM for gfx950 (tile code 28): the decoder W2 forward / dgrad / weight gradient of round 3.
//
// Replaces the matmuls of `ConvDecoder3.forward` (/root/reference/ghn3/nn.py:742-750: conv.2 over the cropped positions)
// and their autograd backward.  Same problem contract as the other GHN3_GEMM_OP16 kernels (include/ghn3_hip.h): both
// operands 16-bit copies already in HBM, k-contiguous, zero padded to 64 along k; C fp32; optional row maps of A / B / C,
// k-map of B, ragged extents, XCD pins, fused epilogue (alpha, bias, ReLU, dReLU, residual, accumulate, aux_out).
//
// Why a new kernel.  The two-stage kernel of round 2 (gemm.hip: h16d_tile<256, 256>) keeps ONE k-tile in flight, issues it
// after the first MFMA batch of the current tile and drains it with vmcnt(0) at the next barrier: 850-940 TF.  Here
// (tools/gemm_lab.hip: 1030-1245 TF on the same shapes, same box):
//   * 512 threads = 8 waves: wave row wr = wave >> 2 (half of the tile's rows), wave column wc = wave & 3 (64 columns).
//   * A k-tile (64 k) is four PHASES, one quadrant Q(a, b) of the wave's (32 MI) x 64 output each, in the order Q00, Q01,
//     Q11, Q10: 4 MI v_mfma_f32_16x16x32 on 2 MI independent accumulators.  Tile height BM = 64 MI, MI in {3, 4, 5}:
//     a family of 533 full-width decoder rows is 320 + 256 rows instead of three 256-row tiles (24 % padding before).
//   * LDS: ring of 2 k-tiles x 4 half-tiles [A0 | B0 | B1 | A1]: A_h = rows of sub-tile a = h of both wave rows (32 MI
//     rows), B_h = columns of sub-tile b = h of the four wave columns (128 columns); 128-byte rows, 16-byte slot s of row
//     r holds k chunk s ^ ((r >> 1) & 7) (conflict-free ds_read_b128 fragments for the 16-lane groups of 16x16x32).
//   * Every phase issues ONE half-tile of LDS-DMA (issue order B0, A0, B1, A1), 7 half-tiles ahead of the phase; ONE
//     counted vmcnt per k-tile (phase 3: the next k-tile has landed, the three newest half-tiles stay in flight) -- the
//     DMA queue is never drained inside the loop.  A buffer is re-staged two phases after its last read, except B0: one
//     phase, but its four ds_reads are issued first in phase 0 and retired by a counted lgkmcnt before that phase's first
//     barrier.  A staged half-tile is read at the earliest in the phase after the wait that retires it.
//   * Two barriers per phase; wave row 1 runs one barrier behind wave row 0: on every SIMD one wave multiplies while its
//     partner reads fragments and issues DMA (cdna_hip_programming.md "256^2 8-phase template",
//     MI355X_MICROARCH.md "Two waves per SIMD").
//   * The products are taken transposed (B fragment = first MFMA operand): a lane owns 4 consecutive output columns of a
//     row, the epilogue is float4 accesses straight from the accumulators, no LDS staging.
// No split-K with atomics (K splits are separate problems writing partial planes), no gathers, no GELU.

#include "ghn3_internal.h"

#define GAS __attribute__((address_space(1)))
#define LAS __attribute__((address_space(3)))
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef float GAS* gf;
typedef const float GAS* gcf;
typedef f32x4 GAS* gf4;
typedef const f32x4 GAS* gcf4;

namespace {

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ int p8_map_row(int r, int q, int s) {
    if (q > 0) {          // (r / q) * s + r % q, exact for 0 <= r < 2^24 (float reciprocal + correction)
        int d = (int)((float)r * __builtin_amdgcn_rcpf((float)q));
        int m = r - d * q;
        if (m < 0) { m += q; --d; } else if (m >= q) { m -= q; ++d; }
        if (m < 0) { m += q; --d; } else if (m >= q) { m -= q; ++d; }
        r = d * s + m;
    }
    return r;
}

template <int CT>
__device__ __forceinline__ f32x4 mfma16x16(u16x8 a, u16x8 b, f32x4 c) {
    if (CT == GHN3_CT_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// One output tile: rows [m0, m0 + 64 MI), columns [n0, n0 + 256), reduction over [0, K) (K > 0 or K == 0: zeros).
template <int CT, int MI>
__device__ __forceinline__ void p8_tile(const GemmProbDev* __restrict__ P, const int m0, const int n0, const int K, char* sm) {
    constexpr int BM = 64 * MI, BK = 64, D = 7;
    constexpr int AH = 32 * MI * 128, BH = 128 * 128, KT = 2 * AH + 2 * BH;     // bytes: A half, B half, one k-tile of the ring
    constexpr int NFA = MI / 2, ODD = MI & 1, NPA = NFA + ODD;                   // DMA rounds of an A half-tile (last one: 32 lanes)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r16 = lane & 15, kc = lane >> 4;
    const int M = P->M, N = P->N;
    const int nkt = (K + BK - 1) / BK;
    const int nq = 4 * nkt;
    const char GAS* Ab = (const char GAS*)P->A;
    const char GAS* Bb = (const char GAS*)P->B;

    // byte offsets of this thread's DMA pieces at k = 0 (the host checked that both operands span < 4 GB)
    unsigned oa[2][NPA], ob[2][2];
    const int slot = tid & 7, rb = tid >> 3;
    const int ck = (slot ^ ((rb >> 1) & 7)) * 8;     // k offset (16-bit elements) of this lane's chunk in the full rounds
    {
        const unsigned lda2 = (unsigned)P->lda * 2u, ldb2 = (unsigned)P->ldb * 2u;
        const int aq = P->a_q, as = P->a_s, bq = P->b_q, bs = P->b_s;
        const int kq0 = P->kq;
        const int ckb = (kq0 > 0 && kq0 < 64) ? (ck / kq0) * P->ks + ck % kq0 : ck;   // k-map, per-lane part (64 % kq == 0)
        auto arow = [&](int rho, int h) -> unsigned {    // buffer row of A_h -> operand row
            const int w = rho >= 16 * MI;
            const int r = min(m0 + w * (BM / 2) + h * 16 * MI + (rho - w * 16 * MI), M - 1);
            return (unsigned)p8_map_row(r, aq, as);
        };
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < NFA; ++i) oa[h][i] = arow(rb + 64 * i, h) * lda2 + (unsigned)ck * 2u;
            if (ODD) {
                const int l = lane & 31;
                const int rho = 64 * NFA + wave * 4 + (l >> 3);
                oa[h][NFA] = arow(rho, h) * lda2 + (unsigned)(((l & 7) ^ ((rho >> 1) & 7)) * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rho = rb + 64 * i;
                const int r = min(n0 + 64 * (rho >> 5) + 32 * h + (rho & 31), N - 1);
                ob[h][i] = (unsigned)p8_map_row(r, bq, bs) * ldb2 + (unsigned)ckb * 2u;
            }
        }
    }
    // k position of the issue stream.  Half-tiles are issued strictly in order (B0, A0, B1, A1 of k-tile 0, then of k-tile
    // 1, ...), so the byte offsets along k are wave-uniform running values: A advances 128 bytes per k-tile, B follows the
    // k-map (physical k = (k / kq) * ks + k % kq; the host admits kq % 64 == 0 -- a k-tile never straddles a period -- or
    // 64 % kq == 0 -- the per-lane part is constant and already folded into ob[][]).
    const int kq = P->kq, ks = P->ks;
    int kA = 0, kB = 0, tmod = 0, par = 0, qi = 0;   // bytes, bytes, k inside the current period, ring parity offset, next q
    const int stepB = kq > 0 && kq < 64 ? (64 / kq) * ks * 2 : 128;
    auto issue = [&](const int j) {                   // j = (next half-tile) & 3, known at every call site
        if (qi < nq) {
            char* kt = sm + par;
            if (j == 1 || j == 3) {
                const int h = j == 3;
                LAS char* dst = (LAS char*)(kt + (h ? AH + 2 * BH : 0));
#pragma unroll
                for (int i = 0; i < NFA; ++i)
                    __builtin_amdgcn_global_load_lds((const void GAS*)(Ab + (oa[h][i] + (unsigned)kA)), (LAS void*)(dst + i * 8192 + wave * 1024), 16, 0, 0);
                if (ODD && lane < 32)
                    __builtin_amdgcn_global_load_lds((const void GAS*)(Ab + (oa[h][NFA] + (unsigned)kA)), (LAS void*)(dst + NFA * 8192 + wave * 512), 16, 0, 0);
            } else {
                const int h = j == 2;
                LAS char* dst = (LAS char*)(kt + AH + h * BH + wave * 1024);
                __builtin_amdgcn_global_load_lds((const void GAS*)(Bb + (ob[h][0] + (unsigned)kB)), (LAS void*)dst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const void GAS*)(Bb + (ob[h][1] + (unsigned)kB)), (LAS void*)(dst + 8192), 16, 0, 0);
            }
        }
        ++qi;
        if (j == 2) {                                 // B of this k-tile is out: advance along the k-map
            if (kq >= 64) {
                tmod += 64;
                if (tmod >= kq) { tmod -= kq; kB += (ks - kq + 64) * 2; } else kB += 128;
            } else kB += stepB;
        }
        if (j == 3) { kA += 128; par = KT - par; }
    };

    f32x4 acc[2][2][MI][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[a][b][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragments: lane l -> row r16 = l & 15 of a 16-row MFMA tile, 16-byte chunk 4 ks + kc (kc = l >> 4)
    const int sw = r16 >> 1;
    const int offA = (16 * MI * wr + r16) * 128 + ((kc ^ sw) << 4);       // + mi * 2048; ^ 64 for the second k-step
    const int offB = AH + (32 * wc + r16) * 128 + ((kc ^ sw) << 4);       // + ni * 2048
    u16x8 fa[MI][2], fb[2][2][2];
    auto read_a = [&](int T, int a) {
        const char* base = sm + (T & 1) * KT + (a ? AH + 2 * BH : 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            fa[mi][0] = *reinterpret_cast<const u16x8*>(base + offA + mi * 2048);
            fa[mi][1] = *reinterpret_cast<const u16x8*>(base + (offA ^ 64) + mi * 2048);
        }
    };
    auto read_b = [&](int T, int b) {
        const char* base = sm + (T & 1) * KT + b * BH;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            fb[b][ni][0] = *reinterpret_cast<const u16x8*>(base + offB + ni * 2048);
            fb[b][ni][1] = *reinterpret_cast<const u16x8*>(base + (offB ^ 64) + ni * 2048);
        }
    };
    auto mfma_q = [&](int a, int b) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)       // transposed product: lane = output row, registers = 4 columns
                    acc[a][b][mi][ni] = mfma16x16<CT>(fb[b][ni][k2], fa[mi][k2], acc[a][b][mi][ni]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // everything up to half-tile `needed` has landed (this thread's pieces); `last` = newest half-tile issued so far
    auto wait_landed = [&](int last, int needed) {
        last = min(last, nq - 1);
        const int keep = last - needed;               // half-tiles that may stay in flight (3 in steady state)
        int n = 0;
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if (e < keep && last - e >= 0) { const int j = (last - e) & 3; n += (j == 1 || j == 3) ? NPA : 2; }
        if (n >= 2 * NPA + 2) wait_vm<2 * NPA + 2>();
        else if (n >= NPA + 4 && NPA + 4 < 2 * NPA + 2) wait_vm<NPA + 4>();
        else if (n >= NPA + 2) wait_vm<NPA + 2>();
        else if (n >= 4 && 4 < NPA + 2) wait_vm<4>();
        else if (n >= NPA && NPA >= 2) wait_vm<(NPA >= 2 ? NPA : 2)>();
        else if (n >= 2) wait_vm<2>();
        else wait_vm<0>();
    };

#pragma unroll
    for (int q = 0; q < D; ++q) issue(q & 3);
    wait_landed(D - 1, 3);                            // k-tile 0 (half-tiles 0..3)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();        // wave row 1 runs one barrier behind

    for (int T = 0; T < nkt; ++T) {
        const int g = 4 * T;
        read_b(T, 0);                                 // (first: retired by the counted lgkmcnt below)
        __builtin_amdgcn_sched_barrier(0);
        read_a(T, 0);
        __builtin_amdgcn_sched_barrier(0);
        issue(3);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * MI) : "memory");   // the B0 reads are done: its buffer is re-staged next phase
        __builtin_amdgcn_s_barrier();
        mfma_q(0, 0);
        __builtin_amdgcn_s_barrier();
        read_b(T, 1);
        issue(0);
        __builtin_amdgcn_s_barrier();
        mfma_q(0, 1);
        __builtin_amdgcn_s_barrier();
        read_a(T, 1);
        issue(1);
        __builtin_amdgcn_s_barrier();
        mfma_q(1, 1);
        __builtin_amdgcn_s_barrier();
        issue(2);
        wait_landed(g + 3 + D, g + 7);                // k-tile T + 1 complete
        __builtin_amdgcn_s_barrier();
        mfma_q(1, 0);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();        // pairs the extra barrier of wave row 1

    // ---- epilogue: lane = output row (r16 of a 16-row MFMA tile), its 4 registers = columns 4 kc .. 4 kc + 3 (N % 4 == 0:
    // a lane's four columns are all inside or all outside).  Order: acc * alpha -> + bias -> aux_out -> ReLU -> dReLU(aux_in)
    // -> + residual -> (+ C) -> store
    {
        gf C = (gf)P->C;
        gcf residual = (gcf)P->residual;
        gcf aux_in = (gcf)P->aux_in;
        gf aux_out = (gf)P->aux_out;
        gcf bias = (P->flags & GHN3_GEMM_BIASGRAD) ? nullptr : (gcf)P->bias;
        const int ldc = P->ldc, cq = P->c_q, cs = P->c_s;
        const bool act_relu = P->act == GHN3_ACT_RELU, dact_relu = P->dact == GHN3_DACT_RELU;
        const bool accum = (P->flags & GHN3_GEMM_ACCUM) != 0;
        const float alpha = P->alpha_amax ? P->alpha * ghn3_pow2_inv_scale(*P->alpha_amax) : P->alpha;
        const int col0 = n0 + wc * 64 + 4 * kc;       // + 32 b + 16 ni
        f32x4 bv[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int col = col0 + b * 32 + ni * 16;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (bias && col < N) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int bi = col + e;
                        if (P->bias_q > 0) bi = (bi / P->bias_q) * P->bias_s + (bi % P->bias_q);
                        v[e] = bias[(int64_t)bi * P->bias_stride];
                    }
                }
                bv[b][ni] = v;
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = m0 + wr * (BM / 2) + a * 16 * MI + mi * 16 + r16;
                if (row >= M) continue;
                const int64_t rbase = (int64_t)p8_map_row(row, cq, cs) * ldc;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const int col = col0 + b * 32 + ni * 16;
                        if (col >= N) continue;
                        const int64_t ci = rbase + col;
                        f32x4 v = acc[a][b][mi][ni] * alpha + bv[b][ni];
                        if (aux_out) *reinterpret_cast<gf4>(aux_out + ci) = v;
                        if (act_relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        if (dact_relu) {
                            const f32x4 x = *reinterpret_cast<gcf4>(aux_in + ci);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = x[e] > 0.f ? v[e] : 0.f;
                        }
                        if (residual) v += *reinterpret_cast<gcf4>(residual + ci);
                        if (accum) v += *reinterpret_cast<gcf4>(C + ci);
                        *reinterpret_cast<gf4>(C + ci) = v;
                    }
            }
    }
}

// Row tiles.  With a row-tile table (GemmProbDev::mtab: int32 triples {m0, MI, extent}, written by the host for the
// stacked decoder families) tile mt covers rows [m0, m0 + 64 MI) and carries its own ragged extent; without one the
// problem is cut into 256-row tiles and the extent comes from the per-128-row `lim` array as in the older kernels.
template <int CT>
__device__ __forceinline__ void p8_dispatch(const GemmProbDev* __restrict__ probs, int n_probs, int tile_id, char* sm) {
    const GemmProbDev* P;
    int mt, nt;
    const int pin_end = probs[0].pin_end;
    if (tile_id < pin_end) {
        // XCD-pinned problems: id -> (XCD, local index); the problems of an XCD are found through the directory in entry x
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
        mt = t % P->tiles_m;                          // row tiles of a column tile back to back: they share its B rows
        nt = t / P->tiles_m;
    } else {
        const int n_pin = probs[0].pin_total;
        int lo = n_pin, hi = n_probs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (probs[mid].tile_start <= tile_id) lo = mid; else hi = mid - 1;
        }
        P = probs + lo;
        // the row tiles of one column tile (same streamed B panel) are congruent mod 8: same XCD, same L2
        const int t = tile_id - P->tile_start, grp = t >> 3;
        nt = (grp / P->tiles_m) * 8 + (t & 7);
        mt = grp % P->tiles_m;
        if (nt >= P->tiles_n) return;
    }
    int m0, mi, ext;
    if (P->mtab) {
        const int* e = P->mtab + 3 * mt;
        m0 = e[0]; mi = e[1]; ext = e[2];
    } else {
        m0 = mt * 256; mi = 4; ext = 0x7fffffff;
        if (P->lim) {
            ext = P->lim[m0 >> 7];
            if (m0 + 128 < P->M) ext = max(ext, P->lim[(m0 >> 7) + 1]);
        }
    }
    if (m0 >= P->M) return;
    const int n0 = nt * 256;
    int K = P->K;
    if (P->lim_kind == 1) { if (n0 >= ext) return; }
    else if (P->lim_kind == 2) K = min(K, ext);
    // (MI = 5 -- 320 rows -- is written and verified in tools/gemm_lab.hip, but with the row / k maps of the library
    // contract its 160 accumulators + 72 fragment registers leave no room: the compiler spills an address register inside
    // the loop and the reload drains the DMA queue.  Every row count >= 384 is a sum of 192s and 256s rounded up to 64, so
    // the row padding is the same; 192-row tiles cost ~15 % more per row.)
    if (mi == 3) p8_tile<CT, 3>(P, m0, n0, K, sm);
    else p8_tile<CT, 4>(P, m0, n0, K, sm);
}

template <int CT>
__global__ __launch_bounds__(512, 2) void gemm_p8_kernel(const GemmProbDev* __restrict__ probs, int n_probs, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        p8_dispatch<CT>(probs, n_probs, tile, reinterpret_cast<char*>(smem));
        __syncthreads();                               // the ring is reused by the next tile
    }
}

constexpr int kP8Lds = 2 * (2 * 4 * 4096 + 32768);    // MI = 4: 128 KB

}  // namespace

static bool g_p8_ready = false;

int ghn3_gemm_p8_init() {
    if (g_p8_ready) return GHN3_OK;
    hipError_t e = hipFuncSetAttribute((const void*)gemm_p8_kernel<GHN3_CT_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, kP8Lds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)gemm_p8_kernel<GHN3_CT_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, kP8Lds);
    if (e != hipSuccess) { ghn3_set_error("hipFuncSetAttribute(p8): %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    g_p8_ready = true;
    return GHN3_OK;
}

int ghn3_gemm_p8_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int ctype, int grid_cap, hipStream_t stream) {
    if (n_probs <= 0 || total_tiles <= 0) return GHN3_OK;
    if (ctype != GHN3_CT_F16 && ctype != GHN3_CT_BF16) {
        ghn3_set_error("8-phase GEMM needs compute type f16 or bf16 (got %d)", ctype);
        return GHN3_E_ARG;
    }
    int rc = ghn3_gemm_p8_init();
    if (rc) return rc;
    const int grid = grid_cap > 0 && grid_cap < total_tiles ? grid_cap : total_tiles;
    if (ctype == GHN3_CT_F16)
        hipLaunchKernelGGL(gemm_p8_kernel<GHN3_CT_F16>, dim3(grid), dim3(512), kP8Lds, stream, d_probs, n_probs, total_tiles);
    else
        hipLaunchKernelGGL(gemm_p8_kernel<GHN3_CT_BF16>, dim3(grid), dim3(512), kP8Lds, stream, d_probs, n_probs, total_tiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ghn3_set_error("p8 gemm launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    return GHN3_OK;
}
