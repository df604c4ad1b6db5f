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
// cache-policy bits of the LDS-DMA loads of the forward / dgrad kernel (aux: 1 = sc0, 2 = nt, 16 = sc1); experiments only
#ifndef P8_AUX_A
#define P8_AUX_A 0
#endif
#ifndef P8_AUX_B
#define P8_AUX_B 0
#endif
#define LAS __attribute__((address_space(3)))
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef float GAS* gf;
typedef const float GAS* gcf;
typedef f32x4 GAS* gf4;
typedef const f32x4 GAS* gcf4;

#ifdef GHN3_P8_PROBE
// tools/p8_probe.hip: cycles summed over every tile of a launch by thread 0 of its workgroup --
// [0] tiles, [1] prologue (entry -> first k-tile landed), [2] k loop, [3] inside the k loop's DMA waits, [4] epilogue, [5] k-tiles
__device__ unsigned long long g_p8_probe[8];
#define P8_CLK() ((long long)__builtin_readcyclecounter())
#endif

namespace {

// Values read from the problem table are wave-uniform, but the compiler cannot prove it (the table index comes out of a
// search over loaded values): without help they live in VGPRs, every "uniform" branch becomes an exec-mask dance and the
// operand bases are re-read with v_readfirstlane at each DMA.  rfl() pins them to SGPRs.
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float rflf(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
template <typename T> __device__ __forceinline__ T* rflp(T* p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}

template <int N> struct P8Int { static constexpr int value = N; };

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void p8_divmod(int r, int q, int& d, int& m) {   // exact for 0 <= r < 2^24, q > 0 (float reciprocal + correction)
    d = (int)((float)r * __builtin_amdgcn_rcpf((float)q));
    m = r - d * q;
    if (m < 0) { m += q; --d; } else if (m >= q) { m -= q; ++d; }
    if (m < 0) { m += q; --d; } else if (m >= q) { m -= q; ++d; }
}
__device__ __forceinline__ int p8_map_row(int r, int q, int s) {
    if (q > 0) {          // (r / q) * s + r % q
        int d, m;
        p8_divmod(r, q, d, m);
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

// One output tile: rows [m0, m0 + 32 (MI0 + MI1)), columns [n0, n0 + 256), reduction over [0, K) (K > 0 or K == 0: zeros).
// Round 6: the two row sub-tiles of a wave may differ by one 16-row MFMA tile (MI0 = MI1 or MI1 + 1): tile heights of 192,
// 224, 256, 288, 320 rows, so that the row tiles of one streamed W2 panel can be EQUAL (533 rows = 288 + 288 instead of
// 256 + 320): equal tiles on the CUs of one XCD run at the same pace and share the panel through the XCD's L2 for their
// whole length, unequal ones drift apart by a fifth of a tile and fetch it twice (5.0 GB counted for 2.9 GB consumed, round 5).
template <int CT, int MI0, int MI1>
__device__ __forceinline__ void p8_tile(const GemmProbDev* __restrict__ P, const int m0, const int n0, const int K, char* sm) {
    static_assert(MI0 >= MI1 && MI0 - MI1 <= 1, "sub-tile heights");
    constexpr int MI = MI0;                                                       // (array extents: the larger sub-tile)
    constexpr int BM = 32 * (MI0 + MI1), BK = 64, D = 7;
    constexpr int AH0 = 32 * MI0 * 128, AH1 = 32 * MI1 * 128, BH = 128 * 128, KT = AH0 + AH1 + 2 * BH;   // bytes: A halves, B half, one k-tile of the ring
    constexpr int NFA0 = MI0 / 2, ODD0 = MI0 & 1, NPA0 = NFA0 + ODD0;            // DMA rounds of A half-tile 0 (last one: 32 lanes)
    constexpr int NFA1 = MI1 / 2, ODD1 = MI1 & 1, NPA1 = NFA1 + ODD1;
    constexpr int NPA = NPA0 > NPA1 ? NPA0 : NPA1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r16 = lane & 15, kc = lane >> 4;
    const int M = rfl(P->M), N = rfl(P->N);
    const int nkt = (K + BK - 1) / BK;
    const int nq = 4 * nkt;
    const char GAS* Ab = (const char GAS*)rflp(P->A);
    const char GAS* Bb = (const char GAS*)rflp(P->B);

    // byte offsets of this thread's DMA pieces at k = 0 (the host checked that both operands span < 4 GB)
    unsigned oa[2][NPA], ob[2][2];
    const int slot = tid & 7, rb = tid >> 3;
    const int ck = (slot ^ ((rb >> 1) & 7)) * 8;     // k offset (16-bit elements) of this lane's chunk in the full rounds
    {
        const unsigned lda2 = (unsigned)rfl(P->lda) * 2u, ldb2 = (unsigned)rfl(P->ldb) * 2u;
        const int aq = rfl(P->a_q), as = rfl(P->a_s), bq = rfl(P->b_q), bs = rfl(P->b_s);
        const int kq0 = rfl(P->kq);
        const int ckb = (kq0 > 0 && kq0 < 64) ? (ck / kq0) * rfl(P->ks) + ck % kq0 : ck;   // k-map, per-lane part (64 % kq == 0)
        auto arow = [&](int rho, int h) -> unsigned {    // buffer row of A_h -> operand row
            const int mih = h ? MI1 : MI0;
            const int w = rho >= 16 * mih;
            const int r = min(m0 + w * (BM / 2) + (h ? 16 * MI0 : 0) + (rho - w * 16 * mih), M - 1);
            return (unsigned)p8_map_row(r, aq, as);
        };
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int nfa = h ? NFA1 : NFA0, odd = h ? ODD1 : ODD0;
#pragma unroll
            for (int i = 0; i < NPA; ++i)
                if (i < nfa) oa[h][i] = arow(rb + 64 * i, h) * lda2 + (unsigned)ck * 2u;
            if (odd) {
                const int l = lane & 31;
                const int rho = 64 * nfa + wave * 4 + (l >> 3);
                oa[h][nfa] = arow(rho, h) * lda2 + (unsigned)(((l & 7) ^ ((rho >> 1) & 7)) * 16);
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
    const int kq = rfl(P->kq), ks = rfl(P->ks);
    int kA = 0, kB = 0, tmod = 0, par = 0, qi = 0;   // bytes, bytes, k inside the current period, ring parity offset, next q
    const int stepB = kq > 0 && kq < 64 ? (64 / kq) * ks * 2 : 128;
    auto issue = [&](const int j) {                   // j = (next half-tile) & 3, known at every call site
        if (qi < nq) {
            char* kt = sm + par;
            if (j == 1 || j == 3) {
                const int h = j == 3;
                const int nfa = h ? NFA1 : NFA0, odd = h ? ODD1 : ODD0;
                LAS char* dst = (LAS char*)(kt + (h ? AH0 + 2 * BH : 0));
#pragma unroll
                for (int i = 0; i < NPA; ++i)
                    if (i < nfa)
                        __builtin_amdgcn_global_load_lds((const void GAS*)(Ab + (oa[h][i] + (unsigned)kA)), (LAS void*)(dst + i * 8192 + wave * 1024), 16, 0, P8_AUX_A);
                if (odd && lane < 32)
                    __builtin_amdgcn_global_load_lds((const void GAS*)(Ab + (oa[h][nfa] + (unsigned)kA)), (LAS void*)(dst + nfa * 8192 + wave * 512), 16, 0, P8_AUX_A);
            } else {
                const int h = j == 2;
                LAS char* dst = (LAS char*)(kt + AH0 + h * BH + wave * 1024);
                __builtin_amdgcn_global_load_lds((const void GAS*)(Bb + (ob[h][0] + (unsigned)kB)), (LAS void*)dst, 16, 0, P8_AUX_B);
                __builtin_amdgcn_global_load_lds((const void GAS*)(Bb + (ob[h][1] + (unsigned)kB)), (LAS void*)(dst + 8192), 16, 0, P8_AUX_B);
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
    const int offA0 = (16 * MI0 * wr + r16) * 128 + ((kc ^ sw) << 4);     // + mi * 2048; ^ 64 for the second k-step
    const int offA1 = (16 * MI1 * wr + r16) * 128 + ((kc ^ sw) << 4);
    const int offB = AH0 + (32 * wc + r16) * 128 + ((kc ^ sw) << 4);      // + ni * 2048
    u16x8 fa[MI][2], fb[2][2][2];
    auto read_a = [&](int T, int a) {
        const char* base = sm + (T & 1) * KT + (a ? AH0 + 2 * BH : 0);
        const int offA = a ? offA1 : offA0;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if (mi >= (a ? MI1 : MI0)) continue;
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
            for (int mi = 0; mi < MI; ++mi) {
                if (mi >= (a ? MI1 : MI0)) continue;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)       // transposed product: lane = output row, registers = 4 columns
                    acc[a][b][mi][ni] = mfma16x16<CT>(fb[b][ni][k2], fa[mi][k2], acc[a][b][mi][ni]);
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // everything up to half-tile `needed` has landed (this thread's pieces); `last` = newest half-tile issued so far
    auto wait_landed = [&](int last, int needed) {
#ifndef P8_OLD_WAIT
        // steady state (both call sites: `last` = a B1 half-tile, three half-tiles B0 / A0 / B1 stay in flight): ONE scalar compare
        // in front of the counted wait -- the k loop is sensitive to every scalar branch here, all eight waves pass it between two
        // barriers: the round-5 selection chain cost 180 of a k-tile's 2850 cycles, a nine-way chain of exact counts 380
        // (tools/p8_replay, profiles/r06m_*); peeling the last k-tiles off into their own loop body bought nothing more
        if (last <= nq - 1) { wait_vm<NPA0 + 4>(); return; }
#endif
        last = min(last, nq - 1);
        const int keep = last - needed;               // half-tiles that may stay in flight (3 in steady state)
        int n = 0;                                    // DMA instructions of this wave that may stay in flight
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if (e < keep && last - e >= 0) { const int j = (last - e) & 3; n += j == 1 ? NPA0 : j == 3 ? NPA1 : 2; }
#ifdef P8_OLD_WAIT
        constexpr int NLO = NPA0 < NPA1 ? NPA0 : NPA1;
        if (n >= 2 * NPA0 + 2) wait_vm<2 * NPA0 + 2>();
        else if (n >= NPA0 + 4 && NPA0 + 4 < 2 * NPA0 + 2) wait_vm<NPA0 + 4>();
        else if (n >= NLO + 2) wait_vm<NLO + 2>();
        else if (n >= 4 && 4 < NLO + 2) wait_vm<4>();
        else if (n >= NLO && NLO >= 2) wait_vm<(NLO >= 2 ? NLO : 2)>();
        else if (n >= 2) wait_vm<2>();
        else wait_vm<0>();
        return;
#endif
        // exactly n may stay in flight (n is wave-uniform: scalar compares; at most 2 NPA0 + 2 <= 8)
        if (n >= 8) wait_vm<8>();
        else if (n == 7) wait_vm<7>();
        else if (n == 6) wait_vm<6>();
        else if (n == 5) wait_vm<5>();
        else if (n == 4) wait_vm<4>();
        else if (n == 3) wait_vm<3>();
        else if (n == 2) wait_vm<2>();
        else if (n == 1) wait_vm<1>();
        else wait_vm<0>();
    };

#ifdef GHN3_P8_PROBE
    const long long pr_t0 = P8_CLK();
    long long pr_wait = 0;
#endif
#pragma unroll
    for (int q = 0; q < D; ++q) issue(q & 3);
    wait_landed(D - 1, 3);                            // k-tile 0 (half-tiles 0..3)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();        // wave row 1 runs one barrier behind
#ifdef GHN3_P8_PROBE
    const long long pr_t1 = P8_CLK();
#endif

    for (int T = 0; T < nkt; ++T) {
        const int g = 4 * T;
        read_b(T, 0);                                 // (first: retired by the counted lgkmcnt below)
        __builtin_amdgcn_sched_barrier(0);
        read_a(T, 0);
        __builtin_amdgcn_sched_barrier(0);
        issue(3);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * MI0) : "memory");  // the B0 reads are done: its buffer is re-staged next phase
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
#ifdef GHN3_P8_PROBE
        const long long pr_w0 = P8_CLK();
#endif
        wait_landed(g + 3 + D, g + 7);                // k-tile T + 1 complete
#ifdef GHN3_P8_PROBE
        pr_wait += P8_CLK() - pr_w0;
#endif
        __builtin_amdgcn_s_barrier();
        mfma_q(1, 0);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();        // pairs the extra barrier of wave row 1
#ifdef GHN3_P8_PROBE
    const long long pr_t2 = P8_CLK();
#endif

    // ---- epilogue (round 5b: staged).  In the accumulator layout -- lane = output row r16 of a 16-row MFMA tile, its 4 registers =
    // columns 4 kc .. 4 kc + 3 -- a 16-lane pass of a store instruction is 16 ROWS x 16 bytes and the CU's store path takes 72 cycles
    // per instruction: 18.6k cycles per 256 x 256 tile (tools/store_probe, per workgroup), 13 % of a forward tile.  With 16 adjacent
    // lanes on the 256 bytes of one row it takes 16.  The ring is free now (every fragment read retired in front of the last
    // barriers), so each wave re-lays 32 x 64 blocks of its part out through its own 8 KB of it: ds_write_b128 in the accumulator
    // layout, ds_read_b128 as 4 rows x 256 bytes, 16-byte chunks XOR-swizzled by the row (conflict-free both ways); a wave's LDS
    // operations execute in order: no barrier.  Everything element-wise happens behind the re-layout, where loads of aux_in /
    // residual / C and the store of aux_out have the same full-line pattern.  Order: acc * alpha -> + bias -> aux_out -> ReLU ->
    // dReLU(aux_in) -> + residual -> (+ C) -> store.  N % 4 == 0: a lane's four columns are all inside or all outside.
    {
        const int flags = rfl(P->flags);
        gf C = (gf)rflp(P->C);
        gcf residual = (gcf)rflp(P->residual);
        gcf aux_in = (gcf)rflp(P->aux_in);
        gf aux_out = (gf)rflp(P->aux_out);
        gcf bias = (flags & GHN3_GEMM_BIASGRAD) ? nullptr : (gcf)rflp(P->bias);
        const int ldc = rfl(P->ldc), cq = rfl(P->c_q), cs = rfl(P->c_s);
        const int bias_q = rfl(P->bias_q), bias_s = rfl(P->bias_s), bias_stride = rfl(P->bias_stride);
        const bool act_relu = rfl(P->act) == GHN3_ACT_RELU, dact_relu = rfl(P->dact) == GHN3_DACT_RELU;
        const bool accum = (flags & GHN3_GEMM_ACCUM) != 0;
        const float* amax_p = rflp(P->alpha_amax);
        const float alpha = rflf(amax_p ? P->alpha * ghn3_pow2_inv_scale(*amax_p) : P->alpha);
        // row map (r / cq) * cs + r % cq = r + (r / cq) (cs - cq), the quotient from one multiply-high (M cq < 2^32: runtime check)
        const unsigned magic = (unsigned)rfl((int)(cq > 0 ? (unsigned)(4294967296.0 / (double)cq) + 1u : 0u));
        const int wrap = cq > 0 ? cs - cq : 0;
        char* stg = sm + wave * 8192;
        const int wofs = r16 * 256, wsw = r16 & 7;
        const int rr0 = lane >> 4, rc = lane & 15;                    // staged image: this lane reads rows rr0 + 4 pass, chunk rc
        const int col = n0 + wc * 64 + 4 * rc;
        const bool col_ok = col < N;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias && col_ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int bi = col + e;
                if (bias_q > 0) bi = (bi / bias_q) * bias_s + (bi % bias_q);
                bv[e] = bias[(int64_t)bi * bias_stride];
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int mp = 0; mp < MI; mp += 2) {
                const int mia = a ? MI1 : MI0;
                if (mp >= mia) continue;
                const int nu = (mp + 1 < mia) ? 2 : 1;                // blocks of 16 rows in this round trip (compile-time after unrolling)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (u >= nu) continue;
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            *reinterpret_cast<f32x4*>(stg + u * 4096 + wofs + (((8 * b + 4 * ni + kc) ^ wsw) << 4)) = acc[a][b][mp + u][ni] * alpha;
                }
                const int row0 = m0 + wr * (BM / 2) + 16 * (a * MI0 + mp);
#pragma unroll
                for (int ps = 0; ps < 8; ++ps) {
                    if (ps >= 4 * nu) continue;
                    const int rl = rr0 + 4 * ps;                      // row of the staged image (0 .. 16 nu - 1)
                    f32x4 v = *reinterpret_cast<const f32x4*>(stg + rl * 256 + ((rc ^ (rl & 7)) << 4));
                    const int row = row0 + rl;
                    if (row < M && col_ok) {
                        const int mrow = row + (int)__umulhi((unsigned)row, magic) * wrap;
                        const int64_t ci = (int64_t)mrow * ldc + col;
                        v += bv;
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
#ifdef P8_NT_STORE
                        __builtin_nontemporal_store(v, reinterpret_cast<gf4>(C + ci));   // (experiment: tools/p8_replay -DP8_NT_STORE)
#else
                        *reinterpret_cast<gf4>(C + ci) = v;
#endif
                    }
                }
            }
    }
#ifdef GHN3_P8_PROBE
    if (tid == 0) {
        const long long pr_t3 = P8_CLK();
        atomicAdd(&g_p8_probe[0], 1ull);
        atomicAdd(&g_p8_probe[1], (unsigned long long)(pr_t1 - pr_t0));
        atomicAdd(&g_p8_probe[2], (unsigned long long)(pr_t2 - pr_t1));
        atomicAdd(&g_p8_probe[3], (unsigned long long)pr_wait);
        atomicAdd(&g_p8_probe[4], (unsigned long long)(pr_t3 - pr_t2));
        atomicAdd(&g_p8_probe[5], (unsigned long long)nkt);
    }
#endif
}

// Row tiles.  With a row-tile table (GemmProbDev::mtab: int32 triples {m0, MI, extent}, written by the host for the
// stacked decoder families) tile mt covers rows [m0, m0 + 64 MI) and carries its own ragged extent; without one the
// problem is cut into 256-row tiles and the extent comes from the per-128-row `lim` array as in the older kernels.
template <int CT>
__device__ __forceinline__ void p8_dispatch(const GemmProbDev* __restrict__ probs, int n_probs, int tile_id, char* sm) {
    int idx, mt, nt;
    const int pin_end = rfl(probs[0].pin_end);
    if (tile_id < pin_end) {
        // XCD-pinned problems: id -> (XCD, local index); the problems of an XCD are found through the directory in entry x
        const int x = tile_id & 7, local = tile_id >> 3;
        const int cnt = rfl(probs[x].pin_count);
        if (cnt == 0) return;
        int lo = rfl(probs[x].pin_first), hi = lo + cnt - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (rfl(probs[mid].tile_start) <= local) lo = mid; else hi = mid - 1;
        }
        idx = lo;
        const int t = local - rfl(probs[idx].tile_start);
        const int tm = rfl(probs[idx].tiles_m);
        if (t >= tm * rfl(probs[idx].tiles_n)) return;
        // row tile by row tile (a chunk's <= 36 tiles all start at once on its XCD, so they still share B rows in L2): the
        // rows are sorted by decreasing extent, i.e. the long tiles are dispatched first and the short ones (a third of the
        // reduction for the narrow groups of a family) fill the remaining CUs instead of delaying a long tile
        const int tn = rfl(probs[idx].tiles_n);
        mt = t / tn;
        nt = t % tn;
    } else {
        int lo = rfl(probs[0].pin_total), hi = n_probs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (rfl(probs[mid].tile_start) <= tile_id) lo = mid; else hi = mid - 1;
        }
        idx = lo;
        // the row tiles of one column tile (same streamed B panel) are congruent mod 8: same XCD, same L2
        const int t = tile_id - rfl(probs[idx].tile_start), grp = t >> 3;
        const int tm = rfl(probs[idx].tiles_m);
        nt = (grp / tm) * 8 + (t & 7);
        mt = grp % tm;
        if (nt >= rfl(probs[idx].tiles_n)) return;
    }
    const GemmProbDev* P = probs + idx;
    const int M = rfl(P->M);
    const int* mtab = rflp(P->mtab);
    int m0, mi, ext;
    if (mtab) {
        m0 = rfl(mtab[3 * mt]); mi = rfl(mtab[3 * mt + 1]); ext = rfl(mtab[3 * mt + 2]);
    } else {
        m0 = mt * 256; mi = 8; ext = 0x7fffffff;
        const int* lim = rflp(P->lim);
        if (lim) {
            ext = rfl(lim[m0 >> 7]);
            if (m0 + 128 < M) ext = max(ext, rfl(lim[(m0 >> 7) + 1]));
        }
    }
    if (m0 >= M) return;
    const int n0 = nt * 256;
    int K = rfl(P->K);
    const int lim_kind = rfl(P->lim_kind);
    if (lim_kind == 1) { if (n0 >= ext) return; }
    else if (lim_kind == 2) K = min(K, ext);
    // height code = rows / 32 (ABI v19): 6 .. 10 = 192 .. 320 rows (7 = 224 and 9 = 288 are new in round 6), 2 / 4 = 64 / 128
    // rows: the skinny row tiles of an inference forward -- a family of <= 128 decoder rows streams its W2 rows once, at the
    // rate the LDS-DMA ring pulls them, instead of through the 128 x 128 two-stage kernel
    if (mi == 2) p8_tile<CT, 1, 1>(P, m0, n0, K, sm);
    else if (mi == 4) p8_tile<CT, 2, 2>(P, m0, n0, K, sm);
    else if (mi == 10) p8_tile<CT, 5, 5>(P, m0, n0, K, sm);         // (160 accumulators + 72 fragment registers: fits 256 only because
    else if (mi == 9) p8_tile<CT, 5, 4>(P, m0, n0, K, sm);          //  every table value lives in SGPRs; spills stay outside the k loop)
    else if (mi == 7) p8_tile<CT, 4, 3>(P, m0, n0, K, sm);
    else if (mi == 6) p8_tile<CT, 3, 3>(P, m0, n0, K, sm);
    else p8_tile<CT, 4, 4>(P, m0, n0, K, sm);
}

template <int CT>
__global__ __launch_bounds__(512, 2) void gemm_p8_kernel(const GemmProbDev* __restrict__ probs, int n_probs, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        p8_dispatch<CT>(probs, n_probs, tile, reinterpret_cast<char*>(smem));
        __syncthreads();                               // the ring is reused by the next tile
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Tile code 29: the 8-phase loop as a PERSISTENT STREAM for output-heavy plain problems -- the decoder's W2 weight gradient
// dW2 [o i x 8C] = d_tiles^T u (autograd of nn.py:747-749): K = the family's rows (9..23 k-tiles) and 256 KB of fp32 output
// per tile, so a tile's prologue (first DMA round trip) and its stores weigh as much as its k loop.  Here
//   * a workgroup walks its tiles (XCD-blocked order of tile code 25) and the half-tile DMA stream simply CONTINUES across
//     tile boundaries: the issue side runs 7 half-tiles ahead of the compute side, switches to the next tile's operand
//     rows when it has issued the last k-tile of the current one (the offsets are recomputed there, no second set of
//     registers), and the LDS ring never drains between tiles;
//   * the stores of a finished tile are DEFERRED into the first k-tile of the next one: phase p of that k-tile stores
//     quadrant p of the old accumulators from its load segment -- i.e. while the partner wave of the SIMD multiplies --
//     right before the quadrant's first MFMA of the new tile overwrites it (that MFMA takes C = 0 as an inline constant:
//     no re-zeroing pass).  Stores and DMA share vmcnt; loads retire in order among themselves, so a counted wait with
//     stores in between only ever over-waits.
// Contract as tile code 25 (checked by the host): C = alpha * A B^T with an optional row map of C; no bias / activation /
// residual / accumulate / gathers / k-map / ragged extents / split-K; N % 4 == 0, K >= 1; 256 x 256 tiles.
// ---------------------------------------------------------------------------------------------------------------------
#ifdef GHN3_P8W_PROBE
__device__ long long g_p8w_probe[48];
#endif
template <int CT, int QUIET>
__global__ __launch_bounds__(512, 2) void gemm_p8w_kernel(const GemmProbDev* __restrict__ probs, int n_probs, int total_tiles_all,
                                                           int vgrid, int tpw) {
    constexpr int MI = 4, BK = 64;
    constexpr int AH = 32 * MI * 128, BH = 128 * 128, KT = 2 * AH + 2 * BH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sm = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = rfl(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r16 = lane & 15, kc = lane >> 4;
    // Virtual persistent grid: worker v of `vgrid` owns the tile ids v, v + vgrid, v + 2 vgrid ... (vgrid % 8 == 0: a worker's
    // tiles keep their XCD).  tpw == 0: one workgroup per worker walks all of them.  tpw > 0: workgroup (chunk c, worker v)
    // walks tpw of them and exits -- the launch then has vgrid * chunks workgroups that the dispatcher starts in order as CUs
    // free up, so a higher-priority stream (the dependent chain this weight gradient runs beside) gets CUs every few tiles
    // instead of never, at the price of one un-overlapped prologue / store phase per tpw tiles.
    const bool nt_store = rfl(tpw >> 30) != 0;      // (GHN3_WGRAD_NT=1: streaming stores of the 1.8 GB output; experiment)
    tpw &= 0x3fffffff;
    const int stride = vgrid;
    const int w_v = (int)blockIdx.x % vgrid, w_c = (int)blockIdx.x / vgrid;
    const int t_first = w_v + vgrid * w_c * tpw;
    const int total_tiles = tpw > 0 ? min(total_tiles_all, t_first + vgrid * tpw) : total_tiles_all;

    // first valid tile at or behind id t (XCD-blocked order: see gemm_h16w_kernel / runtime.hip) -> problem index, origin, k-tiles
    auto next_tile = [&](int t, int& idx, int& m0, int& n0, int& nkt) -> int {
        for (; t < total_tiles; t += stride) {
            int lo = 0, hi = n_probs - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (rfl(probs[mid].tile_start) <= t) lo = mid; else hi = mid - 1;
            }
            const GemmProbDev* Q = probs + lo;
            const int tl = t - rfl(Q->tile_start), x = tl & 7, grp = tl >> 3;
            const int G = rfl(Q->xcd_cols), Gm = 8 / G;
            const int tn = rfl(Q->tiles_n), tm = rfl(Q->tiles_m);
            const int npg = (tn + G - 1) / G;
            const int nt = (x % G) * npg + grp % npg, mt = (grp / npg) * Gm + x / G;
            if (nt >= tn || mt >= tm) continue;
            idx = lo; m0 = mt * 256; n0 = nt * 256; nkt = (rfl(Q->K) + BK - 1) / BK;
            return t;
        }
        return total_tiles;
    };

    // ---- issue side -------------------------------------------------------------------------------------------------
    int i_idx = 0, i_m0 = 0, i_n0 = 0, i_nkt = 0;
    int it = next_tile(t_first, i_idx, i_m0, i_n0, i_nkt);
    if (it >= total_tiles) return;                    // (uniform over the workgroup)
    // ---- compute side starts at the same tile
    int c_idx = i_idx, c_m0 = i_m0, c_n0 = i_n0, c_nkt = i_nkt, ct = it;

    unsigned oa[2][2], ob[2][2];
    const char GAS* Ab = nullptr;
    const char GAS* Bb = nullptr;
    const int slot = tid & 7, rb = tid >> 3;
    const unsigned ck2 = (unsigned)((slot ^ ((rb >> 1) & 7)) * 16);
    auto load_issue_tile = [&]() {
        const GemmProbDev* Q = probs + i_idx;
        Ab = (const char GAS*)rflp(Q->A);
        Bb = (const char GAS*)rflp(Q->B);
        const int M = rfl(Q->M), N = rfl(Q->N);
        const unsigned lda2 = (unsigned)rfl(Q->lda) * 2u, ldb2 = (unsigned)rfl(Q->ldb) * 2u;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rho = rb + 64 * i;
                const int ra = min(i_m0 + 128 * (rho >> 6) + 64 * h + (rho & 63), M - 1);
                const int rn = min(i_n0 + 64 * (rho >> 5) + 32 * h + (rho & 31), N - 1);
                oa[h][i] = (unsigned)ra * lda2 + ck2;
                ob[h][i] = (unsigned)rn * ldb2 + ck2;
            }
    };
    load_issue_tile();
    bool i_live = true;
    int kA = 0, i_left = i_nkt, par = 0, issued = 0;  // k byte offset, k-tiles left to issue, ring parity offset, half-tiles issued
    auto issue = [&](const int j) {                   // j = (next half-tile) & 3: 0 = B0, 1 = A0, 2 = B1, 3 = A1
        if (!i_live) return;
        char* kt = sm + par;
        if (j == 1 || j == 3) {
            const int h = j == 3;
            LAS char* dst = (LAS char*)(kt + (h ? AH + 2 * BH : 0) + wave * 1024);
            __builtin_amdgcn_global_load_lds((const void GAS*)(Ab + (oa[h][0] + (unsigned)kA)), (LAS void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void GAS*)(Ab + (oa[h][1] + (unsigned)kA)), (LAS void*)(dst + 8192), 16, 0, 0);
        } else {
            const int h = j == 2;
            LAS char* dst = (LAS char*)(kt + AH + h * BH + wave * 1024);
            __builtin_amdgcn_global_load_lds((const void GAS*)(Bb + (ob[h][0] + (unsigned)kA)), (LAS void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void GAS*)(Bb + (ob[h][1] + (unsigned)kA)), (LAS void*)(dst + 8192), 16, 0, 0);
        }
        ++issued;
        if (j == 3) {
            kA += 128; par = KT - par;
            if (--i_left == 0) {                      // the issue side moves on to the workgroup's next tile
                it = next_tile(it + stride, i_idx, i_m0, i_n0, i_nkt);
                i_live = it < total_tiles;
                if (i_live) { load_issue_tile(); kA = 0; i_left = i_nkt; }
            }
        }
    };

    // ---- compute side -----------------------------------------------------------------------------------------------
    f32x4 acc[2][2][MI][2];
    const int sw = r16 >> 1;
    const int offA = (16 * MI * wr + r16) * 128 + ((kc ^ sw) << 4);
    const int offB = AH + (32 * wc + r16) * 128 + ((kc ^ sw) << 4);
    u16x8 fa[MI][2], fb[2][2][2];
    auto read_a = [&](const char* ring, int a) {
        const char* base = ring + (a ? AH + 2 * BH : 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            fa[mi][0] = *reinterpret_cast<const u16x8*>(base + offA + mi * 2048);
            fa[mi][1] = *reinterpret_cast<const u16x8*>(base + (offA ^ 64) + mi * 2048);
        }
    };
    auto read_b = [&](const char* ring, int b) {
        const char* base = ring + b * BH;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            fb[b][ni][0] = *reinterpret_cast<const u16x8*>(base + offB + ni * 2048);
            fb[b][ni][1] = *reinterpret_cast<const u16x8*>(base + (offB ^ 64) + ni * 2048);
        }
    };
    auto mfma_q = [&](int a, int b, const bool first) {   // first: the quadrant starts from zero (inline constant C)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[a][b][mi][ni] = mfma16x16<CT>(fb[b][ni][k2], fa[mi][k2], (first && k2 == 0) ? z : acc[a][b][mi][ni]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // previous tile of this workgroup, whose accumulators are still in the registers
    bool p_valid = false;
    gf pC = nullptr;
    int p_ldc = 0, p_cq = 0, p_cs = 0, p_M = 0, p_N = 0, p_m0 = 0, p_n0 = 0;
    float p_alpha = 1.f;
    // GHN3_GEMM_SUMSQ: sum of the squares of what this wave stores of a tile -> slot [8 * tile id + wave] (the squared gradient
    // norm of clip_grad_norm_ then needs no pass over the 1.8 GB of dW2; fixed order inside a wave, one writer per slot)
    float* p_sq = nullptr;
    int p_tile = 0;
    float ss = 0.f;                                   // (one add per 16 x 32 block: the squares of a block are summed as a tree first)
    // Round 5b: the epilogue's own instructions were half of the k-tile that stores (tools/p8w_probe -DP8W_X=2: 15k of its 25k cycles
    // WITHOUT any store instruction -- per store a row-map division with two fix-up branches, bounds branches and a dependent
    // chain of four FMAs, on SIMDs that two waves share).  Now: the row map (r / q) * s + r % q = r + (r / q) (s - q) takes its
    // quotient from one v_mul_hi_u32 with the per-tile magic number floor(2^32 / q) + 1 (exact while M q < 2^32, which the runtime checks; 0 = no map),
    // interior tiles skip every bounds test, and the squares of a block are summed as a tree before they join the lane's sum.
    unsigned p_magic = 0;
    int p_wrap = 0, p_colq = 0;
    bool p_inner = false;
    // Staged stores (round 5b).  In the accumulator layout a 16-lane pass of a store instruction is 16 ROWS x 16 bytes: the CU's
    // store path takes 72 cycles per such instruction -- 18.6k cycles per 256 KB tile, whoever else runs (tools/store_probe, timed
    // per workgroup) -- but 16 cycles when eight adjacent lanes write the 128 bytes of one cache line (4.2k per tile).  So every
    // 16 x 32 block (two accumulator registers) goes through a wave-private 2 KB image behind the ring (two images per wave) -- ds_write_b128 in the
    // accumulator layout, ds_read_b128 as 8 rows x 128 bytes, 16-byte chunks XOR-swizzled by the row: conflict-free both ways --
    // and leaves as two instructions of eight full lines each.  No barrier: a wave's LDS operations execute in order.
    auto store_prev_t = [&](const int a, const int b, const bool inner) __attribute__((always_inline)) {
        char* stg = sm + 2 * KT + wave * 4096;        // two 16 x 32 fp32 images per wave: two blocks per LDS round trip
        const int wofs = r16 * 128, wsw = r16 & 7;
        const int rr0 = lane >> 3, rc = lane & 7;
        const int rofs = rr0 * 128 + ((rc ^ rr0) << 4);
        const int col = p_colq + b * 32 + 4 * rc;     // this lane's four columns of the staged rows
        const bool col_ok = inner || col < p_N;
#pragma unroll
        for (int mp = 0; mp < MI; mp += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int mi = mp + u;
                const int row0 = p_m0 + wr * 128 + 16 * (a * MI + mi);
                const f32x4 x = acc[a][b][mi][0] * p_alpha, y = acc[a][b][mi][1] * p_alpha;
                {
                    const f32x4 xx = x * x, yy = y * y;
                    const float tx = (xx[0] + xx[1]) + (xx[2] + xx[3]), ty = (yy[0] + yy[1]) + (yy[2] + yy[3]);
                    if (inner) ss += tx + ty;
                    else if (row0 + r16 < p_M) {      // (rows / columns beyond the problem hold copies of the last row / column)
                        const int c0 = p_colq + 4 * kc + b * 32;
                        ss += (c0 < p_N ? tx : 0.f) + (c0 + 16 < p_N ? ty : 0.f);
                    }
                }
#if defined(GHN3_P8W_PROBE) && defined(P8W_X) && P8W_X == 8
                asm volatile("" :: "v"(x), "v"(y));
#else
                *reinterpret_cast<f32x4*>(stg + 2048 * u + wofs + ((kc ^ wsw) << 4)) = x;
                *reinterpret_cast<f32x4*>(stg + 2048 * u + wofs + (((4 + kc) ^ wsw) << 4)) = y;
#endif
            }
            f32x4 v[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h) v[u][h] = *reinterpret_cast<const f32x4*>(stg + 2048 * u + rofs + 1024 * h);
#if defined(GHN3_P8W_PROBE) && defined(P8W_X) && P8W_X == 8
            // (tools/p8w_probe: 8 = the accumulators themselves in the staged ADDRESS pattern -- wrong data, no LDS round trip)
            v[0][0] = acc[a][b][mp][0]; v[0][1] = acc[a][b][mp][1]; v[1][0] = acc[a][b][mp + 1][0]; v[1][1] = acc[a][b][mp + 1][1];
#endif
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = p_m0 + wr * 128 + 16 * (a * MI + mp + u) + rr0 + 8 * h;
                    if (inner || (col_ok && row < p_M)) {
                        const int mrow = row + (int)__umulhi((unsigned)row, p_magic) * p_wrap;
                        float GAS* dst = pC + ((int64_t)mrow * p_ldc + col);
#if defined(GHN3_P8W_PROBE) && defined(P8W_X)
                        if (P8W_X == 2) { asm volatile("" :: "v"(v[u][h])); } else     // (tools/p8w_probe: 2 = no store instruction)
#endif
                        if (nt_store) __builtin_nontemporal_store(v[u][h], reinterpret_cast<gf4>(dst));
                        else *reinterpret_cast<gf4>(dst) = v[u][h];
                    }
                }
        }
    };
    auto store_prev = [&](int a, int b) __attribute__((always_inline)) {
        if (p_inner) store_prev_t(a, b, true); else store_prev_t(a, b, false);
    };
    auto finish_sq = [&]() {                          // behind the last quadrant's stores of a tile
        float t = ss;
        ss = 0.f;
        if (!p_sq) return;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (lane == 0) p_sq[p_tile * 8 + wave] = t;
    };
#ifdef GHN3_P8W_PROBE
    // tools/p8w_probe.hip: cycles of workgroup 0 / thread 0 per k-tile (first k-tile of a tile = the one that also stores the
    // previous tile), inside the DMA wait of a k-tile, and -- store k-tiles -- inside store_prev / at the barriers
    long long pr_tot[2] = {0, 0}, pr_wait[2] = {0, 0}, pr_cnt[2] = {0, 0}, pr_store = 0, pr_bar = 0;
    long long pr_ph[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pr_last = 0;
#define P8W_MARK(i) { if (st) { const long long t_ = P8W_CLK(); pr_ph[i] += t_ - pr_last; pr_last = t_; } }
#define P8W_CLK() ((long long)__builtin_readcyclecounter())
#define P8W_ST_T(x) { const long long t_ = P8W_CLK(); x; pr_store += P8W_CLK() - t_; }
#define P8W_BAR_T() { const long long t_ = P8W_CLK(); __builtin_amdgcn_s_barrier(); if (st) pr_bar += P8W_CLK() - t_; }
#else
#define P8W_ST_T(x) x
#define P8W_BAR_T() __builtin_amdgcn_s_barrier()
#define P8W_MARK(i)
#endif
    int G = 0;                                        // k-tiles computed so far (ring parity, half-tile bookkeeping)
    auto wait_next_ktile = [&]() {                    // k-tile G + 1 (half-tiles .. 4 G + 7) has landed; the newer ones stay in flight
        const int keep = issued - 1 - (4 * G + 7);
        if (keep >= 3) wait_vm<6>();
        else if (keep == 2) wait_vm<4>();
        else if (keep == 1) wait_vm<2>();
        else wait_vm<0>();
    };
    auto ktile = [&](const bool first) {
#ifdef GHN3_P8W_PROBE
        const long long pr_t0 = P8W_CLK();
#endif
        const char* ring = sm + (G & 1) * KT;
        const bool st = first && p_valid;
        // QUIET store k-tile (round 5, default; GHN3_P8W_QUIET=0 turns it off): stores and LDS-DMA share the wave's vmcnt, so a counted
        // wait behind stores also waits for their write acknowledgements -- in the k-tile that carries a tile's 256 KB of
        // stores that wait (and the store instructions queueing behind DMA requests) cost most of its 21-27k cycles.  Quiet:
        // the k-tile first lets everything in flight land (one exposed round trip for the newest half-tile), issues NO DMA
        // while it stores, and issues its three half-tiles at its very end; the first counted wait that sees the stores
        // again is the next k-tile's, a whole k-tile later.
        const bool quiet = QUIET && st;
#ifdef GHN3_P8W_PROBE
        if (st) pr_last = P8W_CLK();
#endif
        if (!quiet) {
            if (st) P8W_ST_T(store_prev(0, 0));
            read_b(ring, 0);                          // (first: retired by the counted lgkmcnt below)
            __builtin_amdgcn_sched_barrier(0);
            read_a(ring, 0);
            __builtin_amdgcn_sched_barrier(0);
            issue(3);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * MI) : "memory");
            P8W_BAR_T();
        } else {
            // Round 5b: ALL of the tile's stores in one burst, both wave rows at once (spread over the four phases only one row
            // stored at a time: 8 x ~2k cycles per tile; tools/p8w_probe phase marks).  Wave row 1 runs one barrier behind row 0, so
            // row 0 first lets it catch up (one extra barrier), both rows store -- in front of their LDS-DMA: a DMA instruction issued
            // while stores drain waits for them -- and row 1 then falls back by one barrier again.
            if (wr == 0) P8W_BAR_T();
            issue(3);
            P8W_MARK(0);
            wait_vm<0>();                             // quiet: everything in flight lands; no DMA is issued until the k-tile's end
            P8W_MARK(1);
            P8W_ST_T(store_prev(0, 0); store_prev(0, 1); store_prev(1, 1); store_prev(1, 0); finish_sq());
            P8W_MARK(2);
            if (wr == 1) P8W_BAR_T();
            // (the fragment reads behind the burst: no fragment register is live across it; nothing re-stages this k-tile's
            // half-tiles before the issues at its end)
            read_b(ring, 0);
            __builtin_amdgcn_sched_barrier(0);
            read_a(ring, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * MI) : "memory");
            P8W_BAR_T();
            P8W_MARK(3);
        }
        mfma_q(0, 0, first);
        P8W_MARK(4);
        P8W_BAR_T();
        P8W_MARK(5);
        if (st && !quiet) P8W_ST_T(store_prev(0, 1));
        P8W_MARK(6);
        read_b(ring, 1);
        if (!quiet) issue(0);
        P8W_BAR_T();
        P8W_MARK(7);
        mfma_q(0, 1, first);
        P8W_MARK(8);
        P8W_BAR_T();
        P8W_MARK(9);
        if (st && !quiet) P8W_ST_T(store_prev(1, 1));
        P8W_MARK(10);
        read_a(ring, 1);
        if (!quiet) issue(1);
        P8W_BAR_T();
        P8W_MARK(11);
        mfma_q(1, 1, first);
        P8W_MARK(12);
        P8W_BAR_T();
        P8W_MARK(13);
#ifdef GHN3_P8W_PROBE
        const long long pr_w0 = P8W_CLK();
#endif
        if (!quiet) { issue(2); wait_next_ktile(); }
#ifdef GHN3_P8W_PROBE
        pr_wait[st ? 0 : 1] += P8W_CLK() - pr_w0;
#endif
        if (st && !quiet) { P8W_ST_T(store_prev(1, 0)); finish_sq(); }    // (behind the wait: these stores are not waited for with the DMA)
        P8W_MARK(14);
        if (quiet) { issue(0); issue(1); issue(2); }  // (k-tile G + 1 had landed before the stores; these are for G + 2)
        P8W_BAR_T();
        P8W_MARK(15);
        mfma_q(1, 0, first);
        P8W_MARK(16);
        P8W_BAR_T();
        P8W_MARK(17);
        ++G;
#ifdef GHN3_P8W_PROBE
        pr_tot[st ? 0 : 1] += P8W_CLK() - pr_t0; pr_cnt[st ? 0 : 1] += 1;
#endif
    };

#pragma unroll
    for (int q = 0; q < 7; ++q) issue(q & 3);
    {                                                 // k-tile 0 (half-tiles 0..3)
        const int keep = issued - 1 - 3;
        if (keep >= 3) wait_vm<6>(); else if (keep == 2) wait_vm<4>(); else if (keep == 1) wait_vm<2>(); else wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();        // wave row 1 runs one barrier behind

    while (ct < total_tiles) {
        ktile(true);
        for (int kt = 1; kt < c_nkt; ++kt) ktile(false);
        {                                             // this tile's accumulators are stored during the next tile's first k-tile
            const GemmProbDev* Q = probs + c_idx;
            pC = (gf)rflp(Q->C);
            p_ldc = rfl(Q->ldc); p_cq = rfl(Q->c_q); p_cs = rfl(Q->c_s); p_M = rfl(Q->M); p_N = rfl(Q->N);
            p_m0 = c_m0; p_n0 = c_n0;
            const float* amax_p = rflp(Q->alpha_amax);
            p_alpha = rflf(amax_p ? Q->alpha * ghn3_pow2_inv_scale(*amax_p) : Q->alpha);
            p_sq = (rfl(Q->flags) & GHN3_GEMM_SUMSQ) ? rflp(Q->aux_out) : nullptr;
            p_tile = ct;
            p_valid = true;
            // (see store_prev_t; all wave-uniform: kept in SGPRs)
            p_colq = rfl(p_n0 + wc * 64);
            p_inner = p_m0 + 256 <= p_M && p_n0 + 256 <= p_N;
            p_magic = (unsigned)rfl((int)(p_cq > 0 ? (unsigned)(4294967296.0 / (double)p_cq) + 1u : 0u));   // (M q < 2^32: checked by the runtime)
            p_wrap = p_cq > 0 ? p_cs - p_cq : 0;
        }
        ct = next_tile(ct + stride, c_idx, c_m0, c_n0, c_nkt);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();        // pairs the extra barrier of wave row 1
    store_prev(0, 0); store_prev(0, 1); store_prev(1, 1); store_prev(1, 0);
    finish_sq();
#ifdef GHN3_P8W_PROBE
    if (blockIdx.x == 0 && tid == 0) {
        for (int i = 0; i < 2; ++i) { g_p8w_probe[3 * i] = pr_tot[i]; g_p8w_probe[3 * i + 1] = pr_wait[i]; g_p8w_probe[3 * i + 2] = pr_cnt[i]; }
        g_p8w_probe[6] = pr_store; g_p8w_probe[7] = pr_bar;
    }
    if (blockIdx.x == 0 && (tid == 0 || tid == 256))
        for (int i = 0; i < 20; ++i) g_p8w_probe[8 + (tid ? 20 : 0) + i] = pr_ph[i];
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Tile code 30 (round 6): the persistent weight-gradient stream with TWO accumulator sets -- the stores of a finished tile
// leave during the WHOLE next tile instead of in one burst that stops the matrix cores.
//
// Why.  dW2 band problems are output-heavy: K = the family's rows (9 .. 24 k-tiles) against 256 KB of fp32 per 256 x 256 tile.
// In gemm_p8w_kernel a tile's stores are one quiet burst inside the first k-tile of the next tile: 19k cycles against 2.7k for
// a k-tile, and the DMA ring is empty behind it (the next k-tiles wait 500 cycles each): the K = 536 problems -- 44 % of the
// tiles of the bench step -- run at 600 TF, 36 % of their cycles in the burst (tools/p8w_probe, profiles/r06o_*).  The
// accumulators cannot leave earlier (one set) nor faster (a CU's store path + 64 MB per chip-wide burst).
// Here a tile is 256 x 128: 64 accumulator registers per lane, so TWO sets fit the 128 the 256 x 256 tile used.  A finished
// tile's set is copied aside (64 v_mov) and drained one 16 x 32 block per k-tile of the NEXT tile -- through the wave-private
// LDS image, two full-line store instructions per block -- while the matrix cores work on the other set.  Nothing stalls for
// stores; the counted DMA wait of a k-tile sees the two store instructions of the previous k-tile, a whole k-tile old.
//   * 8 waves: wave row wr = wave >> 2 (128 rows), wave column wc = wave & 3 (32 columns); sub-tiles a = 0, 1 (64 rows each).
//   * k-tile (64 k) = TWO phases: Q(a = 0) then Q(a = 1), 16 MFMAs each (4 row tiles x 2 column tiles x 2 k-steps) -- the
//     phase length of the 256 x 256 kernels; the B fragments (32 columns) are read once per k-tile.
//   * LDS: ring of THREE k-tiles x [A0 | B | A1] (16 KB each, 128-byte rows, XOR-swizzled chunks as above) = 144 KB + 2 KB of
//     store staging per wave = 160 KB.  k-tile G issues the three pieces of k-tile G + 2 (B and A0 in phase 0, A1 in phase
//     1) into the buffers k-tile G - 1 was read from a k-tile earlier; its wait leaves exactly these six instructions in flight.
//   * The DMA stream continues across tiles (issue side two k-tiles ahead of the compute side), the first MFMA of a quadrant
//     of a new tile takes C = 0 as an inline constant.
// Contract as tile code 29; tiles are 256 x 128 (the runtime numbers them with 128-column tiles).
// ---------------------------------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(512, 2) void gemm_p8d_kernel(const GemmProbDev* __restrict__ probs, int n_probs, int total_tiles_all,
                                                           int vgrid) {
    constexpr int MI = 4, BK = 64;
    constexpr int PIECE = 128 * 128, KT = 3 * PIECE;            // bytes: one piece (A0 / B / A1), one k-tile of the ring
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sm = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = rfl(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r16 = lane & 15, kc = lane >> 4;
    const int stride = vgrid;
    const int t_first = (int)blockIdx.x;
    const int total_tiles = total_tiles_all;

    // first valid tile at or behind id t (XCD-blocked order as tile code 29, 128-column tiles) -> problem, origin, k-tiles
    auto next_tile = [&](int t, int& idx, int& m0, int& n0, int& nkt) -> int {
        for (; t < total_tiles; t += stride) {
            int lo = 0, hi = n_probs - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (rfl(probs[mid].tile_start) <= t) lo = mid; else hi = mid - 1;
            }
            const GemmProbDev* Q = probs + lo;
            const int tl = t - rfl(Q->tile_start), x = tl & 7, grp = tl >> 3;
            const int G = rfl(Q->xcd_cols), Gm = 8 / G;
            const int tn = rfl(Q->tiles_n), tm = rfl(Q->tiles_m);
            const int npg = (tn + G - 1) / G;
            const int nt = (x % G) * npg + grp % npg, mt = (grp / npg) * Gm + x / G;
            if (nt >= tn || mt >= tm) continue;
            idx = lo; m0 = mt * 256; n0 = nt * 128; nkt = (rfl(Q->K) + BK - 1) / BK;
            return t;
        }
        return total_tiles;
    };

    // ---- issue side -------------------------------------------------------------------------------------------------
    int i_idx = 0, i_m0 = 0, i_n0 = 0, i_nkt = 0;
    int it = next_tile(t_first, i_idx, i_m0, i_n0, i_nkt);
    if (it >= total_tiles) return;                    // (uniform over the workgroup)
    int c_idx = i_idx, c_m0 = i_m0, c_n0 = i_n0, c_nkt = i_nkt, ct = it;

    unsigned oa[2][2], ob[2];
    const char GAS* Ab = nullptr;
    const char GAS* Bb = nullptr;
    const int slot = tid & 7, rb = tid >> 3;
    const unsigned ck2 = (unsigned)((slot ^ ((rb >> 1) & 7)) * 16);
    auto load_issue_tile = [&]() {
        const GemmProbDev* Q = probs + i_idx;
        Ab = (const char GAS*)rflp(Q->A);
        Bb = (const char GAS*)rflp(Q->B);
        const int M = rfl(Q->M), N = rfl(Q->N);
        const unsigned lda2 = (unsigned)rfl(Q->lda) * 2u, ldb2 = (unsigned)rfl(Q->ldb) * 2u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rho = rb + 64 * i;              // buffer row 0 .. 127: wave row rho >> 6, row rho & 63 of its sub-tile
#pragma unroll
            for (int h = 0; h < 2; ++h)
                oa[h][i] = (unsigned)min(i_m0 + 128 * (rho >> 6) + 64 * h + (rho & 63), M - 1) * lda2 + ck2;
            ob[i] = (unsigned)min(i_n0 + rho, N - 1) * ldb2 + ck2;
        }
    };
    load_issue_tile();
    bool i_live = true;
    int kA = 0, i_left = i_nkt, islot = 0, issued = 0;   // k byte offset, k-tiles left to issue, ring slot (bytes), PIECES issued
    auto issue = [&](const int j) {                   // j = piece of the k-tile being issued: 0 = B, 1 = A0, 2 = A1
        if (!i_live) return;
        char* kt = sm + islot;
        if (j == 0) {
            LAS char* dst = (LAS char*)(kt + PIECE + wave * 1024);
            __builtin_amdgcn_global_load_lds((const void GAS*)(Bb + (ob[0] + (unsigned)kA)), (LAS void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void GAS*)(Bb + (ob[1] + (unsigned)kA)), (LAS void*)(dst + 8192), 16, 0, 0);
        } else {
            const int h = j == 2;
            LAS char* dst = (LAS char*)(kt + (h ? 2 * PIECE : 0) + wave * 1024);
            __builtin_amdgcn_global_load_lds((const void GAS*)(Ab + (oa[h][0] + (unsigned)kA)), (LAS void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void GAS*)(Ab + (oa[h][1] + (unsigned)kA)), (LAS void*)(dst + 8192), 16, 0, 0);
        }
        ++issued;
        if (j == 2) {
            kA += 128;
            islot = islot == 2 * KT ? 0 : islot + KT;
            if (--i_left == 0) {                      // the issue side moves on to the workgroup's next tile
                it = next_tile(it + stride, i_idx, i_m0, i_n0, i_nkt);
                i_live = it < total_tiles;
                if (i_live) { load_issue_tile(); kA = 0; i_left = i_nkt; }
            }
        }
    };

    // ---- compute side -----------------------------------------------------------------------------------------------
    f32x4 acc[2][MI][2], prv[2][MI][2];
    const int sw = r16 >> 1;
    const int offA = (64 * wr + r16) * 128 + ((kc ^ sw) << 4);            // + mi * 2048 (inside piece A_a); ^ 64: second k-step
    const int offB = PIECE + (32 * wc + r16) * 128 + ((kc ^ sw) << 4);    // + ni * 2048
    u16x8 fa[MI][2], fb[2][2];
    auto read_a = [&](const char* ring, int a) {
        const char* base = ring + (a ? 2 * PIECE : 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            fa[mi][0] = *reinterpret_cast<const u16x8*>(base + offA + mi * 2048);
            fa[mi][1] = *reinterpret_cast<const u16x8*>(base + (offA ^ 64) + mi * 2048);
        }
    };
    auto read_b = [&](const char* ring) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            fb[ni][0] = *reinterpret_cast<const u16x8*>(ring + offB + ni * 2048);
            fb[ni][1] = *reinterpret_cast<const u16x8*>(ring + (offB ^ 64) + ni * 2048);
        }
    };
    auto mfma_q = [&](int a, const bool first) {      // first: the quadrant starts from zero (inline constant C)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[a][mi][ni] = mfma16x16<CT>(fb[ni][k2], fa[mi][k2], (first && k2 == 0) ? z : acc[a][mi][ni]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // previous tile of this workgroup: its accumulators sit in prv[][][] and leave block by block
    bool p_valid = false;
    gf pC = nullptr;
    int p_ldc = 0, p_M = 0, p_N = 0, p_m0 = 0, p_tile = 0, p_colq = 0, p_wrap = 0;
    unsigned p_magic = 0;
    bool p_inner = false;
    float p_alpha = 1.f;
    float* p_sq = nullptr;
    float ss = 0.f;
    // block q (0 .. 7) = rows 16 (4 a + mi) .. + 15 of the wave row x the wave's 32 columns: through the wave-private 2 KB image
    // (ds_write_b128 in the accumulator layout, ds_read_b128 as 8 rows x 128 bytes, chunks XOR-swizzled by the row: conflict-free
    // both ways; a wave's LDS operations execute in order: no barrier), out as two instructions of eight full lines each
    auto drain_block = [&](const int a, const int mi) __attribute__((always_inline)) {
        char* stg = sm + 3 * KT + wave * 2048;
        const int wofs = r16 * 128, wsw = r16 & 7;
        const int rr0 = lane >> 3, rc = lane & 7;
        const int rofs = rr0 * 128 + ((rc ^ rr0) << 4);
        const int col = p_colq + 4 * rc;
        const int row0 = p_m0 + wr * 128 + 16 * (a * MI + mi);
        const f32x4 x = prv[a][mi][0] * p_alpha, y = prv[a][mi][1] * p_alpha;
        {
            const f32x4 xx = x * x, yy = y * y;
            const float tx = (xx[0] + xx[1]) + (xx[2] + xx[3]), ty = (yy[0] + yy[1]) + (yy[2] + yy[3]);
            if (p_inner) ss += tx + ty;
            else if (row0 + r16 < p_M) {              // (rows / columns beyond the problem hold copies of the last row / column)
                const int c0 = p_colq + 4 * kc;
                ss += (c0 < p_N ? tx : 0.f) + (c0 + 16 < p_N ? ty : 0.f);
            }
        }
        *reinterpret_cast<f32x4*>(stg + wofs + ((kc ^ wsw) << 4)) = x;
        *reinterpret_cast<f32x4*>(stg + wofs + (((4 + kc) ^ wsw) << 4)) = y;
        f32x4 v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) v[h] = *reinterpret_cast<const f32x4*>(stg + rofs + 1024 * h);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = row0 + rr0 + 8 * h;
            if (p_inner || (col < p_N && row < p_M)) {
                const int mrow = row + (int)__umulhi((unsigned)row, p_magic) * p_wrap;
                *reinterpret_cast<gf4>(pC + ((int64_t)mrow * p_ldc + col)) = v[h];
            }
        }
    };
    auto finish_sq = [&]() {                          // behind the last block of a tile
        float t = ss;
        ss = 0.f;
        if (!p_sq) return;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (lane == 0) p_sq[p_tile * 8 + wave] = t;
    };

    int G = 0;                                        // k-tiles computed so far
    int cslot = 0;                                    // ring slot (bytes) of k-tile G
    // k-tile G + 1 (pieces .. 3 G + 5) has landed; the newer pieces stay in flight (two DMA instructions each; store
    // instructions of the previous k-tile are older than all of them)
    auto wait_next_ktile = [&]() {
        const int keep = issued - (3 * G + 6);
        if (keep >= 3) wait_vm<6>();
        else if (keep == 2) wait_vm<4>();
        else if (keep == 1) wait_vm<2>();
        else wait_vm<0>();
    };
    // Q: block of the previous tile this k-tile drains (compile time; < 0: none)
    auto ktile = [&](const bool first, const auto qc) __attribute__((always_inline)) {
        constexpr int Q = decltype(qc)::value;
        const char* ring = sm + cslot;
        read_b(ring);
        __builtin_amdgcn_sched_barrier(0);
        read_a(ring, 0);
        __builtin_amdgcn_sched_barrier(0);
        issue(0);
        issue(1);
        __builtin_amdgcn_s_barrier();
        mfma_q(0, first);
        __builtin_amdgcn_s_barrier();
        read_a(ring, 1);
        __builtin_amdgcn_sched_barrier(0);
        issue(2);
        wait_next_ktile();
        if (Q >= 0) {
            if (p_valid) {
                drain_block(Q >> 2, Q & 3);
                if (Q == 7) { finish_sq(); p_valid = false; }
            }
        }
        __builtin_amdgcn_s_barrier();
        mfma_q(1, first);
        __builtin_amdgcn_s_barrier();
        ++G;
        cslot = cslot == 2 * KT ? 0 : cslot + KT;
    };

    // prologue: k-tiles 0 and 1 of the stream (six pieces); k-tile 0 must have landed
    issue(0); issue(1); issue(2);
    issue(0); issue(1); issue(2);
    {
        const int keep = issued - 3;
        if (keep >= 3) wait_vm<6>(); else if (keep == 2) wait_vm<4>(); else if (keep == 1) wait_vm<2>(); else wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();        // wave row 1 runs one barrier behind

#define P8D_Q(n) P8Int<n>{}
    while (ct < total_tiles) {
        // the first eight k-tiles of a tile drain one block of the previous tile each; shorter tiles drain the rest in a burst
        ktile(true, P8D_Q(0));
        if (c_nkt > 1) ktile(false, P8D_Q(1));
        if (c_nkt > 2) ktile(false, P8D_Q(2));
        if (c_nkt > 3) ktile(false, P8D_Q(3));
        if (c_nkt > 4) ktile(false, P8D_Q(4));
        if (c_nkt > 5) ktile(false, P8D_Q(5));
        if (c_nkt > 6) ktile(false, P8D_Q(6));
        if (c_nkt > 7) ktile(false, P8D_Q(7));
        for (int kt = 8; kt < c_nkt; ++kt) ktile(false, P8D_Q(-1));
        if (p_valid) {                                // (fewer than eight k-tiles: the remaining blocks now)
            if (c_nkt <= 1) drain_block(0, 1);
            if (c_nkt <= 2) drain_block(0, 2);
            if (c_nkt <= 3) drain_block(0, 3);
            if (c_nkt <= 4) drain_block(1, 0);
            if (c_nkt <= 5) drain_block(1, 1);
            if (c_nkt <= 6) drain_block(1, 2);
            drain_block(1, 3);
            finish_sq();
            p_valid = false;
        }
        {                                             // this tile's accumulators move aside and leave during the next tile
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) prv[a][mi][ni] = acc[a][mi][ni];
            const GemmProbDev* Q = probs + c_idx;
            pC = (gf)rflp(Q->C);
            p_ldc = rfl(Q->ldc); p_M = rfl(Q->M); p_N = rfl(Q->N);
            p_m0 = c_m0;
            const int cq = rfl(Q->c_q), cs = rfl(Q->c_s);
            const float* amax_p = rflp(Q->alpha_amax);
            p_alpha = rflf(amax_p ? Q->alpha * ghn3_pow2_inv_scale(*amax_p) : Q->alpha);
            p_sq = (rfl(Q->flags) & GHN3_GEMM_SUMSQ) ? rflp(Q->aux_out) : nullptr;
            p_tile = ct;
            p_valid = true;
            p_colq = rfl(c_n0 + wc * 32);
            p_inner = c_m0 + 256 <= p_M && c_n0 + 128 <= p_N;
            p_magic = (unsigned)rfl((int)(cq > 0 ? (unsigned)(4294967296.0 / (double)cq) + 1u : 0u));   // (M q < 2^32: checked by the runtime)
            p_wrap = cq > 0 ? cs - cq : 0;
        }
        ct = next_tile(ct + stride, c_idx, c_m0, c_n0, c_nkt);
    }
#undef P8D_Q
    if (wr == 0) __builtin_amdgcn_s_barrier();        // pairs the extra barrier of wave row 1
    if (p_valid) {
        drain_block(0, 0); drain_block(0, 1); drain_block(0, 2); drain_block(0, 3);
        drain_block(1, 0); drain_block(1, 1); drain_block(1, 2); drain_block(1, 3);
        finish_sq();
    }
}

constexpr int kP8dLds = 3 * 3 * 128 * 128 + 8 * 2048;   // ring of three k-tiles + 2 KB of store staging per wave: 160 KB

constexpr int kP8Lds = 2 * (2 * 5 * 4096 + 32768);    // MI = 5: 144 KB
constexpr int kP8wLds = 128 * 1024 + 8 * 4096;         // the weight gradient's ring (MI = 4) + 4 KB of store staging per wave: 160 KB

}  // namespace

static bool g_p8_ready = false;
static int g_p8_n_cu = 256;

int ghn3_gemm_p8_init() {
    if (g_p8_ready) return GHN3_OK;
    hipError_t e = hipFuncSetAttribute((const void*)gemm_p8_kernel<GHN3_CT_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, kP8Lds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)gemm_p8_kernel<GHN3_CT_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, kP8Lds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)gemm_p8w_kernel<GHN3_CT_F16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kP8wLds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)gemm_p8w_kernel<GHN3_CT_BF16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kP8wLds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)gemm_p8w_kernel<GHN3_CT_F16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kP8wLds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)gemm_p8w_kernel<GHN3_CT_BF16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kP8wLds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)gemm_p8d_kernel<GHN3_CT_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, kP8dLds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)gemm_p8d_kernel<GHN3_CT_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, kP8dLds);
    if (e != hipSuccess) { ghn3_set_error("hipFuncSetAttribute(p8): %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    {
        int dev = 0, n_cu = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n_cu > 0)
            g_p8_n_cu = n_cu;
    }
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

// tile code 29: persistent stream (one workgroup per CU, or per CU the caller leaves to this launch; a multiple of 8 so
// that a workgroup's tiles keep their XCD)
int ghn3_gemm_p8w_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int ctype, int grid_cap, hipStream_t stream) {
    if (n_probs <= 0 || total_tiles <= 0) return GHN3_OK;
    if (ctype != GHN3_CT_F16 && ctype != GHN3_CT_BF16) {
        ghn3_set_error("8-phase GEMM needs compute type f16 or bf16 (got %d)", ctype);
        return GHN3_E_ARG;
    }
    int rc = ghn3_gemm_p8_init();
    if (rc) return rc;
    // grid_cap: low 16 bits = workers (CUs' worth of workgroups in flight), high bits = tiles per workgroup (0: persistent)
    const int tpw = grid_cap > 0 ? (grid_cap >> 16) : 0;
    grid_cap &= 0xffff;
    int grid = grid_cap > 0 ? grid_cap : g_p8_n_cu;
    if (grid > total_tiles) grid = total_tiles;
    grid = grid >= 8 ? grid / 8 * 8 : grid;
    if (grid < 8) grid = 8 < total_tiles ? 8 : total_tiles;     // (fewer than 8 tiles: one workgroup each)
    if (total_tiles >= 8 && grid % 8) grid = 8;
    const int per_worker = (total_tiles + grid - 1) / grid;
    const int chunks = tpw > 0 ? (per_worker + tpw - 1) / tpw : 1;
    static const int nt = getenv("GHN3_WGRAD_NT") ? atoi(getenv("GHN3_WGRAD_NT")) != 0 : 0;
    const int tpw_arg = tpw | (nt << 30);
    // quiet store k-tile (see the kernel): on by default since round 5 -- 0.95-0.97 -> 0.91-0.93 ms for the ghn3xlm16 bench
    // workload, bit-identical results (profiles/r05h_ab_wgrad_quiet_store_ktile.txt); GHN3_P8W_QUIET=0 = the round-4 loop
    static const int quiet = getenv("GHN3_P8W_QUIET") ? atoi(getenv("GHN3_P8W_QUIET")) != 0 : 1;
    if (ctype == GHN3_CT_F16 && quiet)
        hipLaunchKernelGGL((gemm_p8w_kernel<GHN3_CT_F16, 1>), dim3(grid * chunks), dim3(512), kP8wLds, stream, d_probs, n_probs,
                           total_tiles, grid, tpw_arg);
    else if (ctype == GHN3_CT_F16)
        hipLaunchKernelGGL((gemm_p8w_kernel<GHN3_CT_F16, 0>), dim3(grid * chunks), dim3(512), kP8wLds, stream, d_probs, n_probs,
                           total_tiles, grid, tpw_arg);
    else if (quiet)
        hipLaunchKernelGGL((gemm_p8w_kernel<GHN3_CT_BF16, 1>), dim3(grid * chunks), dim3(512), kP8wLds, stream, d_probs, n_probs,
                           total_tiles, grid, tpw_arg);
    else
        hipLaunchKernelGGL((gemm_p8w_kernel<GHN3_CT_BF16, 0>), dim3(grid * chunks), dim3(512), kP8wLds, stream, d_probs, n_probs,
                           total_tiles, grid, tpw_arg);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ghn3_set_error("p8w gemm launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    return GHN3_OK;
}

// tile code 30: persistent stream with two accumulator sets, 256 x 128 tiles (one workgroup per CU, or per CU the caller leaves
// to this launch; a multiple of 8 so that a workgroup's tiles keep their XCD)
int ghn3_gemm_p8d_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int ctype, int grid_cap, hipStream_t stream) {
    if (n_probs <= 0 || total_tiles <= 0) return GHN3_OK;
    if (ctype != GHN3_CT_F16 && ctype != GHN3_CT_BF16) {
        ghn3_set_error("8-phase GEMM needs compute type f16 or bf16 (got %d)", ctype);
        return GHN3_E_ARG;
    }
    int rc = ghn3_gemm_p8_init();
    if (rc) return rc;
    grid_cap &= 0xffff;
    int grid = grid_cap > 0 ? grid_cap : g_p8_n_cu;
    if (grid > total_tiles) grid = total_tiles;
    grid = grid >= 8 ? grid / 8 * 8 : grid;
    if (grid < 8) grid = 8 < total_tiles ? 8 : total_tiles;     // (fewer than 8 tiles: one workgroup each)
    if (total_tiles >= 8 && grid % 8) grid = 8;
    if (ctype == GHN3_CT_F16)
        hipLaunchKernelGGL(gemm_p8d_kernel<GHN3_CT_F16>, dim3(grid), dim3(512), kP8dLds, stream, d_probs, n_probs, total_tiles, grid);
    else
        hipLaunchKernelGGL(gemm_p8d_kernel<GHN3_CT_BF16>, dim3(grid), dim3(512), kP8dLds, stream, d_probs, n_probs, total_tiles, grid);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ghn3_set_error("p8d gemm launch: %s", hipGetErrorString(e)); return GHN3_E_HIP; }
    return GHN3_OK;
}
