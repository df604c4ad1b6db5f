// Standalone lab for the 16-bit-operand W2 GEMM main loop (gfx950).  Not part of the library: kernels that win here
// move into ghn3_amd/csrc/gemm.hip.   C[M][N] (fp32) = A[M][K] * B[N][K]^T, A / B f16, k-contiguous.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip -o tools/gemm_lab
//   ./gemm_lab [variant-mask]
//
// Variants
//   0  two-stage 256 x 256 x 64 kernel of round 2 (one k-tile in flight, vmcnt(0) per k-tile): the baseline
//   1  "8-phase" 256 x 256 x 64: half-tile ring (8 x 16 KB), v_mfma_f32_16x16x32_f16, two wave groups half a phase apart
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

#define GAS __attribute__((address_space(1)))
#define LAS __attribute__((address_space(3)))
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef const unsigned short GAS* gch;
typedef f32x4 GAS* gf4;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct Prob {
    const unsigned short* A; const unsigned short* B; float* C;
    int M, N, K, lda, ldb, ldc;
    int tiles_m, tiles_n;
};

// XCD-aware tile order: the m-tiles of one n-tile (same streamed B panel) are congruent mod 8 -> same XCD / L2
__device__ __forceinline__ bool tile_origin(const Prob& P, int t, int BM, int BN, int& m0, int& n0) {
    const int grp = t >> 3;
    const int nt = (grp / P.tiles_m) * 8 + (t & 7), mt = grp % P.tiles_m;
    m0 = mt * BM; n0 = nt * BN;
    return m0 < P.M && n0 < P.N;
}

// ---------------------------------------------------------------------------------------------------------------
// variant 0: round-2 kernel (copy of h16d_tile<256, 256, 2, 4, NS = 2>, plain problems)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void gemm_v0(Prob P) {
    constexpr int BM = 256, BN = 256, BK = 64, NT = 512, WGN = 4, TM = 4, TN = 2, NS = 2;
    constexpr int OPA = BM * BK * 2, OPB = BN * BK * 2, STAGE = OPA + OPB, PA = BM * 8 / NT, PB = BN * 8 / NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sm = reinterpret_cast<char*>(smem);
    int m0, n0;
    if (!tile_origin(P, blockIdx.x, BM, BN, m0, n0)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WGN) * (BM / 2), wn0 = (wave % WGN) * 64;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int nkt = (P.K + BK - 1) / BK;
    gch pa[PA], pb[PB];
    {
        const int slot = tid & 7, rbase = tid >> 3;
        const int ck = (slot ^ ((rbase >> 1) & 7)) * 8;
        for (int i = 0; i < PA; ++i) pa[i] = (gch)P.A + (int64_t)min(m0 + rbase + (NT / 8) * i, P.M - 1) * P.lda + ck;
        for (int i = 0; i < PB; ++i) pb[i] = (gch)P.B + (int64_t)min(n0 + rbase + (NT / 8) * i, P.N - 1) * P.ldb + ck;
    }
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto issue = [&](int i) {
        const int k0 = i * BK;
        LAS char* la = (LAS char*)(sm + (i % NS) * STAGE);
#pragma unroll
        for (int j = 0; j < PA; ++j)
            __builtin_amdgcn_global_load_lds((const void GAS*)(pa[j] + k0), (LAS void*)(la + (wave * 64 + NT * j) * 16), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < PB; ++j)
            __builtin_amdgcn_global_load_lds((const void GAS*)(pb[j] + k0), (LAS void*)(la + OPA + (wave * 64 + NT * j) * 16), 16, 0, 0);
    };
    issue(0);
    for (int i = 0; i < nkt; ++i) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const char* a_s = sm + (i % NS) * STAGE;
        const char* b_s = a_s + OPA;
        u16x8 af[2][TM], bf[2][TN];
        auto load_frags = [&](int kk, int buf) {
            const int slot = kk * 2 + lhi;
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
        for (int kk = 0; kk < BK / 16; ++kk) {
            if (kk + 1 < BK / 16) load_frags(kk + 1, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ii = 0; ii < TM; ++ii)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf[kk & 1][j]),
                                                                       __builtin_bit_cast(f16x8, af[kk & 1][ii]), acc[ii][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk == 0 && i + 1 < nkt) { issue(i + 1); __builtin_amdgcn_sched_barrier(0); }
        }
    }
    // transposed products: lane = output row (l31), registers 4 g .. 4 g + 3 = columns 8 g + 4 lhi
#pragma unroll
    for (int ii = 0; ii < TM; ++ii) {
        const int row = m0 + wm0 + ii * 32 + l31;
        if (row >= P.M) continue;
        float GAS* crow = (float GAS*)P.C + (int64_t)row * P.ldc;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + wn0 + j * 32 + 8 * g + 4 * lhi;
                if (col + 3 < P.N) {
                    f32x4 v = {acc[ii][j][4 * g], acc[ii][j][4 * g + 1], acc[ii][j][4 * g + 2], acc[ii][j][4 * g + 3]};
                    *reinterpret_cast<gf4>(crow + col) = v;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// variant 1: 8-phase kernel
//   512 threads = 8 waves: wave row wr = wave >> 2 (128 output rows each), wave column wc = wave & 3 (64 columns each).
//   A k-tile (64 k) is four PHASES, one 64 x 32 quadrant Q(a, b) of the wave's 128 x 64 output each, in the order
//   Q00, Q01, Q11, Q10: 16 v_mfma_f32_16x16x32 on 8 independent accumulators (4 x 2 MFMA tiles x 2 k-steps).
//   LDS: ring of 2 k-tiles x 4 half-tiles (A0, B0, B1, A1) of 16 KB: A_h = the rows of sub-tile a = h of both wave rows,
//   B_h = the columns of sub-tile b = h of the four wave columns, so that phase 0 reads {A0, B0} (8 + 4 ds_read_b128),
//   phase 1 {B1} (4), phase 2 {A1} (8), phase 3 nothing.  Every phase issues ONE half-tile of LDS-DMA (2 pieces per
//   thread), D = 6 half-tiles ahead of the phase, into a buffer whose last read is at least two phases old; a half-tile
//   is waited for (counted vmcnt, 4 half-tiles = 64 KB stay in flight) in the phase before the one that reads it.
//   Two barriers per phase; wave row 1 runs one barrier behind wave row 0, so that on every SIMD one wave multiplies
//   while its partner reads fragments / issues DMA (cdna_hip_programming.md, "256^2 8-phase template").
// ---------------------------------------------------------------------------------------------------------------
// D = half-tiles issued ahead; WPP = 1: counted wait in every phase that precedes a read, 0: one wait per k-tile (phase 3);
// PRIO: 0 none, 1 s_setprio around the MFMA cluster, 2 static priority for wave row 1; NTB: aux bits of the B-operand loads
template <int TAIL_EXACT, int D = 6, int WPP = 1, int PRIO = 1, int NTB = 0>
__global__ __launch_bounds__(512, 2) void gemm_v1(Prob P) {
    constexpr int BM = 256, BN = 256, BK = 64, NT = 512;
    constexpr int HALF = 128 * 128;                  // bytes of a half-tile image: 128 rows x 128 B
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sm = reinterpret_cast<char*>(smem);
    int m0, n0;
    if (!tile_origin(P, blockIdx.x, BM, BN, m0, n0)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r16 = lane & 15, kc = lane >> 4;
    const int nkt = (P.K + BK - 1) / BK;
    const int nq = 4 * nkt;                          // half-tiles of this tile

    // DMA source pointers: piece i of half-tile h -> buffer row rho = (tid >> 3) + 64 i, slot tid & 7 holding the
    // 16-byte k chunk slot ^ ((rho >> 1) & 7)
    gch pa[2][2], pb[2][2];
    {
        const int slot = tid & 7, rb = tid >> 3;
        const int ck = (slot ^ ((rb >> 1) & 7)) * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rho = rb + 64 * i;
                const int ra = min(m0 + 128 * (rho >> 6) + 64 * h + (rho & 63), P.M - 1);
                const int rbn = min(n0 + 64 * (rho >> 5) + 32 * h + (rho & 31), P.N - 1);
                pa[h][i] = (gch)P.A + (int64_t)ra * P.lda + ck;
                pb[h][i] = (gch)P.B + (int64_t)rbn * P.ldb + ck;
            }
    }
    // half-tile q = 4 T + j, j: 0 = A0, 1 = B0, 2 = B1, 3 = A1 -> buffer (T & 1) * 4 + j
    auto issue = [&](int q) {
        int T = q >> 2;
        const int j = q & 3;
        if (TAIL_EXACT) { if (q >= nq) return; }
        else T = min(T, nkt - 1);                     // past the end: re-fetch the last k-tile into a dead buffer
        const int k0 = T * BK;
        LAS char* dst = (LAS char*)(sm + (((q >> 2) & 1) * 4 + j) * HALF + wave * 1024);
        if (j == 0 || j == 3) {
            const int h = j == 3;
            __builtin_amdgcn_global_load_lds((const void GAS*)(pa[h][0] + k0), (LAS void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void GAS*)(pa[h][1] + k0), (LAS void*)(dst + 8192), 16, 0, 0);
        } else {
            const int h = j == 2;
            __builtin_amdgcn_global_load_lds((const void GAS*)(pb[h][0] + k0), (LAS void*)dst, 16, 0, NTB);
            __builtin_amdgcn_global_load_lds((const void GAS*)(pb[h][1] + k0), (LAS void*)(dst + 8192), 16, 0, NTB);
        }
    };

    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[a][b][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses: lane l -> row r16 = l & 15 of a 16-row MFMA tile, 16-byte chunk ks * 4 + kc (kc = l >> 4)
    const int sw = r16 >> 1;
    const int offA = (64 * wr + r16) * 128 + ((kc ^ sw) << 4);      // + mi * 2048, ^ 64 for ks = 1
    const int offB = (32 * wc + r16) * 128 + ((kc ^ sw) << 4);      // + ni * 2048
    u16x8 fa[4][2], fb[2][2][2];
    auto read_a = [&](int T, int a) {
        const char* base = sm + ((T & 1) * 4 + (a ? 3 : 0)) * HALF;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            fa[mi][0] = *reinterpret_cast<const u16x8*>(base + offA + mi * 2048);
            fa[mi][1] = *reinterpret_cast<const u16x8*>(base + (offA ^ 64) + mi * 2048);
        }
    };
    auto read_b = [&](int T, int b) {
        const char* base = sm + ((T & 1) * 4 + 1 + b) * HALF;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            fb[b][ni][0] = *reinterpret_cast<const u16x8*>(base + offB + ni * 2048);
            fb[b][ni][1] = *reinterpret_cast<const u16x8*>(base + (offB ^ 64) + ni * 2048);
        }
    };
    auto mfma_q = [&](int a, int b) {
        wait_lgkm0();
        __builtin_amdgcn_sched_barrier(0);
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)       // transposed product: lane = output row, registers = 4 columns
                    acc[a][b][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                        __builtin_bit_cast(f16x8, fb[b][ni][ks]), __builtin_bit_cast(f16x8, fa[mi][ks]), acc[a][b][mi][ni], 0, 0, 0);
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // prologue: half-tiles 0 .. D - 1; k-tile 0's A0, B0 (read in phase 0) and B1 (phase 1) must have landed
#pragma unroll
    for (int q = 0; q < D; ++q) issue(q);
    if (TAIL_EXACT && nq < D) wait_vmcnt<0>(); else wait_vmcnt<2 * (D - (WPP ? 3 : 4))>();
    __builtin_amdgcn_s_barrier();
    if (wr == 1) { if (PRIO == 2) __builtin_amdgcn_s_setprio(1); __builtin_amdgcn_s_barrier(); }   // wave row 1 runs one barrier behind

    for (int T = 0; T < nkt; ++T) {
        const int g = 4 * T;
        // what phase p + 1 reads must have landed (this wave's pieces) before phase p's first barrier: with D = 6 the
        // newest half-tile needed is always 4 half-tiles older than the newest one issued
        auto wait_landed = [&](int newest_needed, int newest_issued) {
            constexpr int Y = WPP ? D - 2 : D - 4;        // half-tiles that may stay in flight in steady state
            if (TAIL_EXACT) {
                const int younger = min(newest_issued, nq - 1) - newest_needed;
                if (younger >= Y) wait_vmcnt<2 * Y>();
                else if (younger == 5) wait_vmcnt<10>();
                else if (younger == 4) wait_vmcnt<8>();
                else if (younger == 3) wait_vmcnt<6>();
                else if (younger == 2) wait_vmcnt<4>();
                else if (younger == 1) wait_vmcnt<2>();
                else wait_vmcnt<0>();
            } else wait_vmcnt<2 * Y>();
        };
        // ---- phase 0: Q00
        read_b(T, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(T, 0);
        issue(g + D);
        if (WPP) wait_landed(g + 2, g + D);           // B1 of this k-tile (read in phase 1)
        __builtin_amdgcn_s_barrier();
        mfma_q(0, 0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 1: Q01
        read_b(T, 1);
        issue(g + 1 + D);
        if (WPP) wait_landed(g + 3, g + 1 + D);       // A1 (phase 2)
        __builtin_amdgcn_s_barrier();
        mfma_q(0, 1);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: Q11
        read_a(T, 1);
        issue(g + 2 + D);
        __builtin_amdgcn_s_barrier();
        mfma_q(1, 1);
        __builtin_amdgcn_s_barrier();
        // ---- phase 3: Q10 (B0 and A1 are still in registers)
        issue(g + 3 + D);
        wait_landed(WPP ? g + 5 : g + 7, g + 3 + D);  // A0, B0 of the next k-tile (phase 0) / all of it
        __builtin_amdgcn_s_barrier();
        mfma_q(1, 0);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();        // pair the extra barrier of wave row 1
    wait_vmcnt<0>();

    // epilogue: lane = output row (r16 of a 16-row tile), its 4 registers = columns 4 kc .. 4 kc + 3
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int row = m0 + wr * 128 + a * 64 + mi * 16 + r16;
            if (row >= P.M) continue;
            float GAS* crow = (float GAS*)P.C + (int64_t)row * P.ldc;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int col = n0 + wc * 64 + b * 32 + ni * 16 + 4 * kc;
                    if (col + 3 < P.N) *reinterpret_cast<gf4>(crow + col) = acc[a][b][mi][ni];
                }
        }
}


// ---------------------------------------------------------------------------------------------------------------
// variant 2: the 8-phase kernel with a tile height of 64 MI rows (MI = 3, 4, 5 MFMA row tiles per quadrant: BM = 192,
// 256, 320) -- a family of 533 full-width decoder rows is 320 + 256 rows instead of three 256-row tiles.
//   A half-tile = 32 MI rows (4 MI KB), its DMA = 256 MI 16-byte pieces: full 512-thread rounds plus, for odd MI, one
//   round with lanes 0..31 of every wave (every wave issues the same number of instructions: one vmcnt for all).
//   32-bit byte offsets from the operand bases instead of 64-bit pointers (registers: MI = 5 has 160 accumulators).
//   D = 7 half-tiles ahead, ONE counted wait per k-tile (phase 3), exact tail.
// ---------------------------------------------------------------------------------------------------------------
template <int MI>
__global__ __launch_bounds__(512, 2) void gemm_v2(Prob P) {
    constexpr int BM = 64 * MI, BN = 256, BK = 64, D = 7;
    constexpr int AH = 32 * MI * 128, BH = 128 * 128, KT = 2 * AH + 2 * BH;     // bytes: A half, B half, one k-tile of the ring
    constexpr int NFA = MI / 2, ODD = MI & 1, NPA = NFA + ODD;                   // full rounds / half round of an A half-tile
    constexpr int NPT = 2 * NPA + 4;                                              // DMA instructions per thread and k-tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sm = reinterpret_cast<char*>(smem);
    int m0, n0;
    if (!tile_origin(P, blockIdx.x, BM, BN, m0, n0)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r16 = lane & 15, kc = lane >> 4;
    const int nkt = (P.K + BK - 1) / BK;
    const int nq = 4 * nkt;
    const char GAS* Ab = (const char GAS*)P.A;
    const char GAS* Bb = (const char GAS*)P.B;

    // byte offsets of this thread's pieces (k-tile 0)
    unsigned oa[2][NPA], ob[2][2];
    {
        const int slot = tid & 7, rb = tid >> 3;
        const int ck = (slot ^ ((rb >> 1) & 7)) * 16;
        auto arow = [&](int rho, int h) {             // buffer row of A_h -> tile row
            const int w = rho >= 16 * MI;
            return min(m0 + w * (BM / 2) + h * 16 * MI + (rho - w * 16 * MI), P.M - 1);
        };
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < NFA; ++i) oa[h][i] = (unsigned)arow(rb + 64 * i, h) * (unsigned)(P.lda * 2) + ck;
            if (ODD) {
                const int l = lane & 31;
                const int rho = 64 * NFA + wave * 4 + (l >> 3);
                oa[h][NFA] = (unsigned)arow(rho, h) * (unsigned)(P.lda * 2) + (((l & 7) ^ ((rho >> 1) & 7)) * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rho = rb + 64 * i;
                ob[h][i] = (unsigned)min(n0 + 64 * (rho >> 5) + 32 * h + (rho & 31), P.N - 1) * (unsigned)(P.ldb * 2) + ck;
            }
        }
    }
    // half-tile q = 4 T + j in ISSUE order j: 0 = B0, 1 = A0, 2 = B1, 3 = A1.  Phase g issues q = g + 7 into the buffer of
    // half-tile g - 1: A0 / B1 / A1 were last read two phases earlier; B0 was read in the phase just before (phase 0 of the
    // k-tile), but its four ds_reads are issued FIRST there and retired by a counted lgkmcnt before that phase's first barrier
    // (both wave rows), so the re-stage behind the barrier cannot overtake them.
    auto issue = [&](int q) {
        if (q >= nq) return;
        const int T = q >> 2, j = q & 3;
        const unsigned k0 = (unsigned)T * (BK * 2);
        char* kt = sm + (T & 1) * KT;
        if (j == 1 || j == 3) {
            const int h = j == 3;
            LAS char* dst = (LAS char*)(kt + (h ? AH + 2 * BH : 0));
#pragma unroll
            for (int i = 0; i < NFA; ++i)
                __builtin_amdgcn_global_load_lds((const void GAS*)(Ab + (oa[h][i] + k0)), (LAS void*)(dst + i * 8192 + wave * 1024), 16, 0, 0);
            if (ODD && lane < 32)
                __builtin_amdgcn_global_load_lds((const void GAS*)(Ab + (oa[h][NFA] + k0)), (LAS void*)(dst + NFA * 8192 + wave * 512), 16, 0, 0);
        } else {
            const int h = j == 2;
            LAS char* dst = (LAS char*)(kt + AH + h * BH + wave * 1024);
            __builtin_amdgcn_global_load_lds((const void GAS*)(Bb + (ob[h][0] + k0)), (LAS void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void GAS*)(Bb + (ob[h][1] + k0)), (LAS void*)(dst + 8192), 16, 0, 0);
        }
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

    const int sw = r16 >> 1;
    const int offA = (16 * MI * wr + r16) * 128 + ((kc ^ sw) << 4);
    const int offB = AH + (32 * wc + r16) * 128 + ((kc ^ sw) << 4);
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
        wait_lgkm0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[a][b][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                        __builtin_bit_cast(f16x8, fb[b][ni][ks]), __builtin_bit_cast(f16x8, fa[mi][ks]), acc[a][b][mi][ni], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // all but the `younger` newest half-tiles issued so far have landed (this thread's pieces); the newest issued is `last`
    // (clipped to the half-tiles that exist); a half-tile is NPA (A) or 2 (B) instructions
    auto wait_keep = [&](int last, int needed) {      // everything up to half-tile `needed` has landed; `last` = newest issued
        last = min(last, nq - 1);
        const int keep = last - needed;               // half-tiles that may stay in flight (<= 3; fewer at the tail)
        int n = 0;
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if (e < keep && last - e >= 0) { const int j = (last - e) & 3; n += (j == 1 || j == 3) ? NPA : 2; }
        // n is one of a few values: dispatch to immediates
        if (n >= 2 * NPA + 2) wait_vmcnt<2 * NPA + 2>();
        else if (n >= NPA + 4 && NPA + 4 < 2 * NPA + 2) wait_vmcnt<NPA + 4>();
        else if (n >= NPA + 2) wait_vmcnt<NPA + 2>();
        else if (n >= 4 && 4 < NPA + 2) wait_vmcnt<4>();
        else if (n >= NPA && NPA >= 2) wait_vmcnt<(NPA >= 2 ? NPA : 2)>();
        else if (n >= 2) wait_vmcnt<2>();
        else wait_vmcnt<0>();
    };

#pragma unroll
    for (int q = 0; q < D; ++q) issue(q);
    wait_keep(D - 1, 3);                              // k-tile 0 (half-tiles 0..3) complete
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();        // wave row 1 runs one barrier behind

    for (int T = 0; T < nkt; ++T) {
        const int g = 4 * T;
        read_b(T, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(T, 0);
        __builtin_amdgcn_sched_barrier(0);
        issue(g + D);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * MI) : "memory");   // the B0 reads are done: its buffer is re-staged next phase
        __builtin_amdgcn_s_barrier();
        mfma_q(0, 0);
        __builtin_amdgcn_s_barrier();
        read_b(T, 1);
        issue(g + 1 + D);
        __builtin_amdgcn_s_barrier();
        mfma_q(0, 1);
        __builtin_amdgcn_s_barrier();
        read_a(T, 1);
        issue(g + 2 + D);
        __builtin_amdgcn_s_barrier();
        mfma_q(1, 1);
        __builtin_amdgcn_s_barrier();
        issue(g + 3 + D);
        wait_keep(g + 3 + D, g + 7);                  // k-tile T + 1 complete (steady state: the three newest stay in flight)
        __builtin_amdgcn_s_barrier();
        mfma_q(1, 0);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();

#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row = m0 + wr * (BM / 2) + a * 16 * MI + mi * 16 + r16;
            if (row >= P.M) continue;
            float GAS* crow = (float GAS*)P.C + (int64_t)row * P.ldc;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int col = n0 + wc * 64 + b * 32 + ni * 16 + 4 * kc;
                    if (col + 3 < P.N) *reinterpret_cast<gf4>(crow + col) = acc[a][b][mi][ni];
                }
        }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ void ref_kernel(const unsigned short* A, const unsigned short* B, int K, int lda, int ldb, const int* sm_, const int* sn_,
                           float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const _Float16* a = reinterpret_cast<const _Float16*>(A) + (int64_t)sm_[i] * lda;
    const _Float16* b = reinterpret_cast<const _Float16*>(B) + (int64_t)sn_[i] * ldb;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)a[k] * (float)b[k];
    out[i] = s;
}

__global__ void fill_kernel(unsigned short* p, int64_t n, unsigned seed) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)(i * 2654435761u) ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        const float u = (float)(x & 0xffffff) / 16777216.0f * 2.f - 1.f;      // uniform [-1, 1)
        _Float16 h = (_Float16)u;
        p[i] = __builtin_bit_cast(unsigned short, h);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef void (*kern_t)(Prob);
struct Variant { const char* name; kern_t fn; int lds; int bm; };

static void run_case(const char* cname, int M, int N, int K, const std::vector<Variant>& vars, unsigned mask) {
    const int lda = (K + 63) / 64 * 64, ldb = lda, ldc = N;
    unsigned short *A, *B; float* C;
    CK(hipMalloc(&A, (size_t)M * lda * 2)); CK(hipMalloc(&B, (size_t)N * ldb * 2)); CK(hipMalloc(&C, (size_t)M * ldc * 4));
    fill_kernel<<<2048, 256>>>(A, (int64_t)M * lda, 0x1234567u);
    fill_kernel<<<2048, 256>>>(B, (int64_t)N * ldb, 0x89abcdeu);
    const int ns = 4096;
    std::vector<int> hm(ns), hn(ns);
    srand(1);
    for (int i = 0; i < ns; ++i) { hm[i] = rand() % M; hn[i] = rand() % N; }
    // corners / edges
    hm[0] = 0; hn[0] = 0; hm[1] = M - 1; hn[1] = N - 1; hm[2] = 0; hn[2] = N - 1; hm[3] = M - 1; hn[3] = 0;
    int *dm, *dn; float* dref;
    CK(hipMalloc(&dm, ns * 4)); CK(hipMalloc(&dn, ns * 4)); CK(hipMalloc(&dref, ns * 4));
    CK(hipMemcpy(dm, hm.data(), ns * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dn, hn.data(), ns * 4, hipMemcpyHostToDevice));
    ref_kernel<<<(ns + 255) / 256, 256>>>(A, B, K, lda, ldb, dm, dn, dref, ns);
    std::vector<float> href(ns);
    CK(hipMemcpy(href.data(), dref, ns * 4, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t v = 0; v < vars.size(); ++v) {
        if (!(mask & (1u << v))) continue;
        const int bm = vars[v].bm;
        Prob P{A, B, C, M, N, K, lda, ldb, ldc, (M + bm - 1) / bm, (N + 255) / 256};
        const int grid = P.tiles_m * ((P.tiles_n + 7) / 8 * 8);
        CK(hipMemset(C, 0xff, (size_t)M * ldc * 4));
        CK(hipFuncSetAttribute((const void*)vars[v].fn, hipFuncAttributeMaxDynamicSharedMemorySize, vars[v].lds));
        hipLaunchKernelGGL(vars[v].fn, dim3(grid), dim3(512), vars[v].lds, 0, P);
        CK(hipDeviceSynchronize());
        std::vector<float> hc(ns);
        double worst = 0; int nbad = 0;
        for (int i = 0; i < ns; ++i) {
            float c;
            CK(hipMemcpy(&c, C + (size_t)hm[i] * ldc + hn[i], 4, hipMemcpyDeviceToHost));
            const double d = fabs((double)c - href[i]);
            if (!(d <= worst)) worst = d;                 // (NaN sticks)
            if (!(d < 0.05)) { if (nbad < 6) printf("   bad sample m=%d n=%d got %g ref %g\n", hm[i], hn[i], c, href[i]); ++nbad; }
        }
        const int reps = 8;
        float best = 1e30f, sum = 0;
        for (int round = 0; round < 3; ++round) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(vars[v].fn, dim3(grid), dim3(512), vars[v].lds, 0, P);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps; sum += ms; if (ms < best) best = ms;
        }
        const double fl = 2.0 * M * N * K;
        printf("%-14s %-22s M=%6d N=%6d K=%6d  best %8.4f ms %7.1f TF   mean %7.1f TF   max|err| %.3g %s\n", cname, vars[v].name, M, N, K,
               best, fl / best * 1e-9, fl / (sum / 3) * 1e-9, worst, worst < 0.05 ? "ok" : "WRONG"); if (nbad) printf("   %d of %d samples bad\n", nbad, ns);
        fflush(stdout);
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(dm)); CK(hipFree(dn)); CK(hipFree(dref));
}

int main(int argc, char** argv) {
    unsigned mask = argc > 1 ? (unsigned)strtoul(argv[1], 0, 0) : 0xffffffffu;
    std::vector<Variant> vars = {
        {"v0 two-stage", gemm_v0, 128 * 1024, 256},
        {"v1 D6 wpp prio1", gemm_v1<1, 6, 1, 1, 0>, 128 * 1024, 256},
        {"v1 D7 ktile prio1", gemm_v1<1, 7, 0, 1, 0>, 128 * 1024, 256},
        {"v2 MI=4 (256)", gemm_v2<4>, 2 * (2 * 4 * 4096 + 32768), 256},
        {"v2 MI=5 (320)", gemm_v2<5>, 2 * (2 * 5 * 4096 + 32768), 320},
        {"v2 MI=3 (192)", gemm_v2<3>, 2 * (2 * 3 * 4096 + 32768), 192},
    };
    run_case("square4k", 4096, 4096, 4096, vars, mask);
    run_case("square8k", 8192, 8192, 8192, vars, mask);
    run_case("w2fwd", 768, 147456, 3072, vars, mask);
    run_case("w2fwd-640", 640, 147456, 3072, vars, mask);
    run_case("w2fwd-576", 576, 147456, 3072, vars, mask);
    run_case("wgrad-band", 65536, 3072, 576, vars, mask);
    run_case("short-k", 4096, 4096, 192, vars, mask);
    return 0;
}
