// Where a W2 forward launch of the 8-phase kernel (gemm_p8_kernel, tile code 28) spends its cycles, on the bench step's own
// problem mix (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Iinclude -Ighn3_amd/csrc tools/p8_probe.hip -o tools/p8_probe && ./tools/p8_probe
// Prints per variant: launch time, TFLOP/s on the padded tiles, and -- summed by thread 0 of every workgroup, averaged per tile --
// prologue / k-loop / DMA-wait / epilogue cycles.
#ifndef GHN3_P8_PROBE
#define GHN3_P8_PROBE 1
#endif
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../ghn3_amd/csrc/gemm_p8.hip"
void ghn3_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }

struct Range { int rows, o_lo, o_hi, i_ld; std::vector<int> mt; };   // mt: {m0, code, ext} triples

static unsigned short *gA, *gB; static float* gC;
static const int KK = 3072, MS1 = 384;

static void fill_rand(unsigned short* p, size_t n) {       // f16 values in [-1, 1): random mantissas (zero-filled operands run a higher clock)
    std::vector<unsigned short> h(1 << 20);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3000 + ((x >> 9) & 0x0bff) + ((x >> 8) & 0x8000)); }
    for (size_t o = 0; o < n; o += h.size()) hipMemcpy(p + o, h.data(), 2 * std::min(h.size(), n - o), hipMemcpyHostToDevice);
}

static void run(const char* name, const std::vector<Range>& rs, int reps = 5) {
    std::vector<GemmProbDev> hp(rs.size());
    std::vector<int> mt_all;
    int tiles = 0; double flops = 0, real = 0;
    for (size_t i = 0; i < rs.size(); ++i) for (int v : rs[i].mt) mt_all.push_back(v);
    int* dmt; hipMalloc(&dmt, mt_all.size() * 4 + 16); hipMemcpy(dmt, mt_all.data(), mt_all.size() * 4, hipMemcpyHostToDevice);
    size_t mpos = 0;
    for (size_t i = 0; i < rs.size(); ++i) {
        const Range& q = rs[i];
        GemmProbDev& p = hp[i];
        memset(&p, 0, sizeof(p));
        const int ncols = (q.o_hi - q.o_lo) * q.i_ld;
        p.A = reinterpret_cast<const float*>(gA);
        p.B = reinterpret_cast<const float*>(gB + (size_t)q.o_lo * MS1 * KK);
        p.C = gC + (size_t)q.o_lo * q.i_ld;
        p.M = q.rows; p.N = ncols; p.K = KK; p.lda = KK; p.ldb = KK; p.ldc = 147456 + 64; p.alpha = 1.f; p.flags = GHN3_GEMM_OP16;
        p.b_q = q.i_ld; p.b_s = MS1;
        p.tiles_m = (int)q.mt.size() / 3; p.tiles_n = (ncols + 255) / 256;
        p.mtab = dmt + mpos; mpos += q.mt.size();
        p.lim_kind = 1;
        p.tile_start = tiles;
        tiles += 8 * ((p.tiles_n + 7) / 8) * p.tiles_m;
        for (size_t t = 0; t < q.mt.size(); t += 3) {
            const int code = q.mt[t + 1], h = code <= 5 ? 64 * code : 32 * code;
            flops += 2.0 * h * std::min(ncols, q.mt[t + 2]) * KK;
        }
        real += 2.0 * q.rows * ncols * KK;
    }
    GemmProbDev* dp; hipMalloc(&dp, sizeof(GemmProbDev) * hp.size());
    hipMemcpy(dp, hp.data(), sizeof(GemmProbDev) * hp.size(), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) ghn3_gemm_p8_launch(dp, (int)hp.size(), tiles, GHN3_CT_F16, 0, 0);
    hipDeviceSynchronize();
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st[8];
    hipMemcpyToSymbol(HIP_SYMBOL(g_p8_probe), z, sizeof(z));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) ghn3_gemm_p8_launch(dp, (int)hp.size(), tiles, GHN3_CT_F16, 0, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_p8_probe), sizeof(st));
    const double n = st[0] ? (double)st[0] : 1.0;
    printf("%-52s %7.3f ms  %5.0f TF padded %5.0f TF rows | tiles/launch %5.0f: prologue %6.0f  k loop %7.0f (%5.0f per k-tile, DMA wait %5.0f per k-tile)  epilogue %6.0f cycles\n",
           name, ms / reps, flops * reps / ms * 1e-9, real * reps / ms * 1e-9, n / reps, st[1] / n, st[2] / n, (double)st[2] / (double)(st[5] ? st[5] : 1),
           (double)st[3] / (double)(st[5] ? st[5] : 1), st[4] / n);
    hipFree(dp); hipFree(dmt);
}

static std::vector<int> tiles_eq(int rows, int t, int code, int ext) {
    std::vector<int> v; const int h = code <= 5 ? 64 * code : 32 * code;
    for (int k = 0; k < t && k * h < rows; ++k) { v.push_back(k * h); v.push_back(code); v.push_back(ext); }
    return v;
}

int main() {
    hipMalloc(&gA, (size_t)2048 * KK * 2 + 4096);
    hipMalloc(&gB, (size_t)147456 * KK * 2 + 4096);
    hipMalloc(&gC, (size_t)2048 * (147456 + 64) * 4);
    fill_rand(gA, (size_t)2048 * KK); fill_rand(gB, (size_t)147456 * KK);
    const int FULL = 1 << 30;
    // round 5's tiling of the bench forward: family i = 384 (768 rows: 256 + 320 + 192 with a third of the width), family i = 128
    run("round-5 tiles: i384 {256,320,192/3} + i128",
        {{768, 0, 384, 384, {0, 4, 147456, 256, 5, 147456, 576, 3, 49152}}, {385, 0, 384, 128, {0, 4, 49152, 256, 3, 16384}}});
    // round 6: dense column ranges with equal tiles
    std::vector<Range> bal = {{768, 0, 32, 384, tiles_eq(768, 3, 8, FULL)}, {672, 32, 64, 384, tiles_eq(672, 3, 7, FULL)},
                              {660, 64, 128, 384, tiles_eq(660, 3, 7, FULL)}, {533, 128, 384, 384, tiles_eq(533, 2, 9, FULL)},
                              {385, 0, 32, 128, tiles_eq(385, 2, 7, FULL)}, {376, 32, 64, 128, tiles_eq(376, 2, 6, FULL)},
                              {365, 64, 128, 128, tiles_eq(365, 2, 6, FULL)}, {256, 128, 384, 128, tiles_eq(256, 1, 8, FULL)}};
    run("round-6 balanced ranges", bal);
    // single shapes: what a tile height costs when every tile of the launch is the same
    run("533 rows x 147456: 2 x 288", {{533, 0, 384, 384, tiles_eq(533, 2, 9, FULL)}});
    run("533 rows x 147456: 256 + 320", {{533, 0, 384, 384, {0, 4, FULL, 256, 5, FULL}}});
    run("512 rows x 147456: 2 x 256", {{512, 0, 384, 384, tiles_eq(512, 2, 8, FULL)}});
    run("768 rows x 147456: 3 x 256 (hipBLASLt's shape)", {{768, 0, 384, 384, tiles_eq(768, 3, 8, FULL)}});
    run("256 rows x 147456: 1 x 256 (no partner)", {{256, 0, 384, 384, tiles_eq(256, 1, 8, FULL)}});
    run("1024 rows x 147456: 4 x 256", {{1024, 0, 384, 384, tiles_eq(1024, 4, 8, FULL)}});
    run("640 rows x 147456: 2 x 320", {{640, 0, 384, 384, tiles_eq(640, 2, 10, FULL)}});
    return 0;
}
