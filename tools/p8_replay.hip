// Replays a dumped tile-code-28 launch (tools/p8_dump.py) on the GPU box with in-kernel cycle stamps (GHN3_P8_PROBE):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Iinclude -Ighn3_amd/csrc tools/p8_replay.hip -o tools/p8_replay
//   ./tools/p8_replay <dump.txt> [<dump2.txt> ...]
// Per dump: launch time and -- summed by thread 0 of every workgroup, averaged per tile -- prologue / k-loop / DMA-wait /
// epilogue cycles.  Operands are random f16 (zero-filled operands run a higher clock), K = 0 tiles count as tiles.
#ifndef GHN3_P8_PROBE
#define GHN3_P8_PROBE 1
#endif
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../ghn3_amd/csrc/gemm_p8.hip"
void ghn3_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }

static char *g_ws = nullptr, *g_sh = nullptr;
static size_t g_ws_n = 0, g_sh_n = 0;

static void fill_rand(unsigned short* p, size_t n, unsigned seed) {
    std::vector<unsigned short> h(1 << 22);
    unsigned x = seed;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3000 + ((x >> 9) & 0x0bff) + ((x >> 8) & 0x8000)); }
    for (size_t o = 0; o < n; o += h.size()) hipMemcpy(p + o, h.data(), 2 * std::min(h.size(), n - o), hipMemcpyHostToDevice);
}

static int g_cap = 0;
static void replay(const char* path, int reps) {
    FILE* f = fopen(path, "r");
    if (!f) { printf("%s: cannot open\n", path); return; }
    int n; long long wsb, shb;
    if (fscanf(f, "%d %lld %lld", &n, &wsb, &shb) != 3) { printf("%s: bad header\n", path); return; }
    if ((size_t)wsb + (1 << 20) > g_ws_n) { if (g_ws) hipFree(g_ws); g_ws_n = (size_t)wsb + (1 << 20); hipMalloc(&g_ws, g_ws_n); fill_rand((unsigned short*)g_ws, g_ws_n / 2, 1u); }
    if ((size_t)shb + (1 << 20) > g_sh_n) { if (g_sh) hipFree(g_sh); g_sh_n = (size_t)shb + (1 << 20); hipMalloc(&g_sh, g_sh_n); fill_rand((unsigned short*)g_sh, g_sh_n / 2, 7u); }
    struct Rec { long long v[21]; std::vector<int> mt; };
    std::vector<Rec> recs(n);
    std::vector<int> mt_all;
    for (auto& r : recs) {
        for (int i = 0; i < 21; ++i) if (fscanf(f, "%lld", &r.v[i]) != 1) { printf("bad record\n"); return; }
        r.mt.resize(3 * r.v[20]);
        for (auto& t : r.mt) if (fscanf(f, "%d", &t) != 1) { printf("bad mtab\n"); return; }
    }
    fclose(f);
    // launch order of the runtime (runtime.hip): XCD-pinned problems first, by XCD, when the launch has >= 8 members
    std::vector<int> order, pinned;
    const bool pin_ok = n >= 8;
    if (pin_ok)
        for (int x = 0; x < 8; ++x)
            for (int q = 0; q < n; ++q) if (recs[q].v[19] == x + 1) pinned.push_back(q);
    order = pinned;
    for (int q = 0; q < n; ++q) if (std::find(pinned.begin(), pinned.end(), q) == pinned.end()) order.push_back(q);
    const int n_pinned = (int)pinned.size();
    int pin_local[8] = {0}, pin_first[8] = {0}, pin_count[8] = {0}, pin_end = 0;
    for (int q : pinned) pin_local[recs[q].v[19] - 1] += (int)recs[q].v[20] * (int)((recs[q].v[1] + 255) / 256);
    for (int x = 0; x < 8; ++x) pin_end = std::max(pin_end, 8 * pin_local[x]);
    for (int x = 0; x < 8; ++x) pin_local[x] = 0;
    int tiles = pin_end;
    std::vector<GemmProbDev> hp(n);
    size_t mpos = 0;
    for (int q : order) for (int t : recs[q].mt) mt_all.push_back(t);
    int* dmt; hipMalloc(&dmt, mt_all.size() * 4 + 16); hipMemcpy(dmt, mt_all.data(), mt_all.size() * 4, hipMemcpyHostToDevice);
    double flops = 0; long long ktiles = 0;
    for (int i = 0; i < n; ++i) {
        const Rec& r = recs[order[i]];
        GemmProbDev& p = hp[i];
        memset(&p, 0, sizeof(p));
        auto base = [&](long long kind) { return kind ? g_sh : g_ws; };
        p.M = (int)r.v[0]; p.N = (int)r.v[1]; p.K = (int)r.v[2]; p.lda = (int)r.v[3]; p.ldb = (int)r.v[4]; p.ldc = (int)r.v[5];
        p.A = reinterpret_cast<const float*>(base(r.v[6]) + r.v[7]);
        p.B = reinterpret_cast<const float*>(base(r.v[8]) + r.v[9]);
        p.C = reinterpret_cast<float*>(base(r.v[10]) + r.v[11]);
        p.b_q = (int)r.v[12]; p.b_s = (int)r.v[13]; p.kq = (int)r.v[14]; p.ks = (int)r.v[15]; p.c_q = (int)r.v[16]; p.c_s = (int)r.v[17];
        p.lim_kind = (int)r.v[18];
        p.alpha = 1.f; p.flags = GHN3_GEMM_OP16; p.bias_stride = 1;
        p.tiles_m = (int)r.v[20]; p.tiles_n = (p.N + 255) / 256;
        p.mtab = dmt + mpos; mpos += r.mt.size();
        const int pin_x = i < n_pinned ? (int)r.v[19] - 1 : -1;
        p.pin = pin_x + 1; p.pin_end = pin_end; p.pin_total = n_pinned;
        if (pin_x >= 0) {
            p.tile_start = pin_local[pin_x];
            pin_local[pin_x] += p.tiles_m * p.tiles_n;
            if (pin_count[pin_x]++ == 0) pin_first[pin_x] = i;
        } else {
            p.tile_start = tiles;
            tiles += p.tiles_m * ((p.tiles_n + 7) / 8 * 8);
        }
        for (size_t t = 0; t < r.mt.size(); t += 3) {
            const int code = r.mt[t + 1], h = code <= 5 ? 64 * code : 32 * code, ext = std::max(r.mt[t + 2], 0);
            if (r.mt[t] >= p.M) continue;
            if (p.lim_kind == 2) { flops += 2.0 * h * p.N * std::min(p.K, ext); ktiles += (long long)p.tiles_n * ((std::min(p.K, ext) + 63) / 64); }
            else { flops += 2.0 * h * std::min(p.N, ext) * (double)p.K; ktiles += (long long)((std::min(p.N, ext) + 255) / 256) * ((p.K + 63) / 64); }
        }
    }
    if (n_pinned) for (int x = 0; x < 8; ++x) { hp[x].pin_first = pin_first[x]; hp[x].pin_count = pin_count[x]; }
    GemmProbDev* dp; hipMalloc(&dp, sizeof(GemmProbDev) * hp.size());
    hipMemcpy(dp, hp.data(), sizeof(GemmProbDev) * hp.size(), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) ghn3_gemm_p8_launch(dp, n, tiles, GHN3_CT_F16, g_cap, 0);
    hipDeviceSynchronize();
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st[8];
    hipMemcpyToSymbol(HIP_SYMBOL(g_p8_probe), z, sizeof(z));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) ghn3_gemm_p8_launch(dp, n, tiles, GHN3_CT_F16, g_cap, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_p8_probe), sizeof(st));
    const double nt = st[0] ? (double)st[0] : 1.0, nk = st[5] ? (double)st[5] : 1.0;
    printf("%-40s %7.3f ms  %5.0f TF padded | ids %5d, tiles run %6.0f per launch (%5.1f per CU), k-tiles %6.1f per CU: prologue %6.0f  k loop %7.0f (%5.0f per k-tile, DMA wait %5.0f)  epilogue %6.0f cycles per tile\n",
           path, ms / reps, flops * reps / ms * 1e-9, tiles, nt / reps, nt / reps / 256.0, nk / reps / 256.0, st[1] / nt, st[2] / nt, st[2] / nk, st[3] / nk, st[4] / nt);
    hipFree(dp); hipFree(dmt);
}

int main(int argc, char** argv) {
    // GHN3_REPLAY_CAP: grid cap of the launch (0 = one workgroup per tile id; 256 = persistent, one per CU, ids strided)
    if (getenv("GHN3_REPLAY_CAP")) g_cap = atoi(getenv("GHN3_REPLAY_CAP"));
    for (int i = 1; i < argc; ++i) replay(argv[i], 5);
    return 0;
}
