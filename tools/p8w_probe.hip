// Per-k-tile cycles of the persistent W2 weight-gradient kernel (gemm_p8w_kernel) on problem mixes like the bench step's (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Iinclude -Ighn3_amd/csrc tools/p8w_probe.hip -o tools/p8w_probe && ./tools/p8w_probe
#ifndef GHN3_P8W_PROBE
#define GHN3_P8W_PROBE 1
#endif
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../ghn3_amd/csrc/gemm_p8.hip"
void ghn3_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }

struct Prob { int M, N, K; int cq = 0, cs = 0; };     // cq, cs: row map of C (r / cq) * cs + r % cq, as the band problems of dW2 have
static int g_kernel = 29;     // P8W_KERNEL=30: gemm_p8d_kernel (two accumulator sets, 256 x 128 tiles)
static int launch(const GemmProbDev* dp, int n, int tiles, int cap) {
    return g_kernel == 30 ? ghn3_gemm_p8d_launch(dp, n, tiles, GHN3_CT_F16, cap, 0) : ghn3_gemm_p8w_launch(dp, n, tiles, GHN3_CT_F16, cap, 0);
}
static void run(const char* name, const std::vector<Prob>& ps, int cap = 0) {
    std::vector<GemmProbDev> hp(ps.size());
    int tiles = 0;
    double flops = 0;
    std::vector<void*> frees;
    for (size_t i = 0; i < ps.size(); ++i) {
        const Prob& q = ps[i];
        const int ld = (q.K + 63) / 64 * 64;
        unsigned short *A, *B; float* C;
        hipMalloc(&A, (size_t)q.M * ld * 2 + 4096); hipMalloc(&B, (size_t)q.N * ld * 2 + 4096); hipMalloc(&C, (size_t)(q.cq > 0 ? (q.M / q.cq + 1) * q.cs : q.M) * q.N * 4);
        hipMemset(A, 0x11, (size_t)q.M * ld * 2 + 4096); hipMemset(B, 0x12, (size_t)q.N * ld * 2 + 4096);
        frees.push_back(A); frees.push_back(B); frees.push_back(C);
        GemmProbDev& p = hp[i];
        memset(&p, 0, sizeof(p));
        p.A = reinterpret_cast<const float*>(A); p.B = reinterpret_cast<const float*>(B); p.C = C;
        p.M = q.M; p.N = q.N; p.K = q.K; p.lda = ld; p.ldb = ld; p.ldc = q.N; p.alpha = 1.f; p.flags = GHN3_GEMM_OP16;
        p.c_q = q.cq; p.c_s = q.cs;
        p.tiles_m = (q.M + 255) / 256; p.tiles_n = g_kernel == 30 ? (q.N + 127) / 128 : (q.N + 255) / 256;
        int G = 1;
        while (G < 8 && (long long)q.N * q.K * 2 / G > (5 << 19) && p.tiles_n >= 2 * G) G *= 2;
        p.xcd_cols = G;
        p.tile_start = tiles;
        tiles += 8 * ((p.tiles_n + G - 1) / G) * ((p.tiles_m + 8 / G - 1) / (8 / G));
        flops += 2.0 * q.M * q.N * q.K;
    }
    GemmProbDev* dp; hipMalloc(&dp, sizeof(GemmProbDev) * hp.size());
    hipMemcpy(dp, hp.data(), sizeof(GemmProbDev) * hp.size(), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) launch(dp, (int)hp.size(), tiles, cap);
    hipDeviceSynchronize();
    long long st[48];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_p8w_probe), sizeof(st));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) launch(dp, (int)hp.size(), tiles, cap);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.3f ms  %6.0f TF | workgroup 0: first k-tiles (+ stores) %lld x %lld cycles (DMA wait %lld), other k-tiles %lld x %lld cycles (DMA wait %lld); per store k-tile: in store_prev %lld, at barriers %lld\n",
           name, ms / 5, flops * 5 / ms * 1e-9, st[2], st[2] ? st[0] / st[2] : 0, st[2] ? st[1] / st[2] : 0, st[5], st[5] ? st[3] / st[5] : 0,
           st[5] ? st[4] / st[5] : 0, st[2] ? st[6] / st[2] : 0, st[2] ? st[7] / st[2] : 0);
    {   // phases of the k-tile that stores (thread 0 of wave 0 = wave row 0, thread 256 = wave 4 = wave row 1), average cycles
        static const char* nm[18] = {"reads+issue3", "wait_vm0", "ST00", "bar", "MFMA00", "bar", "ST01", "reads+bar", "MFMA01", "bar", "ST11",
                                     "reads+bar", "MFMA11", "bar", "ST10", "issue+bar", "MFMA10", "bar"};
        const long long n = st[2] ? st[2] : 1;
        for (int w = 0; w < 2; ++w) {
            printf("      wave %d:", 4 * w);
            for (int i = 0; i < 18; ++i) printf(" %s %lld", nm[i], st[8 + 20 * w + i] / n);
            printf("\n");
        }
    }
    for (void* f : frees) hipFree(f);
    hipFree(dp);
}

int main() {
    if (getenv("P8W_KERNEL")) g_kernel = atoi(getenv("P8W_KERNEL"));
    printf("kernel: tile code %d\n", g_kernel);
    run("K 536 (9 k-tiles), M 65536", {{65536, 3072, 536}});
    run("K 1480 (24 k-tiles), M 16384", {{16384, 3072, 1480}});
    run("K 3072 (48 k-tiles), M 8192", {{8192, 3072, 3072}});
    run("bench mix", {{2048, 3072, 904}, {512, 3072, 1248}, {256, 3072, 1296}, {256, 3072, 1480}, {6144, 3072, 904}, {1536, 3072, 1248},
                      {768, 3072, 1296}, {768, 3072, 1424}, {8192, 3072, 856}, {2048, 3072, 1136}, {1024, 3072, 1168}, {1024, 3072, 1296},
                      {16384, 3072, 792}, {4096, 3072, 1032}, {2048, 3072, 1064}, {2048, 3072, 1176}, {65536, 3072, 536}, {16384, 3072, 664},
                      {8192, 3072, 680}, {8192, 3072, 776}});
    run("K 1480, M 16384, row map 64 -> 384", {{16384, 3072, 1480, 64, 384}});
    run("K 1480, M 16384, row map 8 -> 384", {{16384, 3072, 1480, 8, 384}});
    // fewer workgroups (grid cap): does the k-tile that carries a tile's stores get cheaper when fewer CUs store at the same time?
    for (int cap : {128, 64, 32, 8}) {
        char nm[64]; snprintf(nm, sizeof nm, "K 536, M 65536, %d workgroups", cap);
        run(nm, {{65536, 3072, 536}}, cap);
    }
    for (int cap : {128, 64, 32, 8}) {
        char nm[64]; snprintf(nm, sizeof nm, "K 1480, M 16384, %d workgroups", cap);
        run(nm, {{16384, 3072, 1480}}, cap);
    }
    return 0;
}
