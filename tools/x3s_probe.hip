// Phase stamps of the staged split-bf16 GEMM kernels of the Graphormer chain (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Iinclude -Ighn3_amd/csrc tools/x3s_probe.hip -o tools/x3s_probe && ./tools/x3s_probe
// 24 problems (one weight set each, as the 24 layers) launched back to back; A is rewritten by a small kernel in front of
// every launch (as in the chain: the rows were just produced by another kernel, on other XCDs).
#define GHN3_X3S_PROBE 1
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../ghn3_amd/csrc/gemm_x3d.hip"
void ghn3_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }

__global__ void touch_rows(float* x, int n, float v) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) x[i] = x[i] * 0.999f + v; }

static void run(const char* name, int code, int M, int N, int K, int ln, int bm, int bn, bool byval) {
    const int layers = 24;
    float *A, *C, *lnp; unsigned short* W;
    hipMalloc(&A, sizeof(float) * M * K); hipMalloc(&C, sizeof(float) * M * N);
    hipMalloc(&W, sizeof(unsigned short) * (size_t)layers * 2 * N * K);
    hipMalloc(&lnp, sizeof(float) * (size_t)(4 * K + 4 * M + 3 * M * K));
    std::vector<float> h((size_t)M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * (float)((i * 2654435761u) % 200) - 1.0f;
    hipMemcpy(A, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    hipMemset(W, 0x3c, sizeof(unsigned short) * (size_t)layers * 2 * N * K);
    hipMemset(lnp, 0, sizeof(float) * (size_t)(4 * K + 4 * M + 3 * M * K));
    std::vector<GemmProbDev> hp(layers);
    for (int l = 0; l < layers; ++l) {
        GemmProbDev& p = hp[l];
        memset(&p, 0, sizeof(p));
        p.A = A; p.C = C; p.B = reinterpret_cast<const float*>(W + (size_t)l * 2 * N * K); p.B2 = W + (size_t)l * 2 * N * K + (size_t)N * K;
        p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.alpha = 1.f;
        p.tile_start = 0; p.tiles_m = (M + bm - 1) / bm; p.tiles_n = (N + bn - 1) / bn;
        p.ln_kind = ln; p.ln_eps = 1e-5f;
        if (ln == 1) { p.ln_p[0] = lnp; p.ln_p[1] = lnp + K; p.ln_p[2] = lnp + 4 * K; p.ln_p[3] = lnp + 4 * K + M; p.ln_p[4] = lnp + 4 * K + 4 * M; }
        if (ln == 2) { p.ln_p[0] = lnp; p.ln_p[1] = lnp + 4 * K + 4 * M; p.ln_p[2] = lnp + 4 * K; p.ln_p[3] = lnp + 4 * K + M;
                       p.ln_p[4] = lnp + 4 * K + 4 * M + (size_t)M * K; p.ln_p[5] = lnp + 4 * K + 4 * M + 2 * (size_t)M * K; }
    }
    GemmProbDev* dp; hipMalloc(&dp, sizeof(GemmProbDev) * layers);
    hipMemcpy(dp, hp.data(), sizeof(GemmProbDev) * layers, hipMemcpyHostToDevice);
    const int tiles = hp[0].tiles_m * hp[0].tiles_n;
    const char* names[] = {"start -> problem found, weight loads issued", "rows landed, prologue, split, LDS written", "barrier",
                           "weights landed + products", "K-part reduction (barrier)", "epilogue"};
    for (int rep = 0; rep < 3; ++rep) {
        for (int l = 0; l < layers; ++l) {
            hipLaunchKernelGGL(touch_rows, dim3((M * K + 255) / 256), dim3(256), 0, 0, A, M * K, 0.001f);
            if (ghn3_gemm_x3s_launch(dp + l, byval ? hp.data() + l : nullptr, 1, tiles, code, K, ln, 0)) { printf("launch failed\n"); return; }
        }
        hipDeviceSynchronize();
        long long st[16];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_x3s_stamps), sizeof(st));
        printf("%s rep %d: workgroup 0 lives %lld cycles\n", name, rep, st[6] - st[0]);
        for (int i = 0; i < 6; ++i) printf("   %-50s %8lld\n", names[i], st[i + 1] - st[i]);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 10; ++r)
        for (int l = 0; l < layers; ++l) ghn3_gemm_x3s_launch(dp + l, byval ? hp.data() + l : nullptr, 1, tiles, code, K, ln, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.2f us per launch (240 back to back, instrumented build, %d workgroups)\n", name, ms * 1000.f / 240.f, tiles);
}

int main() {
    if (ghn3_gemm_x3s_init()) return 1;
    for (int bv = 0; bv < 2; ++bv) {
        printf("==== problem %s\n", bv ? "by value in the kernel arguments" : "through the device table");
        run("LN1 + to_qkv  (44, N 1152, K 384, ln 1)", 44, 256, 1152, 384, 1, 32, 48, bv);
        run("to_out        (45, N 384, K 384, ln 0)", 45, 256, 384, 384, 0, 16, 32, bv);
        run("ff.net.3      (45, N 384, K 1536, ln 0)", 45, 256, 384, 1536, 0, 16, 32, bv);
        run("LN1' + ff3 dgrad (44, N 1536, K 384, ln 2)", 44, 256, 1536, 384, 2, 32, 48, bv);
    }
    return 0;
}
