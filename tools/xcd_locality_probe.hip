// Does a consumer kernel find its producer kernel's output in L2?  (GPU box)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/xcd_locality_probe.hip -o tools/xcd_locality_probe && ./tools/xcd_locality_probe
// Kernel A: workgroup x (blockIdx x -> XCD x % 8) WRITES ring x (256 KB, a random pointer cycle, every hop a new 256-byte slot).
// Kernel B (next launch on the same stream): workgroup x chases ring (x + shift) % 8: shift 0 = the ring its own XCD wrote,
// shift 1..7 = a ring another XCD wrote.  Prints ns per dependent hop.  (The Graphormer chain hands 0.4-1.5 MB of activations from
// kernel to kernel; its workgroups are spread round-robin over the XCDs, so 7 of 8 such reads cross XCDs.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int SLOTS = 1024, STEP = 64;              // 1024 slots of 256 bytes = 256 KB per ring
__global__ void write_rings(unsigned* rings, const unsigned* next) {
    unsigned* r = rings + (size_t)blockIdx.x * SLOTS * STEP;
    for (int i = threadIdx.x; i < SLOTS; i += blockDim.x) r[(size_t)i * STEP] = next[i] * STEP;
}
__global__ void chase(const unsigned* rings, int shift, int hops, long long* t, unsigned* sink) {
    const unsigned* r = rings + (size_t)((blockIdx.x + shift) & 7) * SLOTS * STEP;
    unsigned p = 0;
    const long long r0 = wall_clock64();
    for (int k = 0; k < hops; ++k) p = r[p];
    const long long r1 = wall_clock64();
    if (threadIdx.x == 0) { t[blockIdx.x] = r1 - r0; sink[blockIdx.x] = p; }
}
int main() {
    std::vector<unsigned> order(SLOTS), next(SLOTS);
    for (int i = 0; i < SLOTS; ++i) order[i] = i;
    unsigned s = 777u;
    for (int i = SLOTS - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; std::swap(order[i], order[(s >> 8) % (i + 1)]); }
    for (int i = 0; i < SLOTS; ++i) next[order[i]] = order[(i + 1) % SLOTS];
    unsigned *d_rings, *d_next, *d_sink; long long* d_t;
    CHK(hipMalloc(&d_rings, (size_t)8 * SLOTS * STEP * 4)); CHK(hipMalloc(&d_next, SLOTS * 4)); CHK(hipMalloc(&d_sink, 64)); CHK(hipMalloc(&d_t, 64));
    CHK(hipMemcpy(d_next, next.data(), SLOTS * 4, hipMemcpyHostToDevice));
    const int hops = 800;                            // < SLOTS: every hop a line the consumer has not touched in this launch
    for (int rep = 0; rep < 2; ++rep)
        for (int shift : {0, 1, 4, 7, 0}) {
            hipLaunchKernelGGL(write_rings, dim3(8), dim3(256), 0, 0, d_rings, d_next);
            hipLaunchKernelGGL(chase, dim3(8), dim3(64), 0, 0, d_rings, shift, hops, d_t, d_sink);
            CHK(hipDeviceSynchronize());
            long long t[8]; CHK(hipMemcpy(t, d_t, 64, hipMemcpyDeviceToHost));
            double avg = 0; for (int i = 0; i < 8; ++i) avg += t[i];
            printf("producer kernel -> consumer kernel, ring written by XCD (x + %d) %% 8: %.0f ns per hop (workgroup 0: %.0f)\n", shift,
                   avg / 8 * 10.0 / hops, t[0] * 10.0 / hops);
        }
    // the same ring chased a second time by the same launch geometry (no producer in between): L2-warm reference
    hipLaunchKernelGGL(chase, dim3(8), dim3(64), 0, 0, d_rings, 0, hops, d_t, d_sink);
    hipLaunchKernelGGL(chase, dim3(8), dim3(64), 0, 0, d_rings, 0, hops, d_t, d_sink);
    CHK(hipDeviceSynchronize());
    long long t[8]; CHK(hipMemcpy(t, d_t, 64, hipMemcpyDeviceToHost));
    printf("re-read by the same XCD without a producer in between: %.0f ns per hop\n", t[0] * 10.0 / hops);
    return 0;
}
