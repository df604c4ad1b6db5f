// Micro-benchmark behind DESIGN.md section 6: cost of dependent back-to-back kernels in one stream on the GPU box
// (about 3 us whatever the grid).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O2 -o tools/launch_floor tools/launch_floor.hip   (in-tree: the binary travels with the
//   gpurun snapshot; it is git-ignored), then on the GPU box: ./tools/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_k(float* p) { if (p == nullptr) return; }
__global__ void touch_k(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
template <typename F> static float timeit(F f, int iters, hipStream_t s) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) f();
    hipEventRecord(a, s);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.0f / iters;
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    float* p; hipMalloc(&p, 1 << 24); hipMemset(p, 0, 1 << 24);
    int grids[] = {1, 16, 96, 288, 384, 1024};
    int blocks[] = {64, 256, 1024};
    for (int g : grids) for (int b : blocks) {
        float e = timeit([&] { empty_k<<<g, b, 0, s>>>(p); }, 2000, s);
        float t = timeit([&] { touch_k<<<g, b, 0, s>>>(p, g * b); }, 2000, s);
        printf("grid %5d block %5d : empty %.2f us  touch %.2f us\n", g, b, e, t);
    }
    return 0;
}
