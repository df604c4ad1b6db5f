// Probe (gfx950): is a producer -> consumer hand-off INSIDE one kernel cheap when all parties run on the same XCD?
//
// Round 1 measured 304 us for "the last workgroup of a row block normalises the rows" with agent-scope release / acquire
// (write-back + invalidate of the XCD's L2 per workgroup).  Workgroup b of a launch runs on XCD b % 8 and an XCD's L2 is
// coherent for the CUs of that XCD, so if every workgroup that touches a row block has the same b % 8, the hand-off only
// needs: stores complete (vmcnt(0): the L1 is write-through), a RELAXED atomic counter (executes in that L2), and consumer
// loads that bypass the reader's L1 (sc1 on the load itself) -- no fence instruction at all.
//
//   hipcc --offload-arch=gfx950 -O3 -o xcd_handoff_probe tools/xcd_handoff_probe.hip && ./xcd_handoff_probe
//
// Layout: X [256 rows][384 cols] fp32 = 8 row blocks of 32 rows x 12 column tiles of 32 columns; workgroup b = (column tile
// b / 8, row block b % 8).  Each producer writes its 32 x 32 tile (value = f(iteration, row, col)); the consumer sums every
// row of its row block (a stand-in for LayerNorm statistics) and writes S[row].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define GAS __attribute__((address_space(1)))
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 256, COLS = 384, RB = 8, CT = 12;

__device__ __forceinline__ float val(int it, int r, int c) { return (float)((it * 7 + r * 3 + c) & 1023) * 0.125f; }

__global__ __launch_bounds__(256) void produce(float* X, int it) {
    const int rb = blockIdx.x % RB, ct = blockIdx.x / RB, tid = threadIdx.x;
    const int r = rb * 32 + (tid >> 3), c = ct * 32 + (tid & 7) * 4;
    f32x4 v = {val(it, r, c), val(it, r, c + 1), val(it, r, c + 2), val(it, r, c + 3)};
    *reinterpret_cast<f32x4*>(X + (size_t)r * COLS + c) = v;
}
__global__ __launch_bounds__(256) void consume(const float* X, float* S) {
    const int rb = blockIdx.x, tid = threadIdx.x;
    const int r = rb * 32 + (tid >> 3);
    float acc = 0.f;
    for (int c = (tid & 7) * 4; c < COLS; c += 32) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(X + (size_t)r * COLS + c);
        acc += (v.x + v.y) + (v.z + v.w);
    }
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
    if ((tid & 7) == 0) S[r] = acc;
}
// fused: producers + last arriver of the row block consumes (all on XCD rb)
template <int MODE>   // 0: relaxed atomic + sc1 loads (XCD-local), 1: agent-scope release / acquire fences (portable)
__global__ __launch_bounds__(256) void fused(float* X, float* S, unsigned* cnt, int it) {
    __shared__ unsigned last;
    const int rb = blockIdx.x % RB, ct = blockIdx.x / RB, tid = threadIdx.x;
    {
        const int r = rb * 32 + (tid >> 3), c = ct * 32 + (tid & 7) * 4;
        f32x4 v = {val(it, r, c), val(it, r, c + 1), val(it, r, c + 2), val(it, r, c + 3)};
        *reinterpret_cast<f32x4*>(X + (size_t)r * COLS + c) = v;
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) last = __hip_atomic_fetch_add(cnt + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (last != CT - 1) return;
    if (tid == 0) __hip_atomic_store(cnt + rb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // for the next launch
    if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int r = rb * 32 + (tid >> 3);
    float acc = 0.f;
    for (int c = (tid & 7) * 4; c < COLS; c += 32) {
        f32x4 v;
        const float GAS* p = (const float GAS*)(X + (size_t)r * COLS + c);
        if (MODE == 0) {
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        } else {
            v = *reinterpret_cast<const f32x4*>(X + (size_t)r * COLS + c);
        }
        acc += (v.x + v.y) + (v.z + v.w);
    }
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
    if ((tid & 7) == 0) S[r] = acc;
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static double expect(int it, int r) { double s = 0; for (int c = 0; c < COLS; ++c) s += (double)(((it * 7 + r * 3 + c) & 1023) * 0.125f); return s; }

int main() {
    float *X, *S; unsigned* cnt;
    CHK(hipMalloc(&X, sizeof(float) * ROWS * COLS)); CHK(hipMalloc(&S, sizeof(float) * ROWS)); CHK(hipMalloc(&cnt, 64));
    CHK(hipMemset(cnt, 0, 64));
    std::vector<float> h(ROWS);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        int bad = 0;
        // correctness: every iteration writes new values; a stale read shows as a wrong row sum
        for (int it = 1; it <= 300; ++it) {
            if (mode == 0) { hipLaunchKernelGGL(produce, dim3(RB * CT), dim3(256), 0, 0, X, it); hipLaunchKernelGGL(consume, dim3(RB), dim3(256), 0, 0, X, S); }
            else if (mode == 1) hipLaunchKernelGGL(fused<0>, dim3(RB * CT), dim3(256), 0, 0, X, S, cnt, it);
            else hipLaunchKernelGGL(fused<1>, dim3(RB * CT), dim3(256), 0, 0, X, S, cnt, it);
            if (it % 37 == 0 || it == 300) {
                CHK(hipMemcpy(h.data(), S, sizeof(float) * ROWS, hipMemcpyDeviceToHost));
                for (int r = 0; r < ROWS; ++r) if (fabs(h[r] - expect(it, r)) > 1e-2 * (1 + fabs(expect(it, r)))) ++bad;
            }
        }
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0, 0));
        for (int it = 1; it <= iters; ++it) {
            if (mode == 0) { hipLaunchKernelGGL(produce, dim3(RB * CT), dim3(256), 0, 0, X, it); hipLaunchKernelGGL(consume, dim3(RB), dim3(256), 0, 0, X, S); }
            else if (mode == 1) hipLaunchKernelGGL(fused<0>, dim3(RB * CT), dim3(256), 0, 0, X, S, cnt, it);
            else hipLaunchKernelGGL(fused<1>, dim3(RB * CT), dim3(256), 0, 0, X, S, cnt, it);
        }
        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-62s %.2f us per iteration, wrong row sums in the checks: %d\n",
               mode == 0 ? "two dependent launches (produce; consume)" :
               mode == 1 ? "one launch, XCD-local hand-off (relaxed atomic + sc1 loads)" :
                           "one launch, agent-scope release / acquire fences", 1e3 * ms / iters, bad);
    }
    return 0;
}
