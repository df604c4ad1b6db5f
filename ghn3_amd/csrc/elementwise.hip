// HBM-bound kernels of the GHN-3 path for gfx950: integer graph prologue, embedding gathers, edge-bias
// table + gather + histogram, LayerNorm, tile/normalise/write, parameter norms, reductions.
// Each kernel cites the reference lines it replaces.  All index work is exact integer arithmetic.

#include "ghn3_internal.h"

#define MAX_DEGREE 100
#define MAX_INPUT_DIST 1000

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static int launch_ok(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ghn3_set_error("%s launch: %s", what, hipGetErrorString(e)); return GHN3_E_HIP; }
    return GHN3_OK;
}

// ------------------------------------------------------------------------------------------------
// graphormer.py:229-232,237 -- edges_1hop = (A == 1); in/out degree = column/row sums (clipped to 100);
// input distance = A[0, :] (clipped to 1000); fw/bw edge pair (A[i,j], A[j,i]) -> one table index.
// One block per (row i, graph b); row read coalesced, column read strided (N <= 1024, L2 resident).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void graph_prologue_kernel(const int64_t* __restrict__ A, int* __restrict__ deg_in,
                                                             int* __restrict__ deg_out, int* __restrict__ dist0,
                                                             int* __restrict__ pair, int N, int V) {
    __shared__ int red[2][4];
    const int i = blockIdx.x, b = blockIdx.y;
    const int64_t* Ab = A + (size_t)b * N * N;
    int cnt_out = 0, cnt_in = 0;
    for (int j = threadIdx.x; j < N; j += 256) {
        const int64_t fw = Ab[(size_t)i * N + j];
        const int64_t bw = Ab[(size_t)j * N + i];
        cnt_out += (fw == 1);
        cnt_in += (bw == 1);
        int f = (int)(fw < 0 ? 0 : (fw >= V ? V - 1 : fw));
        int g = (int)(bw < 0 ? 0 : (bw >= V ? V - 1 : bw));
        pair[((size_t)b * N + i) * N + j] = f * V + g;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cnt_out += __shfl_xor(cnt_out, o, 64);
        cnt_in += __shfl_xor(cnt_in, o, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = cnt_out; red[1][w] = cnt_in; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int o_ = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        int i_ = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        deg_out[b * N + i] = o_ > MAX_DEGREE ? MAX_DEGREE : o_;
        deg_in[b * N + i] = i_ > MAX_DEGREE ? MAX_DEGREE : i_;
        int64_t dd = Ab[i];                                   // A[b, 0, i]
        dist0[b * N + i] = (int)(dd < 0 ? 0 : (dd > MAX_INPUT_DIST ? MAX_INPUT_DIST : dd));
    }
}

int ghn3_graph_prologue(const int64_t* A, int* deg_in, int* deg_out, int* dist0, int* pair, int B, int N, int V,
                        hipStream_t s) {
    hipLaunchKernelGGL(graph_prologue_kernel, dim3(N, B), dim3(256), 0, s, A, deg_in, deg_out, dist0, pair, N, V);
    return launch_ok("graph_prologue");
}

// ------------------------------------------------------------------------------------------------
// nn.py:248-253 (embed + shape_enc + to_dense) fused with graphormer.py:230-235 (centrality / input-dist
// embeddings, node mask).  One wave per dense row; padded rows are written as exact zeros.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_nodes_kernel(
    float* __restrict__ x, const int* __restrict__ node_type, const int* __restrict__ shape_idx,
    const int* __restrict__ n_nodes, const int* __restrict__ node_off, const float* __restrict__ E_type,
    const float* __restrict__ E_ch, const float* __restrict__ E_sp, const float* __restrict__ E_in,
    const float* __restrict__ E_out, const float* __restrict__ E_dist, const int* __restrict__ deg_in,
    const int* __restrict__ deg_out, const int* __restrict__ dist0, int B, int N, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * N) return;
    const int lane = threadIdx.x & 63;
    const int b = row / N, i = row - b * N;
    float* xr = x + (size_t)row * C;
    if (i >= n_nodes[b]) {
        for (int c = lane; c < C; c += 64) xr[c] = 0.f;
        return;
    }
    const int sidx = node_off[b] + i;
    const int t = node_type[sidx];
    const int s0 = shape_idx[4 * sidx], s1 = shape_idx[4 * sidx + 1], s2 = shape_idx[4 * sidx + 2],
              s3 = shape_idx[4 * sidx + 3];
    const int di = deg_in[row], dout = deg_out[row], dd = dist0[row];
    const int cq = C >> 2;
    for (int c = lane; c < C; c += 64) {
        const int part = c / cq, cc = c - part * cq;
        float sh;
        if (part == 0) sh = E_ch[(size_t)s0 * cq + cc];
        else if (part == 1) sh = E_ch[(size_t)s1 * cq + cc];
        else if (part == 2) sh = E_sp[(size_t)s2 * cq + cc];
        else sh = E_sp[(size_t)s3 * cq + cc];
        float v = E_type[(size_t)t * C + c] + sh;
        v += E_in[(size_t)di * C + c];
        v += E_out[(size_t)dout * C + c];
        v += E_dist[(size_t)dd * C + c];
        xr[c] = v;
    }
}

int ghn3_embed_nodes(float* x, const int* node_type, const int* shape_idx, const int* n_nodes, const int* node_off,
                     const float* E_type, const float* E_ch, const float* E_sp, const float* E_in,
                     const float* E_out, const float* E_dist, const int* deg_in, const int* deg_out,
                     const int* dist0, int B, int N, int C, hipStream_t s) {
    if (C % 4) { ghn3_set_error("embed: C=%d not a multiple of 4", C); return GHN3_E_ARG; }
    hipLaunchKernelGGL(embed_nodes_kernel, dim3((B * N + 3) / 4), dim3(256), 0, s, x, node_type, shape_idx, n_nodes,
                       node_off, E_type, E_ch, E_sp, E_in, E_out, E_dist, deg_in, deg_out, dist0, B, N, C);
    return launch_ok("embed_nodes");
}

// Deterministic scatter-add: one workgroup per TABLE ROW gathers the gradient rows of the nodes that index it, in node
// order (float atomics in node-major order gave run-to-run different sums).  Tables in block order: type (n_type rows),
// channel (n_ch), spatial (n_sp), in-degree (101), out-degree (101), input distance (1001); every workgroup scans the
// node index arrays (a few hundred nodes), rows nobody indexes are left untouched (zero-initialised gradients).
__global__ __launch_bounds__(256) void embed_bwd_kernel(
    const float* __restrict__ dx, const int* __restrict__ node_type, const int* __restrict__ shape_idx,
    const int* __restrict__ n_nodes, const int* __restrict__ node_off, float* __restrict__ dE_type,
    float* __restrict__ dE_ch, float* __restrict__ dE_sp, float* __restrict__ dE_in, float* __restrict__ dE_out,
    float* __restrict__ dE_dist, const int* __restrict__ deg_in, const int* __restrict__ deg_out,
    const int* __restrict__ dist0, int B, int N, int C, int n_type, int n_ch, int n_sp) {
    extern __shared__ unsigned long long emb_mask[];     // [2][ceil(B * N / 64)]: nodes matching column block 0 / 1
    int r = blockIdx.x, tab = 0;                         // (no local array: a dynamically indexed one lives in scratch)
    if (r >= n_type) { r -= n_type; tab = 1;
        if (r >= n_ch) { r -= n_ch; tab = 2;
            if (r >= n_sp) { r -= n_sp; tab = 3;
                if (r >= 101) { r -= 101; tab = 4;
                    if (r >= 101) { r -= 101; tab = 5; } } } } }
    const int cq = C >> 2;
    const int width = (tab == 1 || tab == 2) ? cq : C;
    const int base0 = tab == 2 ? 2 * cq : 0, base1 = tab == 1 ? cq : 3 * cq;
    const int rows = B * N, words = (rows + 63) >> 6;
    // pass 1 (parallel): which nodes index this table row -- one ballot word per 64 dense rows
    int any = 0;
    for (int row0 = (threadIdx.x >> 6) * 64; row0 < rows; row0 += 256) {
        const int row = row0 + (threadIdx.x & 63);
        int f = 0;
        if (row < rows) {
            const int b = row / N, i = row - b * N;
            if (i < n_nodes[b]) {                      // x * mask: padded rows carry no gradient
                const int sidx = node_off[b] + i;
                if (tab == 0) f = node_type[sidx] == r;
                else if (tab == 1) f = (shape_idx[4 * sidx] == r) | ((shape_idx[4 * sidx + 1] == r) << 1);
                else if (tab == 2) f = (shape_idx[4 * sidx + 2] == r) | ((shape_idx[4 * sidx + 3] == r) << 1);
                else if (tab == 3) f = deg_in[row] == r;
                else if (tab == 4) f = deg_out[row] == r;
                else f = dist0[row] == r;
            }
        }
        const unsigned long long m0 = __ballot(f & 1), m1 = __ballot(f & 2);
        if ((threadIdx.x & 63) == 0) { emb_mask[row0 >> 6] = m0; emb_mask[words + (row0 >> 6)] = m1; }
        any |= f;
    }
    if (!__syncthreads_or(any)) return;               // nobody indexes this row: its (zero-initialised) gradient stays
    // pass 2: the matching gradient rows, summed in node order.  Thread 0 expands the ballot words into an ordered list
    // (row | block flags << 24); everybody then walks it eight entries at a time with all loads of a group in flight (a
    // hot row -- the `conv` type, degree 1 -- matches >100 nodes: one dependent round trip per node was 118 us).
    int* lst = reinterpret_cast<int*>(emb_mask + 2 * words);
    __shared__ int n_match;
    if (threadIdx.x == 0) {
        int n = 0;
        for (int w = 0; w < words; ++w) {
            const unsigned long long m0 = emb_mask[w], m1 = emb_mask[words + w];
            unsigned long long m = m0 | m1;
            while (m) {
                const int bit = __ffsll((long long)m) - 1;
                m &= m - 1;
                lst[n++] = (w * 64 + bit) | ((int)((m0 >> bit) & 1) << 24) | ((int)((m1 >> bit) & 1) << 25);
            }
        }
        n_match = n;
    }
    __syncthreads();
    const int n = n_match;
    float acc[2] = {0.f, 0.f};                         // columns threadIdx.x and threadIdx.x + 256 (C <= 512)
    for (int i0 = 0; i0 < n; i0 += 8) {
        float v0[8][2], v1[8][2];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ent = i0 + e < n ? lst[i0 + e] : 0;
            const float* g = dx + (size_t)(ent & 0xffffff) * C;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c = threadIdx.x + 256 * k;
                const bool in = i0 + e < n && c < width;
                v0[e][k] = (in && (ent & (1 << 24))) ? g[base0 + c] : 0.f;
                v1[e][k] = (in && (ent & (1 << 25))) ? g[base1 + c] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int k = 0; k < 2; ++k) { acc[k] += v0[e][k]; acc[k] += v1[e][k]; }
    }
    float* dst = tab == 0 ? dE_type : tab == 1 ? dE_ch : tab == 2 ? dE_sp : tab == 3 ? dE_in : tab == 4 ? dE_out : dE_dist;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = threadIdx.x + 256 * k;
        if (c < width) dst[(size_t)r * width + c] += acc[k];
    }
}

int ghn3_embed_bwd(const float* dx, const int* node_type, const int* shape_idx, const int* n_nodes,
                   const int* node_off, float* dE_type, float* dE_ch, float* dE_sp, float* dE_in, float* dE_out,
                   float* dE_dist, const int* deg_in, const int* deg_out, const int* dist0, int B, int N, int C,
                   int n_type, int n_ch, int n_sp, hipStream_t s) {
    if (C > 512 || n_type <= 0 || n_ch <= 0 || n_sp <= 0) {
        ghn3_set_error("embed_bwd: C <= 512 and the table row counts (i3..i5) are required");
        return GHN3_E_ARG;
    }
    if (B * N > 12 * 1024) { ghn3_set_error("embed_bwd: more than 12288 dense node rows"); return GHN3_E_LIMIT; }
    const int blocks = n_type + n_ch + n_sp + 101 + 101 + 1001;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(blocks), dim3(256), (size_t)(16 * ((B * N + 63) / 64) + 4 * (size_t)B * N + 16), s, dx, node_type, shape_idx, n_nodes,
                       node_off, dE_type, dE_ch, dE_sp, dE_in, dE_out, dE_dist, deg_in, deg_out, dist0, B, N, C, n_type,
                       n_ch, n_sp);
    return launch_ok("embed_bwd");
}

// ------------------------------------------------------------------------------------------------
// graphormer.py:115-117 factorised over the <= V*V distinct (fw, bw) pairs:
//   proj_e.0(cat(E[fw+2], E[bw+2])) = W0[:, :C] E[fw+2] + (W0[:, C:] E[bw+2] + b0) = Pfw[fw] + Pbw[bw]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_hidden_kernel(float* __restrict__ hid, const float* __restrict__ Pfw,
                                                          const float* __restrict__ Pbw, int V, int C) {
    const int p = blockIdx.x;
    const int fw = p / V, bw = p - fw * V;
    for (int c = threadIdx.x; c < C; c += 256)
        hid[(size_t)p * C + c] = fmaxf(Pfw[(size_t)fw * C + c] + Pbw[(size_t)bw * C + c], 0.f);
}
int ghn3_edge_hidden(float* hid, const float* Pfw, const float* Pbw, int V, int C, hipStream_t s) {
    hipLaunchKernelGGL(edge_hidden_kernel, dim3(V * V), dim3(256), 0, s, hid, Pfw, Pbw, V, C);
    return launch_ok("edge_hidden");
}

// dhid masked by (hid > 0) in place; dPfw[fw] = sum_bw dhid ; dPbw[bw] = sum_fw dhid.
// One workgroup per (table row, 64 columns): its four waves split the V terms of the sum four ways (independent loads in
// flight) and add their partial sums in a fixed order -- V workgroups of one wave's worth of work each took 52 + 25 us at
// the end of every backward, on an otherwise idle chip.
__global__ __launch_bounds__(256) void edge_hidden_bwd_fw_kernel(float* __restrict__ dPfw, float* __restrict__ dhid,
                                                                 const float* __restrict__ hid, int V, int C) {
    __shared__ float part[4][64];
    const int fw = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < C)
        for (int bw = w; bw < V; bw += 4) {
            const size_t o = ((size_t)fw * V + bw) * C + c;
            const float g = hid[o] > 0.f ? dhid[o] : 0.f;
            dhid[o] = g;
            acc += g;
        }
    part[w][threadIdx.x & 63] = acc;
    __syncthreads();
    if (w == 0 && c < C) dPfw[(size_t)fw * C + c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void edge_hidden_bwd_bw_kernel(float* __restrict__ dPbw,
                                                                 const float* __restrict__ dhid, int V, int C) {
    __shared__ float part[4][64];
    const int bw = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < C)
        for (int fw = w; fw < V; fw += 4) acc += dhid[((size_t)fw * V + bw) * C + c];
    part[w][threadIdx.x & 63] = acc;
    __syncthreads();
    if (w == 0 && c < C) dPbw[(size_t)bw * C + c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}
int ghn3_edge_hidden_bwd(float* dPfw, float* dPbw, float* dhid, const float* hid, int V, int C, hipStream_t s) {
    hipLaunchKernelGGL(edge_hidden_bwd_fw_kernel, dim3(V, (C + 63) / 64), dim3(256), 0, s, dPfw, dhid, hid, V, C);
    hipLaunchKernelGGL(edge_hidden_bwd_bw_kernel, dim3(V, (C + 63) / 64), dim3(256), 0, s, dPbw, dhid, V, C);
    return launch_ok("edge_hidden_bwd");
}

// bias[b,h,i,j] = T[pair[b,i,j]][h]   (T has leading dimension ldT >= H, a multiple of 4)
// One workgroup per (graph, query row, 256 keys).  Round 6: staged through LDS -- the table rows the workgroup's pairs name are
// fetched cooperatively, ldT / 4 adjacent lanes per row (one 16-byte piece each: a row is one coalesced segment instead of
// ldT / 4 separate requests of the lane that owns the pair), parked as [pair][ldT + 1] floats (odd stride: the column reads
// below are conflict-free), and every head's 256 values leave as one coalesced 1 KB store.  ldT > 64 (more than 64 heads)
// keeps the direct gather.
__global__ __launch_bounds__(256) void bias_gather_kernel(float* __restrict__ bias, const float* __restrict__ T,
                                                          const int* __restrict__ pair, int N, int H, int ldT) {
    extern __shared__ float bg_rows[];               // [256][ldT + 1]
    __shared__ int bg_pair[256];
    const int b = blockIdx.z, i = blockIdx.y;
    const int j0 = blockIdx.x * 256, j = j0 + threadIdx.x;
    const int nj = min(256, N - j0);
    bg_pair[threadIdx.x] = j < N ? pair[((size_t)b * N + i) * N + j] : 0;
    __syncthreads();
    const int lpr = ldT >> 2;                         // lanes per table row (16-byte pieces)
    const int ld = ldT + 1;
    for (int e = threadIdx.x; e < nj * lpr; e += 256) {
        const int q = e / lpr, c = e - q * lpr;
        const float4 v = *reinterpret_cast<const float4*>(T + (size_t)bg_pair[q] * ldT + 4 * c);
        float* d = bg_rows + q * ld + 4 * c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    if (j >= N) return;
    const float* r = bg_rows + threadIdx.x * ld;
    for (int h = 0; h < H; ++h) bias[(((size_t)b * H + h) * N + i) * N + j] = r[h];
}
__global__ __launch_bounds__(256) void bias_gather_direct_kernel(float* __restrict__ bias, const float* __restrict__ T,
                                                                 const int* __restrict__ pair, int N, int H, int ldT) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const int p = pair[((size_t)b * N + i) * N + j];
    const float* tr = T + (size_t)p * ldT;
    for (int h = 0; h < H; ++h) bias[(((size_t)b * H + h) * N + i) * N + j] = tr[h];
}
int ghn3_bias_gather(float* bias, const float* T, const int* pair, int B, int N, int H, hipStream_t s) {
    const int ldT = (H + 3) & ~3;
    if (ldT <= 64)
        hipLaunchKernelGGL(bias_gather_kernel, dim3((N + 255) / 256, N, B), dim3(256), 256 * (ldT + 1) * sizeof(float), s, bias, T,
                           pair, N, H, ldT);
    else
        hipLaunchKernelGGL(bias_gather_direct_kernel, dim3((N + 255) / 256, N, B), dim3(256), 0, s, bias, T, pair, N, H, ldT);
    return launch_ok("bias_gather");
}

// dT[p][h] += sum_{(b,i,j): pair == p} dBias[b,h,i,j]  -- deterministic: the histogram is accumulated in 64-bit FIXED
// POINT (integer addition is associative, so the order in which the atomics land does not matter).  Three small passes:
// max |dBias| (atomicMax on the bit pattern) -> scale 2^(30 - e) for amax = m 2^e, so that every term is below 2^31 and
// 2^24 of them still fit 63 bits with 2^-31 relative resolution (finer than an fp32 sum); the LDS-private int64
// histograms per (row chunk, head, graph) flushed with 64-bit global atomics; the conversion back to fp32.
// scratch: int64 [V * V * H] (zeroed by the caller) followed by one float (amax, zeroed with it).
__global__ __launch_bounds__(256) void bias_amax_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ amax) {
    float m = 0.f;
    const int64_t n4 = (reinterpret_cast<uintptr_t>(x) & 15) ? 0 : (n >> 2);
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        const float4 v = x4[e];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (int64_t e = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[e]));
    // (wave reduction + a racy pre-check of the slot: 4096 waves doing an atomic on ONE address took 40 of this kernel's 50 us)
    ghn3_atomic_amax(amax, m);
}
__device__ __forceinline__ float fix_scale(float amax) {            // 2^(30 - e), amax = m 2^e
    if (!(amax > 0.f)) return 1.f;
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127;
    return __uint_as_float((unsigned)(127 + 30 - e) << 23);
}
__global__ __launch_bounds__(256) void bias_hist_kernel(long long* __restrict__ acc, const float* __restrict__ dBias,
                                                        const int* __restrict__ pair, int N, int H, int V,
                                                        int rows_per_block, int use_lds, const float* __restrict__ amax) {
    extern __shared__ __attribute__((aligned(16))) long long hist[];
    const int b = blockIdx.z, h = blockIdx.y;
    const int i0 = blockIdx.x * rows_per_block;
    const int i1 = min(N, i0 + rows_per_block);
    const int bins = V * V;
    const float sc = fix_scale(*amax);
    if (use_lds) {
        for (int k = threadIdx.x; k < bins; k += 256) hist[k] = 0;
        __syncthreads();
    }
    const size_t total = (size_t)(i1 - i0) * N;
    for (size_t e = threadIdx.x; e < total; e += 256) {
        const size_t off = (size_t)i0 * N + e;
        const int p = pair[(size_t)b * N * N + off];
        const long long g = (long long)llrintf(dBias[((size_t)b * H + h) * N * N + off] * sc);
        if (g == 0) continue;
        if (use_lds) atomicAdd(reinterpret_cast<unsigned long long*>(&hist[p]), (unsigned long long)g);
        else atomicAdd(reinterpret_cast<unsigned long long*>(&acc[(size_t)p * H + h]), (unsigned long long)g);
    }
    if (use_lds) {
        __syncthreads();
        for (int k = threadIdx.x; k < bins; k += 256) {
            const long long v = hist[k];
            if (v != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&acc[(size_t)k * H + h]), (unsigned long long)v);
        }
    }
}
__global__ __launch_bounds__(256) void bias_hist_finish_kernel(float* __restrict__ dT, const long long* __restrict__ acc,
                                                               int bins, int H, int ldT, const float* __restrict__ amax) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= bins * H) return;
    const float inv = 1.f / fix_scale(*amax);                         // (exact: a power of two)
    dT[(size_t)(e / H) * ldT + e % H] += (float)((double)acc[e] * (double)inv);
}
int ghn3_bias_hist(float* dT, const float* dBias, const int* pair, int B, int N, int H, int V, void* scratch,
                   int have_amax, hipStream_t s) {
    if (!scratch) { ghn3_set_error("bias_hist: r3 (zeroed int64 [V*V*H] + 16 bytes of scratch) is required"); return GHN3_E_ARG; }
    const int ldT = (H + 3) & ~3;
    long long* acc = reinterpret_cast<long long*>(scratch);
    float* amax = reinterpret_cast<float*>(acc + (size_t)V * V * H);
    const int64_t n = (int64_t)B * H * N * N;
    // (have_amax: the launch that wrote the final dBias -- GHN3_OP_ATTN_BWD with r5 -- left max |dBias| in the scratch)
    if (!have_amax)
        hipLaunchKernelGGL(bias_amax_kernel, dim3((unsigned)std::min<int64_t>(1024, (n + 255) / 256)), dim3(256), 0, s, dBias,
                           n, amax);
    const size_t lds = (size_t)V * V * sizeof(long long);
    const int use_lds = lds <= 64 * 1024;
    const int rpb = 16;
    hipLaunchKernelGGL(bias_hist_kernel, dim3((N + rpb - 1) / rpb, H, B), dim3(256), use_lds ? lds : 0, s, acc, dBias,
                       pair, N, H, V, rpb, use_lds, amax);
    hipLaunchKernelGGL(bias_hist_finish_kernel, dim3((V * V * H + 255) / 256), dim3(256), 0, s, dT, acc, V * V, H, ldT,
                       amax);
    return launch_ok("bias_hist");
}

// ------------------------------------------------------------------------------------------------
// LayerNorm (F.layer_norm, graphormer.py:239,241 and nn.py:262-263), one wave per row.
// ------------------------------------------------------------------------------------------------
// Rows of up to LN_REG * 64 channels (every released GHN-3: C <= 384) are loaded ONCE into registers -- all loads in
// flight together -- and the statistics and the result are computed from there: on this latency-bound chain every
// further pass over the row was another ~1 us round trip (the row was just written by a GEMM on other XCDs, so
// even the first touch misses the local L2).  Same summation order as the wide-row loops: identical results.
// `add` (optional): second K-half plane of the GEMM that produced x (split-K over two workgroup sets, see
// program.py::_split_k): the row is x + add, written back to x for the later readers of the residual stream.
#define LN_REG 8
#define LN_PLANES 7          // addend planes whose loads are issued together (more are summed in a loop)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(float* __restrict__ y, float* __restrict__ x,
                                                            const float* __restrict__ g, const float* __restrict__ bta,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            const float* __restrict__ add, int n_add, int64_t add_stride,
                                                            int rows, int C, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* xr = x + (size_t)row * C;
    float* yr = y + (size_t)row * C;
    const float* ar = add ? add + (size_t)row * C : nullptr;
    if (C <= LN_REG * 64) {
        float xv[LN_REG], gv[LN_REG], bv[LN_REG];
#pragma unroll
        for (int j = 0; j < LN_REG; ++j) {
            const int c = lane + 64 * j;
            const bool in = c < C;
            xv[j] = in ? xr[c] : 0.f;
            gv[j] = in ? g[c] : 0.f;
            bv[j] = in ? bta[c] : 0.f;
        }
        if (ar) {
            // K-slice planes of the producing GEMM: every load of the row is issued before the first add (one round trip
            // for all planes: a loop with a run-time trip count serialised them), summed in plane order
            float pv[LN_PLANES][LN_REG];
#pragma unroll
            for (int p = 0; p < LN_PLANES; ++p)
#pragma unroll
                for (int j = 0; j < LN_REG; ++j) {
                    const int c = lane + 64 * j;
                    pv[p][j] = (p < n_add && c < C) ? ar[p * add_stride + c] : 0.f;
                }
#pragma unroll
            for (int p = 0; p < LN_PLANES; ++p)
#pragma unroll
                for (int j = 0; j < LN_REG; ++j) xv[j] += pv[p][j];
            for (int p = LN_PLANES; p < n_add; ++p)
                for (int j = 0; j < LN_REG; ++j) { const int c = lane + 64 * j; if (c < C) xv[j] += ar[p * add_stride + c]; }
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < LN_REG; ++j) if (lane + 64 * j < C) s += xv[j];
        const float mu = wsum(s) / (float)C;
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < LN_REG; ++j) if (lane + 64 * j < C) { const float d = xv[j] - mu; v += d * d; }
        const float rs = rsqrtf(wsum(v) / (float)C + eps);
#pragma unroll
        for (int j = 0; j < LN_REG; ++j) {
            const int c = lane + 64 * j;
            if (c < C) {
                yr[c] = (xv[j] - mu) * rs * gv[j] + bv[j];
                if (ar) xr[c] = xv[j];
            }
        }
        if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
        return;
    }
    if (ar)
        for (int c = lane; c < C; c += 64) {                          // (each lane re-reads only its own writes)
            float v = xr[c];
            for (int p = 0; p < n_add; ++p) v += ar[p * add_stride + c];
            xr[c] = v;
        }
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mu = wsum(s) / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { float d = xr[c] - mu; v += d * d; }
    const float rs = rsqrtf(wsum(v) / (float)C + eps);
    for (int c = lane; c < C; c += 64) yr[c] = (xr[c] - mu) * rs * g[c] + bta[c];
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
}
int ghn3_layernorm_fwd(float* y, float* x, const float* g, const float* b, float* mean, float* rstd, const float* add,
                       int n_add, int64_t add_stride, int rows, int C, float eps, hipStream_t s) {
    if (add && n_add <= 0) n_add = 1;
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, y, x, g, b, mean, rstd, add, n_add,
                       add_stride, rows, C, eps);
    return launch_ok("layernorm_fwd");
}

// `add` (optional): second K-half plane of the dgrad GEMM that produced dy; dy + add is written back to dy (the
// LayerNorm parameter gradients read it afterwards).
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(float* __restrict__ dx, float* __restrict__ dy,
                                                            const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ res,
                                                            const float* __restrict__ add, int n_add, int64_t add_stride,
                                                            int rows, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (size_t)row * C;
    float* gr = dy + (size_t)row * C;
    float* dr = dx + (size_t)row * C;
    const float* rr = res ? res + (size_t)row * C : nullptr;
    const float* ar = add ? add + (size_t)row * C : nullptr;
    const float mu = mean[row], rs = rstd[row];
    if (C <= LN_REG * 64) {
        float dgv[LN_REG], xh[LN_REG], rv[LN_REG];
#pragma unroll
        for (int j = 0; j < LN_REG; ++j) {
            const int c = lane + 64 * j;
            const bool in = c < C;
            dgv[j] = in ? gr[c] : 0.f;
            xh[j] = in ? xr[c] : 0.f;
            rv[j] = (rr && in) ? rr[c] : 0.f;
        }
        if (ar) {
            float pv[LN_PLANES][LN_REG];
#pragma unroll
            for (int p = 0; p < LN_PLANES; ++p)
#pragma unroll
                for (int j = 0; j < LN_REG; ++j) {
                    const int c = lane + 64 * j;
                    pv[p][j] = (p < n_add && c < C) ? ar[p * add_stride + c] : 0.f;
                }
#pragma unroll
            for (int p = 0; p < LN_PLANES; ++p)
#pragma unroll
                for (int j = 0; j < LN_REG; ++j) dgv[j] += pv[p][j];
            for (int p = LN_PLANES; p < n_add; ++p)
                for (int j = 0; j < LN_REG; ++j) { const int c = lane + 64 * j; if (c < C) dgv[j] += ar[p * add_stride + c]; }
#pragma unroll
            for (int j = 0; j < LN_REG; ++j) { const int c = lane + 64 * j; if (c < C) gr[c] = dgv[j]; }
        }
#pragma unroll
        for (int j = 0; j < LN_REG; ++j) { const int c = lane + 64 * j; dgv[j] *= (c < C) ? g[c] : 0.f; }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < LN_REG; ++j)
            if (lane + 64 * j < C) {
                s1 += dgv[j];
                s2 += dgv[j] * (xh[j] - mu) * rs;
            }
        s1 = wsum(s1) / (float)C;
        s2 = wsum(s2) / (float)C;
#pragma unroll
        for (int j = 0; j < LN_REG; ++j) {
            const int c = lane + 64 * j;
            if (c < C) {
                const float h = (xh[j] - mu) * rs;
                float v = rs * (dgv[j] - s1 - h * s2);
                if (rr) v += rv[j];
                dr[c] = v;
            }
        }
        return;
    }
    if (ar)
        for (int c = lane; c < C; c += 64) {
            float v = gr[c];
            for (int p = 0; p < n_add; ++p) v += ar[p * add_stride + c];
            gr[c] = v;
        }
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float dg = gr[c] * g[c];
        s1 += dg;
        s2 += dg * (xr[c] - mu) * rs;
    }
    s1 = wsum(s1) / (float)C;
    s2 = wsum(s2) / (float)C;
    for (int c = lane; c < C; c += 64) {
        const float xh = (xr[c] - mu) * rs;
        float v = rs * (gr[c] * g[c] - s1 - xh * s2);
        if (rr) v += rr[c];
        dr[c] = v;
    }
}
int ghn3_layernorm_bwd(float* dx, float* dy, const float* x, const float* g, const float* mean,
                       const float* rstd, const float* res, const float* add, int n_add, int64_t add_stride, int rows,
                       int C, hipStream_t s) {
    if (add && n_add <= 0) n_add = 1;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, dx, dy, x, g, mean, rstd, res, add,
                       n_add, add_stride, rows, C);
    return launch_ok("layernorm_bwd");
}

// dgamma[c] += sum_r dy[r,c] * xhat[r,c] ; dbeta[c] += sum_r dy[r,c].  Block = 64 columns x 4 row lanes over ALL rows
// (row r goes to lane r % 4, the four lane sums are added in a fixed order): one writer per column, no atomics,
// run-to-run identical sums.  (Off the critical path: it runs on the side stream.)
__global__ __launch_bounds__(256) void ln_param_grad_kernel(float* __restrict__ dg, float* __restrict__ db,
                                                            const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int rows, int C, int accum) {
    // 16 columns x 16 row lanes per workgroup: row r goes to lane r % 16 (8 rows of a lane in flight at a time), the 16
    // lane sums are added in a fixed order
    __shared__ float sg[16][16], sb[16][16];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float ag = 0.f, ab = 0.f;
    if (c < C) {
        for (int r0 = rl; r0 < rows; r0 += 16 * 8) {
            float d[8], xv[8], mu[8], rs[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = r0 + 16 * k;
                const bool in = r < rows;
                d[k] = in ? dy[(size_t)r * C + c] : 0.f;
                xv[k] = in ? x[(size_t)r * C + c] : 0.f;
                mu[k] = in ? mean[r] : 0.f;
                rs[k] = in ? rstd[r] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { ag += d[k] * (xv[k] - mu[k]) * rs[k]; ab += d[k]; }
        }
    }
    sg[rl][cl] = ag; sb[rl][cl] = ab;
    __syncthreads();
    if (rl == 0 && c < C) {
        float g = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { g += sg[k][cl]; b += sb[k][cl]; }
        dg[c] = accum ? dg[c] + g : g;
        db[c] = accum ? db[c] + b : b;
    }
}
// GHN3_OP_LN_PARAM_GRAD_BATCH: the same reduction for many LayerNorms in one launch (blockIdx.y = item)
__global__ __launch_bounds__(256) void ln_param_grad_batch_kernel(float* __restrict__ gbase, const float* __restrict__ abase,
                                                                  const int64_t* __restrict__ table, int rows, int C) {
    __shared__ float sg[16][16], sb[16][16];
    const int64_t* T = table + 6 * (int64_t)blockIdx.y;
    float* dg = gbase + T[0];
    float* db = gbase + T[1];
    const float* dy = abase + T[2];
    const float* x = abase + T[3];
    const float* mean = abase + T[4];
    const float* rstd = abase + T[5];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float ag = 0.f, ab = 0.f;
    if (c < C) {
        for (int r0 = rl; r0 < rows; r0 += 16 * 8) {
            float d[8], xv[8], mu[8], rs[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = r0 + 16 * k;
                const bool in = r < rows;
                d[k] = in ? dy[(size_t)r * C + c] : 0.f;
                xv[k] = in ? x[(size_t)r * C + c] : 0.f;
                mu[k] = in ? mean[r] : 0.f;
                rs[k] = in ? rstd[r] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { ag += d[k] * (xv[k] - mu[k]) * rs[k]; ab += d[k]; }
        }
    }
    sg[rl][cl] = ag; sb[rl][cl] = ab;
    __syncthreads();
    if (rl == 0 && c < C) {
        float g = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { g += sg[k][cl]; b += sb[k][cl]; }
        dg[c] += g;
        db[c] += b;
    }
}
int ghn3_ln_param_grad_batch(float* gbase, const float* abase, const int64_t* table, int n_items, int rows, int C,
                             hipStream_t s) {
    if (n_items <= 0 || rows <= 0 || C <= 0) return GHN3_OK;
    hipLaunchKernelGGL(ln_param_grad_batch_kernel, dim3((C + 15) / 16, n_items), dim3(256), 0, s, gbase, abase, table, rows, C);
    return launch_ok("ln_param_grad_batch");
}

int ghn3_ln_param_grad(float* dg, float* db, const float* dy, const float* x, const float* mean, const float* rstd,
                       int rows, int C, int accum, hipStream_t s) {
    hipLaunchKernelGGL(ln_param_grad_kernel, dim3((C + 15) / 16), dim3(256), 0, s, dg, db, dy, x, mean, rstd, rows, C,
                       accum);
    return launch_ok("ln_param_grad");
}

// GHN3_OP_ROWSET_COLSUM: bias gradient of a decoder linear from the stacked row sets of its output gradient,
//   out[o' * I + i0_s + i'] += sum_{sets with o' < o_s, i' < i_s} sum_{r < rows_s} X[off_s + r * ld_s + o' * i_s + i']
// (decoder.conv.2.bias: every decoder row contributes to the W2 rows o' < o_r, i' < i_r it consumed, nn.py:749-750;
// decoder.conv.0.bias: one set with o = 1).  Block = 64 columns i' of one o' x 4 row lanes, sets and rows in a fixed
// order, one writer per output: deterministic (the column sums fused into GHN3_OP_CAST16 used float atomics).
struct RowSet { int64_t off; int32_t rows, o, i, ld, i0, _pad; };
__global__ __launch_bounds__(256) void rowset_colsum_kernel(float* __restrict__ out, const float* __restrict__ X,
                                                            const RowSet* __restrict__ sets, int n_sets, int I) {
    __shared__ float red[4][64];
    const int l = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int ip = blockIdx.x * 64 + l, op = blockIdx.y;               // output column / row
    float acc = 0.f;
    for (int k = 0; k < n_sets; ++k) {
        const RowSet S = sets[k];
        const int il = ip - S.i0;                                      // column inside the set
        if (op >= S.o || (int)(blockIdx.x * 64 + 63) < S.i0 || (int)(blockIdx.x * 64) >= S.i0 + S.i) continue;   // (uniform)
        if (il >= 0 && il < S.i) {
            const float* p = X + S.off + (size_t)op * S.i + il;
            for (int r0 = rl; r0 < S.rows; r0 += 4 * 8) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int r = r0 + 4 * e; v[e] = r < S.rows ? p[(size_t)r * S.ld] : 0.f; }
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += v[e];
            }
        }
    }
    red[rl][l] = acc;
    __syncthreads();
    if (rl == 0 && ip < I) out[(size_t)op * I + ip] += (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
}
int ghn3_rowset_colsum(float* out, const float* X, const void* sets, int n_sets, int O, int I, hipStream_t s) {
    if (n_sets <= 0 || O <= 0 || I <= 0) return GHN3_OK;
    hipLaunchKernelGGL(rowset_colsum_kernel, dim3((I + 63) / 64, O), dim3(256), 0, s, out, X,
                       reinterpret_cast<const RowSet*>(sets), n_sets, I);
    return launch_ok("rowset_colsum");
}

// ------------------------------------------------------------------------------------------------
// Tile + normalise + write  (nn.py:422-506, 554-592, 508-552 fused; SURVEY 8(a) closed form).
// The host pre-chunks the work: block k handles `chunk` consecutive elements of one descriptor; the
// (descriptor, start) pairs are stored after the descriptors: int64 pairs.
// ------------------------------------------------------------------------------------------------
#define TILE_CHUNK 2048

__device__ __forceinline__ float norm_apply(float v, int mode, float scale) {
    if (mode == 0) return v * scale;
    if (mode == 1) return 2.f / (1.f + __expf(-0.5f * v));
    return tanhf(0.2f * v);
}
__device__ __forceinline__ float norm_grad(float v, int mode, float scale) {
    if (mode == 0) return scale;
    if (mode == 1) { float sg = 1.f / (1.f + __expf(-0.5f * v)); return sg * (1.f - sg); }
    float th = tanhf(0.2f * v);
    return 0.2f * (1.f - th * th);
}

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access at dword alignment
typedef unsigned short us4h __attribute__((ext_vector_type(4)));
struct SrcTable { const float* p[6]; };
struct DstTable { float* p[6]; };

// Mixed-radix "odometer": an element index is decoded into its 4 digits once per thread; every further element
// of the thread (stride 256) adds the pre-decoded digits of 256 with carries, so the steady state has no integer
// division (the HBM-bound tile kernels were ALU-bound on 64-bit div/mod before).
struct Odo {
    int d[4];      // digits, d[3] fastest
    int T[4];      // radices
    int s[4];      // digits of the stride
    __device__ __forceinline__ void init(unsigned e, unsigned stride, int t0, int t1, int t2, int t3) {
        T[0] = t0; T[1] = t1; T[2] = t2; T[3] = t3;
        unsigned q = e;
        d[3] = q % t3; q /= t3; d[2] = q % t2; q /= t2; d[1] = q % t1; d[0] = q / t1;
        q = stride;
        s[3] = q % t3; q /= t3; s[2] = q % t2; q /= t2; s[1] = q % t1; s[0] = q / t1;
    }
    __device__ __forceinline__ void step() {
        int c = 0;
#pragma unroll
        for (int k = 3; k >= 1; --k) {
            int v = d[k] + s[k] + c;
            c = v >= T[k];
            d[k] = c ? v - T[k] : v;
        }
        d[0] += s[0] + c;
    }
};

// Row blocks (descriptor index stored as ~k < 0, second word = a0): convolution kernels with kh * kw > 1.  The target
// [o][i][kh][kw] is contiguous in (i, kh, kw) while the source keeps one [o][i] matrix per kernel position, so the
// element-wise mapping above reads (backward: writes) one side with a stride of kh * kw floats.  A row block moves
// one o-row of one tensor through LDS instead: T1 * kh * kw contiguous target floats on one side, kh * kw
// segments of consecutive i on the other -- both sides coalesced.  Requires mode 0, S[1] == 1 and no tiling
// along kh / kw (the host checks).
// sq_parts (optional): slot blockIdx.x receives the sum of squares of everything this block wrote (fixed reduction
// order): the per-tensor Frobenius norms of the predicted-parameter loss without another pass over the output.
// b_parts (optional, with sq_parts): slot blockIdx.x receives  max |written value| * replicas * |scale|  for blocks of mode-0
// descriptors of source buffer 0 (0 otherwise): an a-priori bound of the tile backward's output, see tile_bwd_kernel.
__device__ __forceinline__ void tile_block_sumsq(float ss, float bm, float* __restrict__ sq_parts, float* __restrict__ b_parts,
                                                 const ghn3_tile_desc* __restrict__ D) {
    __shared__ float sq_red[4], mx_red[4];
    ss = wsum(ss);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bm = fmaxf(bm, __shfl_xor(bm, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sq_red[threadIdx.x >> 6] = ss; mx_red[threadIdx.x >> 6] = bm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sq_parts[blockIdx.x] = (sq_red[0] + sq_red[1]) + (sq_red[2] + sq_red[3]);
        if (b_parts) {
            float f = 0.f;
            if (D->src_buf == 0 && D->mode == 0) {
                f = fabsf(D->scale);
#pragma unroll
                for (int k = 0; k < 4; ++k) f *= (float)((D->T[k] + D->E[k] - 1) / D->E[k]);
            }
            b_parts[blockIdx.x] = fmaxf(fmaxf(mx_red[0], mx_red[1]), fmaxf(mx_red[2], mx_red[3])) * f;
        }
    }
}

__global__ __launch_bounds__(256) void tile_fwd_kernel(float* __restrict__ flat, SrcTable srcs,
                                                       const ghn3_tile_desc* __restrict__ desc,
                                                       const int64_t* __restrict__ blocks, float* __restrict__ sq_parts,
                                                       float* __restrict__ b_parts) {
    extern __shared__ float tl[];
    float ss = 0.f, bm = 0.f;
    const int64_t di_raw = blocks[2 * (size_t)blockIdx.x];
    if (di_raw < 0) {
        const ghn3_tile_desc* D = desc + (~di_raw);
        const int64_t word = blocks[2 * (size_t)blockIdx.x + 1];
        const int a0 = (int)(word & 0xffffff), i0 = (int)(word >> 24);          // row, first i of the chunk
        const int T1 = D->T[1], T3 = D->T[3], hw = D->T[2] * T3, E1 = D->E[1];
        const int ni = min(D->_pad, T1 - i0);                                   // D->_pad = i per row block
        const float* src = srcs.p[D->src_buf] + D->src_off + (int64_t)(a0 % D->E[0]) * D->S[0];
        const int n = ni * hw;
        const int S2 = (int)D->S[2], S3 = (int)D->S[3];
        const float inv_ni = 1.f / (float)ni, inv_t3 = 1.f / (float)T3;
        // Both sides move 16 bytes per lane: four consecutive i of one kernel position on the source side (when the
        // chunk, the strides and the base allow it), four consecutive target floats on the other (dword-aligned
        // global_store_dwordx4: a tensor's o-row starts at any multiple of kh * kw floats).
        const bool vec = E1 == T1 && (ni & 3) == 0 && (S2 & 3) == 0 && (S3 & 3) == 0 &&
                         ((reinterpret_cast<uintptr_t>(src + i0)) & 15) == 0;
        if (vec) {
            const int ni4 = ni >> 2, n4 = ni4 * hw;
            const float inv_ni4 = 1.f / (float)ni4;
            constexpr int U = 4;
            for (int b0 = threadIdx.x; b0 < n4; b0 += 256 * U) {
                float4 v[U];
                int li[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = b0 + 256 * u;              // (p, i / 4) with i fastest
                    int pp = (int)((float)idx * inv_ni4);
                    int i4 = idx - pp * ni4;
                    if (i4 < 0) { i4 += ni4; --pp; } else if (i4 >= ni4) { i4 -= ni4; ++pp; }
                    int y = (int)((float)pp * inv_t3);
                    int x = pp - y * T3;
                    if (x < 0) { x += T3; --y; } else if (x >= T3) { x -= T3; ++y; }
                    li[u] = i4 * 4 * hw + pp;
                    v[u] = idx < n4 ? *reinterpret_cast<const float4*>(src + (int64_t)y * S2 + (int64_t)x * S3 + i0 + i4 * 4)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (b0 + 256 * u < n4) {
                        tl[li[u]] = v[u].x; tl[li[u] + hw] = v[u].y; tl[li[u] + 2 * hw] = v[u].z; tl[li[u] + 3 * hw] = v[u].w;
                    }
            }
        } else {
            constexpr int U = 8;                                  // loads in flight per thread
            for (int b0 = threadIdx.x; b0 < n; b0 += 256 * U) {
                float v[U];
                int li[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = b0 + 256 * u;                  // (p, i) with i fastest: consecutive source floats
                    int pp = (int)((float)idx * inv_ni);
                    int i = idx - pp * ni;
                    if (i < 0) { i += ni; --pp; } else if (i >= ni) { i -= ni; ++pp; }
                    int y = (int)((float)pp * inv_t3);
                    int x = pp - y * T3;
                    if (x < 0) { x += T3; --y; } else if (x >= T3) { x -= T3; ++y; }
                    li[u] = i * hw + pp;
                    const int si = i0 + i;
                    v[u] = idx < n ? src[(int64_t)y * S2 + (int64_t)x * S3 + (E1 == T1 ? si : si % E1)] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (b0 + 256 * u < n) tl[li[u]] = v[u];
            }
        }
        __syncthreads();
        float* dst = flat + D->dst_off + ((int64_t)a0 * T1 + i0) * hw;
        const float scale = D->scale;
        const int nv = n & ~3;
        for (int j = threadIdx.x * 4; j < nv; j += 1024) {
            f4u o;
            o.x = tl[j] * scale; o.y = tl[j + 1] * scale; o.z = tl[j + 2] * scale; o.w = tl[j + 3] * scale;
            *reinterpret_cast<f4u*>(dst + j) = o;
            ss += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
            bm = fmaxf(fmaxf(bm, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
        for (int j = nv + threadIdx.x; j < n; j += 256) { const float o = tl[j] * scale; dst[j] = o; ss += o * o; bm = fmaxf(bm, fabsf(o)); }
        if (sq_parts) tile_block_sumsq(ss, bm, sq_parts, b_parts, D);
        return;
    }
    const int64_t di = di_raw;
    const unsigned start = (unsigned)blocks[2 * (size_t)blockIdx.x + 1];
    const ghn3_tile_desc* D = desc + di;
    const int T0 = D->T[0], T1 = D->T[1], T2 = D->T[2], T3 = D->T[3];
    const unsigned numel = (unsigned)T0 * T1 * T2 * T3;
    const unsigned end = min(numel, start + TILE_CHUNK);
    const float* src = srcs.p[D->src_buf] + D->src_off;
    float* dst = flat + D->dst_off;
    const int E0 = D->E[0], E1 = D->E[1], E2 = D->E[2], E3 = D->E[3];
    const bool w0 = E0 != T0, w1 = E1 != T1, w2 = E2 != T2, w3 = E3 != T3;   // does the dimension wrap (tile)?
    const int S0 = (int)D->S[0], S1 = (int)D->S[1], S2 = (int)D->S[2], S3 = (int)D->S[3];
    const int mode = D->mode;
    const float scale = D->scale;
    Odo o;
    unsigned e = start + threadIdx.x;
    o.init(e, 256, T0, T1, T2, T3);
#pragma unroll 2
    for (; e < end; e += 256) {
        const int m0 = w0 ? o.d[0] % E0 : o.d[0];
        const int m1 = w1 ? o.d[1] % E1 : o.d[1];
        const int m2 = w2 ? o.d[2] % E2 : o.d[2];
        const int m3 = w3 ? o.d[3] % E3 : o.d[3];
        const int64_t so = (int64_t)m0 * S0 + (int64_t)m1 * S1 + (int64_t)m2 * S2 + (int64_t)m3 * S3;
        const float val = norm_apply(src[so], mode, scale);
        dst[e] = val;
        ss += val * val;
        bm = fmaxf(bm, fabsf(val));
        o.step();
    }
    if (sq_parts) tile_block_sumsq(ss, bm, sq_parts, b_parts, D);
}

int ghn3_tile_fwd(float* flat, const float* const* srcs, const ghn3_tile_desc* d_desc, int n_desc, int64_t total,
                  const int64_t* blocks, int lds_bytes, float* sq_parts, float* b_parts, hipStream_t s) {
    // `total` = number of work blocks in the (descriptor, start) table `blocks`.
    if (n_desc <= 0 || total <= 0) return GHN3_OK;
    SrcTable st;
    for (int i = 0; i < 6; ++i) st.p[i] = srcs[i];
    if (lds_bytes > 48 * 1024) {
        if (lds_bytes > 128 * 1024) { ghn3_set_error("tile_fwd: row blocks need %d bytes of LDS", lds_bytes); return GHN3_E_LIMIT; }
        hipFuncSetAttribute((const void*)tile_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    }
    hipLaunchKernelGGL(tile_fwd_kernel, dim3((unsigned)total), dim3(256), lds_bytes, s, flat, st, d_desc, blocks, sq_parts,
                       sq_parts ? b_parts : nullptr);
    return launch_ok("tile_fwd");
}

// Backward: source-centric.  Each source element of the descriptor's region R0 x R1 x R2 x R3 receives
// norm'(src) * sum over the target replicas (zero outside the consumed E region).  Iteration order over the
// region: dimension 1 fastest (the source-contiguous axis for decoder tiles), then 3, 2, 0.
// Fused predicted-parameter-norm loss (trainer.py:97-98,288-294): with `norms` the upstream gradient of element e of the
// descriptor's tensor t is  dflat[e] + (g / norm_t) out[e]  (dflat optional) -- the norm term's gradient is formed from the
// predicted values themselves instead of being materialised by a pass over the 346 MB output.
// acc[u] += p[b0 + 1024 u .. + 3] (elements below lim), u < U.
// (Measured, round 4: issuing the U loads unconditionally and together -- as compiled here every load sits in its own
// if-block and is waited for there -- costs 36 more VGPRs, occupancy 7 -> 4 waves per SIMD, and the kernel went from 0.195 to
// 0.239 ms: this pass lives on the number of blocks in flight, not on the round trips of one block.)
template <int U>
__device__ __forceinline__ void tile_bwd_gather(const float* __restrict__ p, int b0, int lim, f4u (&acc)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = b0 + 1024 * u;
        if (j + 4 <= lim) {
            acc[u] += *reinterpret_cast<const f4u*>(p + j);
        } else {
            if (j < lim) acc[u].x += p[j];
            if (j + 1 < lim) acc[u].y += p[j + 1];
            if (j + 2 < lim) acc[u].z += p[j + 2];
        }
    }
}
struct NormLoss { const float* out; const float* norms; const int* desc_seg; const float* g; };
// Direct 16-bit gradient tiles (fused norm loss only, no dflat): the gradient of source buffer 0 -- the decoder tiles, whose
// only consumers are the 16-bit W2 dgrad / wgrad GEMMs -- is written ONCE, as the scaled f16 / bf16 operand copy, instead of as
// fp32 followed by a cast pass that first needs the measured maximum.  The power-of-two scale comes from an a-priori bound:
// an element of the tile gradient is  (g / ||p_t||) * scale * (sum of the predicted values of its replicas), at most
// |g| * max_t( max|p_t| * replicas * |scale| / ||p_t|| ) = |g| * norms[-1] (GHN3_OP_PARAM_NORM_FIN).  The bound is what the
// consumers find in the amax slot (block 0 stores it), so copies and GEMM epilogues agree on the scale; it can never
// overflow and is within a few binades of the true maximum (f16 has 29 normal binades, the gradients span ~13).
// tab: per descriptor {h, rel0, ld32 | ld16 << 32}: the descriptor's region starts `rel0` floats into a row-major fp32 matrix
// with row stride ld32 whose 16-bit copy (row stride ld16) starts h 16-bit elements behind dsrcs.p[0]; h = INT64_MIN: the
// descriptor keeps its fp32 output.
struct H16 { const int64_t* tab; int bf16; };
__device__ __forceinline__ unsigned short h16_cast(float x, int bf16) {
    if (bf16) {
        const unsigned u = __float_as_uint(x);
        return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
    return __builtin_bit_cast(unsigned short, (_Float16)x);
}
__device__ __forceinline__ int64_t h16_index(unsigned rel, unsigned ld32, unsigned ld16) {
    const unsigned r = rel / ld32;
    return (int64_t)r * ld16 + (rel - r * ld32);
}
__global__ __launch_bounds__(256) void tile_bwd_kernel(const float* __restrict__ dflat, SrcTable srcs, DstTable dsrcs,
                                                       const ghn3_tile_desc* __restrict__ desc,
                                                       const int64_t* __restrict__ blocks, float* __restrict__ amax,
                                                       NormLoss nl_, H16 h16) {
    extern __shared__ float tl[];
    float mx = 0.f;                                      // running max |x| written to source-grad buffer 0
    const int64_t di_raw = blocks[2 * (size_t)blockIdx.x];
    const int64_t di_any = di_raw < 0 ? ~di_raw : di_raw;
    float kn = 0.f;                                      // g / ||p_t|| of this block's tensor (0: no norm term)
    if (nl_.norms) {
        const float nrm = nl_.norms[nl_.desc_seg[di_any]];
        kn = nrm > 0.f ? nl_.g[0] / nrm : 0.f;
    }
    // direct 16-bit output of this block's descriptor?
    unsigned short* hdst = nullptr;
    unsigned hrel0 = 0, hld32 = 1, hld16 = 0;
    float hsc = 1.f;
    if (h16.tab) {
        const float bnd = nl_.norms[-1] * fabsf(nl_.g[0]);
        if (blockIdx.x == 0 && threadIdx.x == 0 && amax) amax[0] = bnd;
        const int64_t h = h16.tab[3 * di_any];
        if (h != INT64_MIN && desc[di_any].src_buf == 0) {
            hdst = reinterpret_cast<unsigned short*>(dsrcs.p[0]) + h;
            hrel0 = (unsigned)h16.tab[3 * di_any + 1];
            const int64_t lds_ = h16.tab[3 * di_any + 2];
            hld32 = (unsigned)(lds_ & 0xffffffff); hld16 = (unsigned)(lds_ >> 32);
            hsc = ghn3_pow2_scale(bnd);
        }
    }
    const float* __restrict__ outp = nl_.norms ? nl_.out + desc[di_any].dst_off : nullptr;
    if (di_raw < 0) {
        // row block (see tile_fwd_kernel): source row a0 of the region; replicas along o and i are summed in LDS
        const ghn3_tile_desc* D = desc + (~di_raw);
        const int64_t word = blocks[2 * (size_t)blockIdx.x + 1];
        const int a0 = (int)(word & 0xffffff), i0 = (int)(word >> 24);          // source row, first i of the chunk
        const int T0 = D->T[0], T1 = D->T[1], T3 = D->T[3], hw = D->T[2] * T3;
        const int E0 = D->E[0], E1 = D->E[1], R1 = D->R[1];
        const int ni = min(D->_pad, R1 - i0);                                   // i of this block (source side)
        const int nl = max(0, min(ni, E1 - i0));                                // of which inside the consumed region
        float* dsrc = dsrcs.p[D->src_buf] + D->src_off + (int64_t)a0 * D->S[0] + i0;
        const float* g = dflat ? dflat + D->dst_off : nullptr;
        const bool live = a0 < E0 && nl > 0;
        if (live) {
            const int n = nl * hw;
            constexpr int U = 4;                              // 16-byte loads in flight per thread (dword aligned)
            for (int b0 = threadIdx.x * 4; b0 < n; b0 += 1024 * U) {
                // (the norm term is linear: the predicted values are summed on their own and scaled by g / ||p|| once at
                // the end, so the two dependent loads behind `kn` are not waited for inside the replica loop)
                f4u acc[U], acc2[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { acc[u] = f4u{0.f, 0.f, 0.f, 0.f}; acc2[u] = f4u{0.f, 0.f, 0.f, 0.f}; }
                for (int t0 = a0; t0 < T0; t0 += E0)
                    for (int r0 = 0; r0 + i0 < T1; r0 += E1) {              // replicas along i
                        const int lim = min(nl, T1 - r0 - i0) * hw;          // a last partial replica covers fewer i
                        const int64_t go = ((int64_t)t0 * T1 + r0 + i0) * hw;
                        if (g) tile_bwd_gather<U>(g + go, b0, lim, acc);
                        if (outp) tile_bwd_gather<U>(outp + go, b0, lim, acc2);
                    }
                if (outp) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        acc[u].x += kn * acc2[u].x; acc[u].y += kn * acc2[u].y; acc[u].z += kn * acc2[u].z; acc[u].w += kn * acc2[u].w;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = b0 + 1024 * u;
                    if (j + 4 <= n) {
                        *reinterpret_cast<float4*>(tl + j) = make_float4(acc[u].x, acc[u].y, acc[u].z, acc[u].w);
                    } else {
                        if (j < n) tl[j] = acc[u].x;
                        if (j + 1 < n) tl[j + 1] = acc[u].y;
                        if (j + 2 < n) tl[j + 2] = acc[u].z;
                    }
                }
            }
            __syncthreads();
        }
        const float scale = D->scale;
        const int S2 = (int)D->S[2], S3 = (int)D->S[3];
        const float inv_t3 = 1.f / (float)T3;
        if ((ni & 3) == 0 && (S2 & 3) == 0 && (S3 & 3) == 0 && (reinterpret_cast<uintptr_t>(dsrc) & 15) == 0) {
            const int ni4 = ni >> 2, nw4 = ni4 * hw;
            const float inv_ni4 = 1.f / (float)ni4;
            for (int idx = threadIdx.x; idx < nw4; idx += 256) {    // (p, i / 4) with i fastest: 16 bytes per lane
                int pp = (int)((float)idx * inv_ni4);
                int i4 = idx - pp * ni4;
                if (i4 < 0) { i4 += ni4; --pp; } else if (i4 >= ni4) { i4 -= ni4; ++pp; }
                int y = (int)((float)pp * inv_t3);
                int x = pp - y * T3;
                if (x < 0) { x += T3; --y; } else if (x >= T3) { x -= T3; ++y; }
                const int i = i4 * 4;
                float4 val;
                val.x = (live && i < nl) ? tl[i * hw + pp] * scale : 0.f;
                val.y = (live && i + 1 < nl) ? tl[(i + 1) * hw + pp] * scale : 0.f;
                val.z = (live && i + 2 < nl) ? tl[(i + 2) * hw + pp] * scale : 0.f;
                val.w = (live && i + 3 < nl) ? tl[(i + 3) * hw + pp] * scale : 0.f;
                if (hdst) {
                    // (rel is a multiple of 4 like the fp32 address, ld32 % 4 == 0: the four columns stay in one row)
                    unsigned short* hp = hdst + h16_index(hrel0 + (unsigned)(a0 * (int)D->S[0] + i0 + y * S2 + x * S3 + i), hld32, hld16);
                    us4h o;
                    o[0] = h16_cast(val.x * hsc, h16.bf16); o[1] = h16_cast(val.y * hsc, h16.bf16);
                    o[2] = h16_cast(val.z * hsc, h16.bf16); o[3] = h16_cast(val.w * hsc, h16.bf16);
                    if ((reinterpret_cast<uintptr_t>(hp) & 7) == 0) *reinterpret_cast<us4h*>(hp) = o;
                    else { hp[0] = o[0]; hp[1] = o[1]; hp[2] = o[2]; hp[3] = o[3]; }
                    continue;
                }
                *reinterpret_cast<float4*>(dsrc + (int64_t)y * S2 + (int64_t)x * S3 + i) = val;
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(val.x), fabsf(val.y))), fmaxf(fabsf(val.z), fabsf(val.w)));
            }
        } else {
            const float inv_ni = 1.f / (float)ni;
            const int nw = ni * hw;
            for (int idx = threadIdx.x; idx < nw; idx += 256) {    // (p, i) with i fastest: consecutive source floats
                int pp = (int)((float)idx * inv_ni);
                int i = idx - pp * ni;
                if (i < 0) { i += ni; --pp; } else if (i >= ni) { i -= ni; ++pp; }
                int y = (int)((float)pp * inv_t3);
                int x = pp - y * T3;
                if (x < 0) { x += T3; --y; } else if (x >= T3) { x -= T3; ++y; }
                const float val = (live && i < nl) ? tl[i * hw + pp] * scale : 0.f;
                if (hdst) {
                    hdst[h16_index(hrel0 + (unsigned)(a0 * (int)D->S[0] + i0 + y * S2 + x * S3 + i), hld32, hld16)] = h16_cast(val * hsc, h16.bf16);
                    continue;
                }
                dsrc[(int64_t)y * S2 + (int64_t)x * S3 + i] = val;
                mx = fmaxf(mx, fabsf(val));
            }
        }
        if (amax && D->src_buf == 0 && !h16.tab) ghn3_atomic_amax(amax, mx);
        return;
    }
    const int64_t di = di_raw;
    const unsigned start = (unsigned)blocks[2 * (size_t)blockIdx.x + 1];
    const ghn3_tile_desc* D = desc + di;
    const int R0 = D->R[0], R1 = D->R[1], R2 = D->R[2], R3 = D->R[3];
    const unsigned numel = (unsigned)R0 * R1 * R2 * R3;
    const unsigned end = min(numel, start + TILE_CHUNK);
    const float* src = srcs.p[D->src_buf] + D->src_off;
    float* dsrc = dsrcs.p[D->src_buf] + D->src_off;
    const float* g = dflat ? dflat + D->dst_off : nullptr;
    const int T0 = D->T[0], T1 = D->T[1], T2 = D->T[2], T3 = D->T[3];
    const int E0 = D->E[0], E1 = D->E[1], E2 = D->E[2], E3 = D->E[3];
    const int S0 = (int)D->S[0], S1 = (int)D->S[1], S2 = (int)D->S[2], S3 = (int)D->S[3];
    const int mode = D->mode;
    const float scale = D->scale;
    // iteration order over the source region: a1 fastest, then a3, a2, a0  -> odometer digits (a0, a2, a3, a1)
    Odo o;
    unsigned e = start + threadIdx.x;
    o.init(e, 256, R0, R2, R3, R1);
    for (; e < end; e += 256) {
        const int a0 = o.d[0], a2 = o.d[1], a3 = o.d[2], a1 = o.d[3];
        const int64_t so = (int64_t)a0 * S0 + (int64_t)a1 * S1 + (int64_t)a2 * S2 + (int64_t)a3 * S3;
        float acc = 0.f, acc2 = 0.f;
        if (a0 < E0 && a1 < E1 && a2 < E2 && a3 < E3) {
            for (int t0 = a0; t0 < T0; t0 += E0)
                for (int t1 = a1; t1 < T1; t1 += E1)
                    for (int t2 = a2; t2 < T2; t2 += E2)
                        for (int t3 = a3; t3 < T3; t3 += E3) {
                            const int64_t gi = (((int64_t)t0 * T1 + t1) * T2 + t2) * T3 + t3;
                            if (g) acc += g[gi];
                            if (outp) acc2 += outp[gi];
                        }
            acc = (acc + kn * acc2) * norm_grad(src[so], mode, scale);
        }
        if (hdst) hdst[h16_index(hrel0 + (unsigned)so, hld32, hld16)] = h16_cast(acc * hsc, h16.bf16);
        else dsrc[so] = acc;
        mx = fmaxf(mx, fabsf(acc));
        o.step();
    }
    if (amax && D->src_buf == 0 && !h16.tab) ghn3_atomic_amax(amax, mx);
}

int ghn3_tile_bwd(const float* dflat, const float* const* srcs, float* const* dsrcs, const ghn3_tile_desc* d_desc,
                  int n_desc, int64_t total, const int64_t* blocks, int lds_bytes, float* amax, const float* out,
                  const float* norms, const int* desc_seg, const float* gscale, const int64_t* h16_tab, int h16_bf16,
                  hipStream_t s) {
    if (n_desc <= 0 || total <= 0) return GHN3_OK;
    if (!dflat && !norms) { ghn3_set_error("tile_bwd: neither an upstream gradient nor the fused norm loss"); return GHN3_E_ARG; }
    if (norms && (!out || !desc_seg || !gscale)) { ghn3_set_error("tile_bwd: the fused norm loss needs out, the descriptor -> tensor table and g"); return GHN3_E_ARG; }
    if (h16_tab && (dflat || !norms || !amax)) { ghn3_set_error("tile_bwd: direct 16-bit tiles need the fused norm loss alone (no upstream gradient) and the amax slot"); return GHN3_E_ARG; }
    NormLoss nl{out, norms, desc_seg, gscale};
    H16 h16{h16_tab, h16_bf16};
    SrcTable st; DstTable dt;
    for (int i = 0; i < 6; ++i) { st.p[i] = srcs[i]; dt.p[i] = dsrcs[i]; }
    if (lds_bytes > 48 * 1024) {
        if (lds_bytes > 128 * 1024) { ghn3_set_error("tile_bwd: row blocks need %d bytes of LDS", lds_bytes); return GHN3_E_LIMIT; }
        hipFuncSetAttribute((const void*)tile_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    }
    hipLaunchKernelGGL(tile_bwd_kernel, dim3((unsigned)total), dim3(256), lds_bytes, s, dflat, st, dt, d_desc, blocks,
                       amax, nl, h16);
    return launch_ok("tile_bwd");
}

// ------------------------------------------------------------------------------------------------
// loss = sum over predicted tensors of ||p||_F  (trainer.py:97-98,288-294).  seg_off holds (begin, end) pairs;
// norms[] must be zero on entry (it accumulates squared sums, then is square-rooted in place).
// ------------------------------------------------------------------------------------------------
// The segments are laid out back to back in the flat buffer (16-float aligned), so both passes stream it: workgroup b
// owns the floats [b * NORM_CHUNK, (b + 1) * NORM_CHUNK) and visits the (few) segments that intersect them, found by a
// binary search over the sorted (begin, end) table.
#define NORM_CHUNK 8192
__device__ __forceinline__ int first_segment_after(const int64_t* __restrict__ seg_off, int n_seg, int64_t pos) {
    int lo = 0, hi = n_seg;                               // first segment whose end is > pos
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (seg_off[2 * mid + 1] > pos) hi = mid; else lo = mid + 1;
    }
    return lo;
}
// Deterministic: with the host table (first_seg) a chunk writes the partial sum of its k-th segment to slot
// first_slot[chunk] + k of `parts` (first_slot[b] = first_seg[b] + b: a chunk touches segments first_seg[b] ..
// first_seg[b + 1] at most), and param_sqrt_kernel adds a segment's slots in chunk order.  Without the table the partial
// sums are added with float atomics.
__global__ __launch_bounds__(256) void param_sq_kernel(const float* __restrict__ flat,
                                                       const int64_t* __restrict__ seg_off, float* __restrict__ norms,
                                                       int n_seg, const int* __restrict__ first_seg,
                                                       float* __restrict__ parts) {
    __shared__ float red[4];
    const int64_t w0 = (int64_t)blockIdx.x * NORM_CHUNK, w1 = w0 + NORM_CHUNK;
    // (first_seg: host-computed first segment of every chunk -- saves the 8 dependent loads of the search)
    const int sg0 = first_seg ? first_seg[blockIdx.x] : first_segment_after(seg_off, n_seg, w0);
    for (int sgm = sg0; sgm < n_seg && seg_off[2 * sgm] < w1; ++sgm) {
        const int64_t c0 = max(seg_off[2 * sgm], w0), c1 = min(seg_off[2 * sgm + 1], w1);
        float acc = 0.f;
        const int64_t a0 = (c0 + 3) & ~(int64_t)3, a1 = c1 & ~(int64_t)3;     // 16-byte aligned interior
        if (a0 < a1) {
            const float4* f4 = reinterpret_cast<const float4*>(flat + a0);
            const int64_t n4 = (a1 - a0) >> 2;
            for (int64_t q = threadIdx.x; q < n4; q += 256) {
                const float4 v = f4[q];
                acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
            for (int64_t e = c0 + threadIdx.x; e < a0; e += 256) { const float v = flat[e]; acc += v * v; }
            for (int64_t e = a1 + threadIdx.x; e < c1; e += 256) { const float v = flat[e]; acc += v * v; }
        } else {
            for (int64_t e = c0 + threadIdx.x; e < c1; e += 256) { const float v = flat[e]; acc += v * v; }
        }
        acc = wsum(acc);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float t = (red[0] + red[1]) + (red[2] + red[3]);
            if (parts && first_seg) parts[sg0 + (int64_t)blockIdx.x + (sgm - sg0)] = t;
            else if (t != 0.f) atomicAdd(&norms[sgm], t);
        }
    }
}
// one wave per tensor: its slots (chunk order) are summed by the 64 lanes (lane l takes chunks l, l + 64, ...) and a fixed
// shuffle tree; a second single-workgroup pass adds the norms in a fixed order
__global__ __launch_bounds__(256) void param_sqrt_kernel(float* __restrict__ norms, int n_seg,
                                                         const int64_t* __restrict__ seg_off,
                                                         const int* __restrict__ first_seg,
                                                         const float* __restrict__ parts) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n_seg) return;
    float sq = 0.f;
    if (parts && first_seg) {
        if (seg_off[2 * i + 1] > seg_off[2 * i]) {
            const int64_t b0 = seg_off[2 * i] / NORM_CHUNK, b1 = (seg_off[2 * i + 1] - 1) / NORM_CHUNK;
            for (int64_t b = b0 + lane; b <= b1; b += 64) sq += parts[b + i];
        }
        sq = wsum(sq);
    } else {
        sq = norms[i];
    }
    if (lane == 0) norms[i] = sqrtf(sq);
}
__global__ void param_loss_kernel(float* __restrict__ loss, const float* __restrict__ norms, int n_seg) {
    __shared__ float tot[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_seg; i += 256) acc += norms[i];
    tot[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < 256; ++k) t += tot[k];
        loss[0] += t;
    }
}
// GHN3_OP_PARAM_NORM_FIN: norms from the per-work-block sums of squares tile_fwd left; one wave per tensor, its blocks
// in block order by the 64 lanes + a fixed shuffle tree; then the loss = sum of the norms in a fixed order
__global__ __launch_bounds__(256) void param_norm_fin_kernel(float* __restrict__ norms, int n_seg,
                                                             const float* __restrict__ parts,
                                                             const int* __restrict__ first,
                                                             const float* __restrict__ b_parts, float* __restrict__ ratio) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n_seg) return;
    float sq = 0.f, bm = 0.f;
    for (int b = first[i] + lane; b < first[i + 1]; b += 64) { sq += parts[b]; if (b_parts) bm = fmaxf(bm, b_parts[b]); }
    sq = wsum(sq);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bm = fmaxf(bm, __shfl_xor(bm, o, 64));
    if (lane == 0) {
        const float nrm = sqrtf(sq);
        norms[i] = nrm;
        if (ratio) ratio[i] = nrm > 0.f ? bm / nrm : 0.f;
    }
}
// loss = sum of the norms (fixed order); bound = max_t (bound part of t) / ||p_t||: times |g| an upper bound of every element the
// tile backward writes to source-gradient buffer 0 under the fused norm loss (see tile_bwd_kernel)
__global__ void param_loss_set_kernel(float* __restrict__ loss, const float* __restrict__ norms, int n_seg,
                                      const float* __restrict__ ratio, float* __restrict__ bound) {
    __shared__ float tot[256], mx[256];
    float acc = 0.f, m = 0.f;
    for (int i = threadIdx.x; i < n_seg; i += 256) { acc += norms[i]; if (ratio) m = fmaxf(m, ratio[i]); }
    tot[threadIdx.x] = acc;
    mx[threadIdx.x] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f, b = 0.f;
        for (int k = 0; k < 256; ++k) { t += tot[k]; b = fmaxf(b, mx[k]); }
        loss[0] = t;
        if (bound) bound[0] = b;
    }
}
int ghn3_param_norm_fin(float* loss, float* norms, const float* parts, const int* first, int n_seg, const float* b_parts,
                        float* ratio, float* bound, hipStream_t s) {
    if (n_seg <= 0) return GHN3_OK;
    if (b_parts && !(ratio && bound)) { ghn3_set_error("PARAM_NORM_FIN: bound parts need the ratio scratch (r5) and the bound slot (r6)"); return GHN3_E_ARG; }
    if (!b_parts) ratio = nullptr, bound = nullptr;
    hipLaunchKernelGGL(param_norm_fin_kernel, dim3((n_seg + 3) / 4), dim3(256), 0, s, norms, n_seg, parts, first, b_parts, ratio);
    hipLaunchKernelGGL(param_loss_set_kernel, dim3(1), dim3(256), 0, s, loss, norms, n_seg, ratio, bound);
    return launch_ok("param_norm_fin");
}

int ghn3_param_norm_fwd(float* loss, const float* flat, const int64_t* seg_off, float* norms, int n_seg,
                        int64_t flat_numel, const int* first_seg, float* parts, hipStream_t s) {
    if (n_seg <= 0) return GHN3_OK;
    if (!(parts && first_seg)) hipMemsetAsync(norms, 0, sizeof(float) * n_seg, s);
    const int64_t blocks = (flat_numel + NORM_CHUNK - 1) / NORM_CHUNK;
    hipLaunchKernelGGL(param_sq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, flat, seg_off, norms, n_seg,
                       first_seg, parts);
    hipLaunchKernelGGL(param_sqrt_kernel, dim3((n_seg + 3) / 4), dim3(256), 0, s, norms, n_seg, seg_off, first_seg, parts);
    hipLaunchKernelGGL(param_loss_kernel, dim3(1), dim3(256), 0, s, loss, norms, n_seg);
    return launch_ok("param_norm_fwd");
}
__global__ __launch_bounds__(256) void param_norm_bwd_kernel(float* __restrict__ dflat, const float* __restrict__ flat,
                                                             const int64_t* __restrict__ seg_off,
                                                             const float* __restrict__ norms, int n_seg, float g,
                                                             const int* __restrict__ first_seg) {
    const int64_t w0 = (int64_t)blockIdx.x * NORM_CHUNK, w1 = w0 + NORM_CHUNK;
    for (int sgm = first_seg ? first_seg[blockIdx.x] : first_segment_after(seg_off, n_seg, w0);
         sgm < n_seg && seg_off[2 * sgm] < w1; ++sgm) {
        const int64_t c0 = max(seg_off[2 * sgm], w0), c1 = min(seg_off[2 * sgm + 1], w1);
        const float nrm = norms[sgm];
        const float k = nrm > 0.f ? g / nrm : 0.f;
        const int64_t a0 = (c0 + 3) & ~(int64_t)3, a1 = c1 & ~(int64_t)3;
        if (a0 < a1) {
            const float4* f4 = reinterpret_cast<const float4*>(flat + a0);
            float4* d4 = reinterpret_cast<float4*>(dflat + a0);
            const int64_t n4 = (a1 - a0) >> 2;
            for (int64_t q = threadIdx.x; q < n4; q += 256) {
                float4 v = f4[q];
                v.x *= k; v.y *= k; v.z *= k; v.w *= k;
                d4[q] = v;
            }
            for (int64_t e = c0 + threadIdx.x; e < a0; e += 256) dflat[e] = flat[e] * k;
            for (int64_t e = a1 + threadIdx.x; e < c1; e += 256) dflat[e] = flat[e] * k;
        } else {
            for (int64_t e = c0 + threadIdx.x; e < c1; e += 256) dflat[e] = flat[e] * k;
        }
    }
}
int ghn3_param_norm_bwd(float* dflat, const float* flat, const int64_t* seg_off, const float* norms, int n_seg,
                        float g, int64_t flat_numel, const int* first_seg, hipStream_t s) {
    if (n_seg <= 0) return GHN3_OK;
    const int64_t blocks = (flat_numel + NORM_CHUNK - 1) / NORM_CHUNK;
    hipLaunchKernelGGL(param_norm_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dflat, flat, seg_off, norms,
                       n_seg, g, first_seg);
    return launch_ok("param_norm_bwd");
}

// ------------------------------------------------------------------------------------------------
// column sums (bias gradients): out[omap(n)] += sum_m X[m][n]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(float* __restrict__ out, const float* __restrict__ X, int M, int N,
                                                     int ld, int q, int sdim, int stride,
                                                     const int* __restrict__ gather) {
    __shared__ float red[4][64];
    const int l = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + l;
    const int r0 = blockIdx.y * 256;
    float acc = 0.f;
    if (n < N)
        for (int r = r0 + rl; r < min(M, r0 + 256); r += 4) acc += X[(size_t)(gather ? gather[r] : r) * ld + n];
    red[rl][l] = acc;
    __syncthreads();
    if (rl == 0 && n < N) {
        int o = n;
        if (q > 0) o = (n / q) * sdim + (n % q);
        atomicAdd(&out[(size_t)o * stride], red[0][l] + red[1][l] + red[2][l] + red[3][l]);
    }
}
int ghn3_colsum(float* out, const float* X, int M, int N, int ld, int q, int sdim, int stride, int accum,
                const int* gather, hipStream_t s) {
    if (M <= 0 || N <= 0) return GHN3_OK;
    if (!accum) { ghn3_set_error("colsum: only accumulate mode is implemented"); return GHN3_E_ARG; }
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, (M + 255) / 256), dim3(256), 0, s, out, X, M, N, ld, q, sdim,
                       stride, gather);
    return launch_ok("colsum");
}

// out[r][:] (+)= sum_{t in [seg_ptr[r], seg_ptr[r+1])} X[idx[t]][:]      (deterministic gather-sum)
// One workgroup per output row; its four waves take every fourth source row (independent 16-byte loads in flight),
// partial sums are combined in LDS in a fixed order.  VEC: C, ldx, ldo multiples of 4 and 16-byte aligned bases.
template <bool VEC>
__global__ __launch_bounds__(256) void rowseg_sum_kernel(float* __restrict__ out, const float* __restrict__ X,
                                                         const int* __restrict__ seg_ptr, const int* __restrict__ idx,
                                                         int rows, int C, int ldx, int ldo, int accum) {
    extern __shared__ float part[];                    // [4][C]
    const int r = blockIdx.x;
    const int t0 = seg_ptr[r], t1 = seg_ptr[r + 1];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (VEC) {
        const int C4 = C >> 2;
        for (int c4 = lane; c4 < C4; c4 += 64) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int t = t0 + w; t < t1; t += 4) {
                const float4 v = *reinterpret_cast<const float4*>(X + (size_t)idx[t] * ldx + 4 * c4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            *reinterpret_cast<float4*>(part + w * C + 4 * c4) = acc;
        }
    } else {
        for (int c = lane; c < C; c += 64) {
            float acc = 0.f;
            for (int t = t0 + w; t < t1; t += 4) acc += X[(size_t)idx[t] * ldx + c];
            part[w * C + c] = acc;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = accum ? out[(size_t)r * ldo + c] : 0.f;
        acc += (part[c] + part[C + c]) + (part[2 * C + c] + part[3 * C + c]);
        out[(size_t)r * ldo + c] = acc;
    }
}
int ghn3_rowseg_sum(float* out, const float* X, const int* seg_ptr, const int* idx, int rows, int C, int ldx, int ldo,
                    int accum, hipStream_t s) {
    if (rows <= 0) return GHN3_OK;
    const size_t lds = 4 * sizeof(float) * (size_t)C;
    if (lds > 64 * 1024) { ghn3_set_error("rowseg_sum: C=%d too wide", C); return GHN3_E_LIMIT; }
    const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(rowseg_sum_kernel<true>, dim3(rows), dim3(256), lds, s, out, X, seg_ptr, idx, rows, C, ldx,
                           ldo, accum);
    else
        hipLaunchKernelGGL(rowseg_sum_kernel<false>, dim3(rows), dim3(256), lds, s, out, X, seg_ptr, idx, rows, C, ldx,
                           ldo, accum);
    return launch_ok("rowseg_sum");
}

// X[m][n] *= dact(aux[m][n]) in place (deferred epilogue of a split-K dgrad); optional running max |x| of the result
__device__ __forceinline__ float dact_apply(float v, float z, int dact) {
    if (dact == GHN3_DACT_RELU) return z > 0.f ? v : 0.f;
    if (dact == GHN3_DACT_GELU)
        return v * (0.5f * (1.0f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * __expf(-0.5f * z * z));
    return v;
}
// VEC: N == ld (dense rows), N % 4 == 0, 16-byte aligned bases -> one float4 per thread and step
// parts / n_parts / part_stride / rows_parts: the K-split partial sums of the dgrad that are not in X itself (planes
// 1 .. of a plane-wise split): rows m < rows_parts add parts[p * part_stride + m * N + n] for p < n_parts first
// (fixed order: the result is deterministic, unlike atomically accumulated splits).  VEC only.
#define DACT_FLIGHT 8          // planes of one element whose loads are issued together (16 measured no faster for the 15 planes of the bench workload: 118 against 112 us)
template <bool VEC>
__global__ __launch_bounds__(256) void dact_kernel(float* __restrict__ X, const float* __restrict__ aux, int M, int N,
                                                   int ld, int dact, float* __restrict__ amax,
                                                   const float* __restrict__ parts, int n_parts, int64_t part_stride,
                                                   int rows_parts) {
    float mx = 0.f;
    if (VEC) {
        const int64_t total4 = ((int64_t)M * N) >> 2;
        const int64_t lim4 = parts ? ((int64_t)rows_parts * N) >> 2 : 0;
        float4* X4 = reinterpret_cast<float4*>(X);
        const float4* A4 = reinterpret_cast<const float4*>(aux);
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (int64_t)gridDim.x * 256) {
            const float4 z = A4[e];
            float4 v = X4[e];
            if (e < lim4) {
                // the planes of one element, all loads in flight together (the W2 dgrad of the bench workload leaves 15
                // planes: every further batch is another dependent round trip), summed in plane order
                for (int p0 = 0; p0 < n_parts; p0 += DACT_FLIGHT) {
                    float4 q[DACT_FLIGHT];
#pragma unroll
                    for (int p = 0; p < DACT_FLIGHT; ++p)
                        q[p] = p0 + p < n_parts ? *reinterpret_cast<const float4*>(parts + (int64_t)(p0 + p) * part_stride + 4 * e)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int p = 0; p < DACT_FLIGHT; ++p)
                        if (p0 + p < n_parts) { v.x += q[p].x; v.y += q[p].y; v.z += q[p].z; v.w += q[p].w; }
                }
            }
            v.x = dact_apply(v.x, z.x, dact); v.y = dact_apply(v.y, z.y, dact);
            v.z = dact_apply(v.z, z.z, dact); v.w = dact_apply(v.w, z.w, dact);
            X4[e] = v;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        const int64_t total = (int64_t)M * N;
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
            const int64_t m = e / N;
            const int64_t o = m * ld + (e - m * N);
            const float v = dact_apply(X[o], aux[o], dact);
            X[o] = v;
            mx = fmaxf(mx, fabsf(v));
        }
    }
    if (amax) ghn3_atomic_amax(amax, mx);
}
int ghn3_dact(float* X, const float* aux, int M, int N, int ld, int dact, float* amax, const float* parts, int n_parts,
              int64_t part_stride, int rows_parts, hipStream_t s) {
    if (M <= 0 || N <= 0) return GHN3_OK;
    const bool vec = N == ld && (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(aux) & 15) == 0);
    if (parts && n_parts > 0 && (!vec || (reinterpret_cast<uintptr_t>(parts) & 15) || (part_stride & 3))) {
        ghn3_set_error("dact: partial planes need dense 16-byte aligned rows (N == ld, N %% 4 == 0)");
        return GHN3_E_ARG;
    }
    if (!parts || n_parts <= 0) { parts = nullptr; n_parts = 0; }
    int64_t blocks = ((int64_t)M * N / (vec ? 4 : 1) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (vec)
        hipLaunchKernelGGL(dact_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, X, aux, M, N, ld, dact, amax,
                           parts, n_parts, part_stride, rows_parts);
    else
        hipLaunchKernelGGL(dact_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, X, aux, M, N, ld, dact, amax,
                           parts, n_parts, part_stride, rows_parts);
    return launch_ok("dact");
}

// GHN3_OP_RELU_FIX (see include/ghn3_hip.h): in-place ReLU of the classifier tiles; elements within tau of zero are
// recomputed in fp32 from the master weights first, so that the ReLU mask of the backward does not hang on 16-bit rounding.
__global__ __launch_bounds__(256) void relu_fix_kernel(float* __restrict__ X, const float* __restrict__ U,
                                                       const float* __restrict__ W, const float* __restrict__ bias,
                                                       int cols, int ld, int K, int q, int sdim, float tau_rel) {
    __shared__ float part[4];
    __shared__ int cnt;
    __shared__ int list[1024];
    const int r = blockIdx.y, c0 = blockIdx.x * 1024, tid = threadIdx.x;
    float* xr = X + (size_t)r * ld;
    const int c = c0 + tid * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const bool vec = c + 3 < cols;
    if (vec) { const float4 t = *reinterpret_cast<const float4*>(xr + c); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else for (int e = 0; e < 4; ++e) if (c + e < cols) v[e] = xr[c + e];
    float ss = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (tid == 0) cnt = 0;
    if ((tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    const int n_valid = min(1024, cols - c0);
    const float tau = tau_rel > 0.f ? tau_rel * sqrtf((part[0] + part[1] + part[2] + part[3]) / (float)n_valid) : -1.f;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (c + e < cols && fabsf(v[e]) < tau) list[atomicAdd(&cnt, 1)] = tid * 4 + e;
    if (vec) *reinterpret_cast<float4*>(xr + c) = make_float4(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
    else for (int e = 0; e < 4; ++e) if (c + e < cols) xr[c + e] = fmaxf(v[e], 0.f);
    __syncthreads();
    // one wave per candidate: exact fp32 dot product of the weight row with this decoder row's hidden vector
    const int n = cnt, lane = tid & 63;
    const float* ur = U + (size_t)r * K;
    for (int k = tid >> 6; k < n; k += 4) {
        const int cc = c0 + list[k];
        const int wrow = q > 0 ? (cc / q) * sdim + cc % q : cc;
        const float* wr = W + (size_t)wrow * K;
        float acc = 0.f;
        for (int j = lane * 4; j < K; j += 256) {                     // (K % 4 == 0)
            const float4 a = *reinterpret_cast<const float4*>(wr + j);
            const float4 b = *reinterpret_cast<const float4*>(ur + j);
            acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) xr[cc] = fmaxf(acc + (bias ? bias[wrow] : 0.f), 0.f);
    }
}
int ghn3_relu_fix(float* X, const float* U, const float* W, const float* bias, int rows, int cols, int ld, int K, int q,
                  int sdim, float tau_rel, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return GHN3_OK;
    if ((ld & 3) || (K & 3) || (reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(U) & 15) ||
        (reinterpret_cast<uintptr_t>(W) & 15)) {
        ghn3_set_error("relu_fix: ld %% 4, K %% 4 and 16-byte aligned X / U / W required");
        return GHN3_E_ARG;
    }
    hipLaunchKernelGGL(relu_fix_kernel, dim3((cols + 1023) / 1024, rows), dim3(256), 0, s, X, U, W, bias, cols, ld, K, q,
                       sdim, tau_rel);
    return launch_ok("relu_fix");
}

__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) dst[e] += src[e];
}
int ghn3_add(float* dst, const float* src, int64_t n, hipStream_t s) {
    if (n <= 0) return GHN3_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, src, n);
    return launch_ok("add");
}

// ---------------------------------------------------------------------------------------------------------------
// GHN3_OP_WIRE_PACK / GHN3_OP_RANK_REDUCE: the local passes of the mesh gradient exchange (ddp_utils.mesh_all_reduce_avg).
// HBM streaming kernels: 16-byte accesses, grid-stride, bf16 round-to-nearest-even.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned short wire_bf16(float x) {
    unsigned u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN (the guard needs it)
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float wire_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

__global__ __launch_bounds__(256) void wire_pack_kernel(unsigned short* __restrict__ dst, const float* __restrict__ src,
                                                        int64_t n, int64_t n_pad) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 8;
    for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; e < n_pad; e += stride) {
        if (e + 8 <= n && ((reinterpret_cast<uintptr_t>(src + e) | reinterpret_cast<uintptr_t>(dst + e)) & 15) == 0) {
            const float4 a = *reinterpret_cast<const float4*>(src + e), b = *reinterpret_cast<const float4*>(src + e + 4);
            uint4 o;
            o.x = wire_bf16(a.x) | ((unsigned)wire_bf16(a.y) << 16); o.y = wire_bf16(a.z) | ((unsigned)wire_bf16(a.w) << 16);
            o.z = wire_bf16(b.x) | ((unsigned)wire_bf16(b.y) << 16); o.w = wire_bf16(b.z) | ((unsigned)wire_bf16(b.w) << 16);
            *reinterpret_cast<uint4*>(dst + e) = o;
        } else {
            for (int k = 0; k < 8 && e + k < n_pad; ++k) dst[e + k] = e + k < n ? wire_bf16(src[e + k]) : (unsigned short)0;
        }
    }
}
__global__ __launch_bounds__(256) void wire_unpack_kernel(float* __restrict__ dst, const unsigned short* __restrict__ src,
                                                          int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 8;
    for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; e < n; e += stride) {
        if (e + 8 <= n && ((reinterpret_cast<uintptr_t>(src + e) | reinterpret_cast<uintptr_t>(dst + e)) & 15) == 0) {
            const uint4 v = *reinterpret_cast<const uint4*>(src + e);
            float4 a, b;
            a.x = wire_f32(v.x & 0xffff); a.y = wire_f32(v.x >> 16); a.z = wire_f32(v.y & 0xffff); a.w = wire_f32(v.y >> 16);
            b.x = wire_f32(v.z & 0xffff); b.y = wire_f32(v.z >> 16); b.z = wire_f32(v.w & 0xffff); b.w = wire_f32(v.w >> 16);
            *reinterpret_cast<float4*>(dst + e) = a;
            *reinterpret_cast<float4*>(dst + e + 4) = b;
        } else {
            for (int k = 0; k < 8 && e + k < n; ++k) dst[e + k] = wire_f32(src[e + k]);
        }
    }
}
int ghn3_wire_pack(void* dst, const void* src, int64_t n, int64_t n_pad, int reverse, hipStream_t s) {
    const int64_t total = reverse ? n : n_pad;
    if (total <= 0) return GHN3_OK;
    if (!reverse && n_pad < n) { ghn3_set_error("wire_pack: padded length below the valid length"); return GHN3_E_ARG; }
    int64_t blocks = (total / 8 + 255) / 256 + 1;
    if (blocks > 8192) blocks = 8192;
    if (reverse) hipLaunchKernelGGL(wire_unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (float*)dst, (const unsigned short*)src, n);
    else hipLaunchKernelGGL(wire_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (unsigned short*)dst, (const float*)src, n, n_pad);
    return launch_ok("wire_pack");
}

template <bool IN16, bool OUT16>
__global__ __launch_bounds__(256) void rank_reduce_kernel(void* __restrict__ out_, const void* __restrict__ in_, int64_t per, int W,
                                                          float scale) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; e < per; e += stride) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const int cnt = (int)(per - e < 4 ? per - e : 4);
        for (int w = 0; w < W; ++w) {                  // rank order: the same bits on every rank
            if (IN16) {
                const unsigned short* p = reinterpret_cast<const unsigned short*>(in_) + (int64_t)w * per + e;
                if (cnt == 4 && (reinterpret_cast<uintptr_t>(p) & 7) == 0) {
                    const uint2 v = *reinterpret_cast<const uint2*>(p);
                    acc[0] += wire_f32(v.x & 0xffff); acc[1] += wire_f32(v.x >> 16);
                    acc[2] += wire_f32(v.y & 0xffff); acc[3] += wire_f32(v.y >> 16);
                } else for (int k = 0; k < cnt; ++k) acc[k] += wire_f32(p[k]);
            } else {
                const float* p = reinterpret_cast<const float*>(in_) + (int64_t)w * per + e;
                if (cnt == 4 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
                    const float4 v = *reinterpret_cast<const float4*>(p);
                    acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
                } else for (int k = 0; k < cnt; ++k) acc[k] += p[k];
            }
        }
        for (int k = 0; k < cnt; ++k) {
            if (OUT16) reinterpret_cast<unsigned short*>(out_)[e + k] = wire_bf16(acc[k] * scale);
            else reinterpret_cast<float*>(out_)[e + k] = acc[k] * scale;
        }
    }
}
int ghn3_rank_reduce(void* out, const void* in, int64_t per, int W, int in16, int out16, float scale, hipStream_t s) {
    if (per <= 0 || W <= 0) return GHN3_OK;
    int64_t blocks = (per / 4 + 255) / 256 + 1;
    if (blocks > 8192) blocks = 8192;
    const dim3 g((unsigned)blocks), b(256);
    if (in16 && out16) hipLaunchKernelGGL((rank_reduce_kernel<true, true>), g, b, 0, s, out, in, per, W, scale);
    else if (in16) hipLaunchKernelGGL((rank_reduce_kernel<true, false>), g, b, 0, s, out, in, per, W, scale);
    else if (out16) hipLaunchKernelGGL((rank_reduce_kernel<false, true>), g, b, 0, s, out, in, per, W, scale);
    else hipLaunchKernelGGL((rank_reduce_kernel<false, false>), g, b, 0, s, out, in, per, W, scale);
    return launch_ok("rank_reduce");
}

// GHN3_OP_TRANSPOSE32: batched fp32 transpose through LDS (64 x 64 tiles, 16-byte accesses on both sides)
__global__ __launch_bounds__(256) void transpose32_kernel(float* __restrict__ dst, const float* __restrict__ src, int rows, int cols,
                                                          int ld_src, int ld_dst, int64_t sb, int64_t db) {
    __shared__ float t[64][65];
    const float* S = src + (int64_t)blockIdx.z * sb;
    float* D = dst + (int64_t)blockIdx.z * db;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // 16 float4 per 64-float row, 16 rows per pass
    for (int i = ty; i < 64; i += 16) {
        const int r = r0 + i, c = c0 + tx * 4;
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < rows) {
            if (c + 3 < cols && ((ld_src & 3) == 0) && ((reinterpret_cast<uintptr_t>(S) & 15) == 0)) v = *reinterpret_cast<const float4*>(S + (int64_t)r * ld_src + c);
            else { float e[4] = {0.f, 0.f, 0.f, 0.f}; for (int k = 0; k < 4; ++k) if (c + k < cols) e[k] = S[(int64_t)r * ld_src + c + k]; v = {e[0], e[1], e[2], e[3]}; }
        }
        t[i][tx * 4] = v.x; t[i][tx * 4 + 1] = v.y; t[i][tx * 4 + 2] = v.z; t[i][tx * 4 + 3] = v.w;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 16) {
        const int c = c0 + i, r = r0 + tx * 4;                   // output row = source column
        if (c >= cols) continue;
        float e[4] = {t[tx * 4][i], t[tx * 4 + 1][i], t[tx * 4 + 2][i], t[tx * 4 + 3][i]};
        if (r + 3 < rows && ((ld_dst & 3) == 0) && ((reinterpret_cast<uintptr_t>(D) & 15) == 0)) *reinterpret_cast<float4*>(D + (int64_t)c * ld_dst + r) = float4{e[0], e[1], e[2], e[3]};
        else for (int k = 0; k < 4; ++k) if (r + k < rows) D[(int64_t)c * ld_dst + r + k] = e[k];
    }
}
int ghn3_transpose32(float* dst, const float* src, int rows, int cols, int ld_src, int ld_dst, int batch, int64_t sb, int64_t db,
                     hipStream_t s) {
    if (rows <= 0 || cols <= 0 || batch <= 0) return GHN3_OK;
    if (batch > 65535 || (rows + 63) / 64 > 65535) { ghn3_set_error("transpose32: too many batches / row tiles"); return GHN3_E_LIMIT; }
    hipLaunchKernelGGL(transpose32_kernel, dim3((cols + 63) / 64, (rows + 63) / 64, batch), dim3(256), 0, s, dst, src, rows, cols,
                       ld_src, ld_dst, sb, db);
    return launch_ok("transpose32");
}

// ---------------------------------------------------------------------------------------------------------------
// GHN3_OP_CAST16: fp32 -> f16 / bf16 operand copies for the 16-bit-operand GEMM (gemm.hip gemm_h16d_kernel).
// One workgroup = one 64 x 64 source tile: read once (float4, coalesced), converted, written straight (128-byte row
// segments) and / or transposed through LDS (128-byte column segments); source elements outside rows x cols read as
// zero, which produces the zero K padding the GEMM relies on.  The optional column sum (fp32) is the bias gradient.
// ---------------------------------------------------------------------------------------------------------------
// (saturating: a 16-bit operand copy never holds an infinity the fp32 source did not -- |x| > 65504, e.g. a loaded weight >= 1024
//  behind the 2^6 shift of GHN3_CAST_SPLIT_F16, becomes +-65504 and the lo piece x - hi stays finite; NaN stays NaN)
__device__ __forceinline__ unsigned short cast_f16(float x) {
    const float c = x != x ? x : __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    return __builtin_bit_cast(unsigned short, (_Float16)c);
}
__device__ __forceinline__ unsigned short cast_bf16(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7fffu + ((u >> 16) & 1u);              // round to nearest even (finite inputs)
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_back(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ float f16_back(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
typedef unsigned short us4 __attribute__((ext_vector_type(4)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));

struct AdamWArgs { float lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2_sqrt, max_norm, inv_scale; int nt; };
// The optimizer pass streams 28-34 bytes per parameter exactly once.  Through the default cache policy that stream flushes the L2 of
// every XCD within microseconds -- and the next forward's Graphormer chain, which runs beside the decoder ranges of the pass
// (FusedAdamW.step(overlap=True)), then finds its producers' outputs in HBM instead of L2 (tools/contention_probe: 68 -> 350 ns per
// dependent access; none of it with non-temporal loads / stores).  GHN3_ADAMW_NT=0: default policy (A/B).
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float* p, bool nt) {
    if (nt) { const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p)); return make_float4(v[0], v[1], v[2], v[3]); }
    return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ void st4(float* p, const float4& x, bool nt) {
    if (nt) { const f32x4_nt v = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(v, reinterpret_cast<f32x4_nt*>(p)); }
    else *reinterpret_cast<float4*>(p) = x;
}
static int adamw_nt_env() {
    static const int v = getenv("GHN3_ADAMW_NT") ? atoi(getenv("GHN3_ADAMW_NT")) != 0 : 1;
    return v;
}
// gradient / moment buffers congruent to the source of a cast (same float offsets): GHN3_OP_ADAMW_CAST16
struct AdamWSrc { const float* g; float* m; float* v; const float* sumsq; AdamWArgs a; };

__device__ __forceinline__ float nofuse(float x) { asm volatile("" : "+v"(x)); return x; }   // the value exists in a register: no FMA across it
// per-launch scalars of the update, explicitly rounded like adamw_element: the two kernels that share them must not differ by a
// contraction the compiler picks in one and not in the other (1 - lr wd as one FMA: parameters one ulp apart)
__device__ __forceinline__ void adamw_scalars(const AdamWArgs& a, const float* sumsq, float& clip, float& step, float& decay) {
    // (hipcc contracts a * b + c wherever it likes -- -ffp-contract=fast, the __f*_rn functions are plain operators in this HIP
    // version and a contract(off) pragma does not survive inlining -- so every product that feeds a sum passes through nofuse())
    clip = a.inv_scale;                                // (1 / loss scale: the gradients arrive multiplied by it)
    if (sumsq && a.max_norm > 0.f)
        clip = clip * fminf(1.f, a.max_norm / (nofuse(sqrtf(*sumsq) * a.inv_scale) + 1e-6f));
    step = a.lr / a.bias_corr1;
    decay = 1.f - nofuse(a.lr * a.weight_decay);
}
// one element of torch.optim.AdamW (decoupled weight decay); `clip` = clip_grad_norm_'s coefficient / loss scale
__device__ __forceinline__ void adamw_element(float& p, float g, float& m, float& v, const AdamWArgs& a, float clip,
                                              float step, float decay) {
    // (separately rounded operations: the compiler must not contract them differently in the kernels that share this
    // function -- GHN3_OP_ADAMW and GHN3_OP_ADAMW_CAST16 produce the same bits.  nofuse() is what guarantees it: `p decay -
    // update` was one FMA in one of the two kernels only)
    const float gi = g * clip;
    const float mi = __builtin_fmaf(a.beta1, m, nofuse((1.f - a.beta1) * gi));
    const float vi = __builtin_fmaf(a.beta2, v, nofuse(nofuse((1.f - a.beta2) * gi) * gi));
    m = mi; v = vi;
    const float den = nofuse(sqrtf(vi) / a.bias_corr2_sqrt) + a.eps;
    p = nofuse(p * decay) - nofuse(nofuse(step * mi) / den);
}

// ADAMW: the source elements are parameters that first take their optimizer update (gradient / moments at the same float
// offsets in aw.g / aw.m / aw.v; written back in place) and are then cast -- the 16-bit copies of a weight follow its
// update without a second pass over it (host contract: fp32 source, cols % 4 == 0, no column map, every source element
// in exactly one work tile)
template <bool ADAMW>
__device__ __forceinline__ void cast16_body(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                            const ghn3_cast_desc* __restrict__ descs, int n_desc,
                                            float* __restrict__ dbias, int total_items,
                                            const float* __restrict__ amax, const AdamWSrc& aw) {
    __shared__ unsigned short tr[64][66];          // transposed-copy staging (already converted)
    __shared__ float csum[16][64];
    float aw_clip = 1.f, aw_step = 0.f, aw_decay = 1.f;
    if (ADAMW) {
        if (aw.sumsq && !isfinite(*aw.sumsq)) return;   // (NaN guard as adamw_kernel: parameters AND copies stay as they are)
        adamw_scalars(aw.a, aw.sumsq, aw_clip, aw_step, aw_decay);
    }
    // grid-stride over the 64 x 64 work tiles: a launch may cap its grid (side-stream copies that should leave
    // HBM bandwidth to the latency-bound chain they run under)
    for (int item = blockIdx.x; item < total_items; item += gridDim.x) {
    int lo = 0, hi = n_desc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block_start <= item) lo = mid; else hi = mid - 1;
    }
    const ghn3_cast_desc D = descs[lo];
    const int t = item - D.block_start;
    const int tiles_c = (D.cols + 63) >> 6;
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    const int tid = threadIdx.x;
    const int c4 = (tid & 15) * 4, rr = tid >> 4;
    const float* S = src + D.src_off;
    const bool st = D.flags & GHN3_CAST_STRAIGHT, trn = D.flags & GHN3_CAST_TRANSPOSED;
    const bool split = D.flags & GHN3_CAST_SPLIT;   // bf16 hi + lo copies (GHN3_GEMM_X3 operands)
    // fragment-major copies (gemm_x3d.hip): element (n, k) of a [rows n][cols k] matrix lives at
    // ((n / 16) * (K / 32) + k / 32) * 512 + ((k % 32) / 8 * 16 + n % 16) * 8 + k % 8   (K = cols, a multiple of 32)
    const bool frag = D.flags & GHN3_CAST_FRAG;
    // GHN3_CAST_SPLIT_F16: the straight pieces are f16 pieces of x * 2^GHN3_X3F16_WSHIFT (B of GHN3_GEMM_X3F16 problems)
    const bool st_f16s = split && (D.flags & GHN3_CAST_SPLIT_F16);
    const float s16 = st_f16s ? (float)(1 << GHN3_X3F16_WSHIFT) : 1.f;
    const bool st_bf = ((D.flags & GHN3_CAST_STRAIGHT_BF16) || split) && !st_f16s, tr_bf = (D.flags & GHN3_CAST_TRANSPOSED_BF16) || split;
    const float sc = ((D.flags & GHN3_CAST_SCALED) && amax) ? ghn3_pow2_scale(*amax) : 1.f;
    const int rows_w = (D.flags & GHN3_CAST_TIGHT) ? ((D.rows + 7) & ~7) : 0x7fffffff;   // transposed rows written

    if (!ADAMW && st && !trn && !frag && !(D.flags & (GHN3_CAST_COLSUM | GHN3_CAST_SRC16)) && D.src_q == 0 && !(D.ld_dst & 7) && !(D.dst_off & 7) && !(D.lo_off & 7)) {
        // straight copy only (the dgrad operand of the decoder gradients, forward activations): 8 consecutive floats per
        // lane -> one 16-byte store (the general path below writes 8 bytes per lane)
        unsigned short* Dd = dst + D.dst_off;
        // (no transposition here, so the work tile need not be square: block t of the descriptor covers the 4096 consecutive
        // elements [4096 t, 4096 t + 4096) of the row-major (64-padded rows) x (64-padded cols) grid -- 8 KB contiguous
        // source runs per pass instead of 64 runs of 256 bytes from 64 different rows / DRAM pages)
        const int64_t wp = (int64_t)tiles_c * 64;               // padded row length
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t e0 = (int64_t)t * 4096 + 2048 * i + tid * 8;
            const int r = (int)(e0 / wp), c = (int)(e0 - (int64_t)r * wp);
            if (r >= D.rows) continue;
            const float* p = S + (int64_t)r * D.ld_src + c;
            float x[8];
            if (c + 7 < D.cols) {
                const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
                x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = c + e < D.cols ? p[e] : 0.f;
            }
            us8 h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xs = x[e] * sc * s16;
                h[e] = st_bf ? cast_bf16(xs) : cast_f16(xs);
                if (split) l[e] = st_f16s ? cast_f16(xs - f16_back(h[e])) : cast_bf16(xs - bf16_back(h[e]));
            }
            *reinterpret_cast<us8*>(Dd + (int64_t)r * D.ld_dst + c) = h;
            if (split) *reinterpret_cast<us8*>(Dd + D.lo_off + (int64_t)r * D.ld_dst + c) = l;
        }
        continue;                                      // (uniform over the workgroup: no barrier is skipped by a subset)
    }
    // GHN3_CAST_SRC16: the source is a 16-bit matrix (offsets from r1, type of the copies written) that already carries the
    // GHN3_CAST_SCALED scale: the values are re-laid out as they are, the column sums take the scale back out
    const bool src16 = D.flags & GHN3_CAST_SRC16;
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + rr + 16 * i, c = c0 + c4;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < D.rows) {
            // (a float4 never straddles a run of the source column map: src_q % 4 == 0, c % 4 == 0)
            const int sc_ = D.src_q > 0 ? (c / D.src_q) * D.src_s + c % D.src_q : c;
            if (src16) {
                const unsigned short* p = dst + D.src_off + (int64_t)r * D.ld_src + sc_;
                unsigned short h0 = 0, h1 = 0, h2 = 0, h3 = 0;
                if (c + 3 < D.cols) {
                    const uint2 w = *reinterpret_cast<const uint2*>(p);      // (8-byte aligned: every term is a multiple of 4)
                    h0 = (unsigned short)(w.x & 0xffffu); h1 = (unsigned short)(w.x >> 16);
                    h2 = (unsigned short)(w.y & 0xffffu); h3 = (unsigned short)(w.y >> 16);
                } else {
                    if (c < D.cols) h0 = p[0];
                    if (c + 1 < D.cols) h1 = p[1];
                    if (c + 2 < D.cols) h2 = p[2];
                }
                if (tr_bf) x = make_float4(bf16_back(h0), bf16_back(h1), bf16_back(h2), bf16_back(h3));
                else x = make_float4(f16_back(h0), f16_back(h1), f16_back(h2), f16_back(h3));
            } else if (ADAMW) {
                if (c + 3 < D.cols) {
                    const int64_t o = D.src_off + (int64_t)r * D.ld_src + c;
                    float* pp = const_cast<float*>(src) + o;
                    const bool nt = aw.a.nt != 0;
                    x = ld4(pp, nt);
                    const float4 gv = ld4(aw.g + o, nt);
                    float4 mv = ld4(aw.m + o, nt), vv = ld4(aw.v + o, nt);
                    adamw_element(x.x, gv.x, mv.x, vv.x, aw.a, aw_clip, aw_step, aw_decay);
                    adamw_element(x.y, gv.y, mv.y, vv.y, aw.a, aw_clip, aw_step, aw_decay);
                    adamw_element(x.z, gv.z, mv.z, vv.z, aw.a, aw_clip, aw_step, aw_decay);
                    adamw_element(x.w, gv.w, mv.w, vv.w, aw.a, aw_clip, aw_step, aw_decay);
                    st4(aw.m + o, mv, nt);
                    st4(aw.v + o, vv, nt);
                    st4(pp, x, nt);
                }
            } else {
                const float* p = S + (int64_t)r * D.ld_src + sc_;
                if (c + 3 < D.cols) x = *reinterpret_cast<const float4*>(p);
                else {
                    if (c < D.cols) x.x = p[0];
                    if (c + 1 < D.cols) x.y = p[1];
                    if (c + 2 < D.cols) x.z = p[2];
                }
            }
        }
        v[i] = x;
    }
    if (D.flags & GHN3_CAST_COLSUM) {
        const float un = src16 ? 1.f / sc : 1.f;        // (a power of two: exact)
        csum[rr][c4] = (v[0].x + v[1].x + v[2].x + v[3].x) * un;
        csum[rr][c4 + 1] = (v[0].y + v[1].y + v[2].y + v[3].y) * un;
        csum[rr][c4 + 2] = (v[0].z + v[1].z + v[2].z + v[3].z) * un;
        csum[rr][c4 + 3] = (v[0].w + v[1].w + v[2].w + v[3].w) * un;
    }
    if (sc != 1.f && !src16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i].x *= sc; v[i].y *= sc; v[i].z *= sc; v[i].w *= sc; }
    }
    if (st) {
        unsigned short* Dd = dst + D.dst_off;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + rr + 16 * i;
            if (r >= D.rows) continue;
            us4 h;
            const float4 vs = make_float4(v[i].x * s16, v[i].y * s16, v[i].z * s16, v[i].w * s16);
            if (st_bf) { h[0] = cast_bf16(v[i].x); h[1] = cast_bf16(v[i].y); h[2] = cast_bf16(v[i].z); h[3] = cast_bf16(v[i].w); }
            else { h[0] = cast_f16(vs.x); h[1] = cast_f16(vs.y); h[2] = cast_f16(vs.z); h[3] = cast_f16(vs.w); }
            const int c = c0 + c4;
            // (frag: n = r, k = c; the four consecutive k of this lane share one 8-element run)
            const int64_t o = frag ? ((int64_t)(r >> 4) * (D.cols >> 5) + (c >> 5)) * 512 + ((((c & 31) >> 3) * 16 + (r & 15)) << 3) + (c & 7)
                                   : (int64_t)r * D.ld_dst + c;
            if (frag && c >= D.cols) continue;
            *reinterpret_cast<us4*>(Dd + o) = h;
            if (split) {
                us4 l;
                if (st_f16s) {
                    l[0] = cast_f16(vs.x - f16_back(h[0])); l[1] = cast_f16(vs.y - f16_back(h[1]));
                    l[2] = cast_f16(vs.z - f16_back(h[2])); l[3] = cast_f16(vs.w - f16_back(h[3]));
                } else {
                    l[0] = cast_bf16(v[i].x - bf16_back(h[0])); l[1] = cast_bf16(v[i].y - bf16_back(h[1]));
                    l[2] = cast_bf16(v[i].z - bf16_back(h[2])); l[3] = cast_bf16(v[i].w - bf16_back(h[3]));
                }
                *reinterpret_cast<us4*>(Dd + D.lo_off + o) = l;
            }
        }
    }
    if (trn) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rr + 16 * i;
            if (tr_bf) {
                tr[r][c4] = cast_bf16(v[i].x); tr[r][c4 + 1] = cast_bf16(v[i].y);
                tr[r][c4 + 2] = cast_bf16(v[i].z); tr[r][c4 + 3] = cast_bf16(v[i].w);
            } else {
                tr[r][c4] = cast_f16(v[i].x); tr[r][c4 + 1] = cast_f16(v[i].y);
                tr[r][c4 + 2] = cast_f16(v[i].z); tr[r][c4 + 3] = cast_f16(v[i].w);
            }
        }
    }
    __syncthreads();
    if (trn) {
        unsigned short* Dt = dst + D.dstT_off;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = tid + 256 * i;
            const int col = p >> 3, rp = (p & 7) * 8;
            if (c0 + col >= D.cols || r0 + rp >= rows_w) continue;
            us8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = tr[rp + e][col];
            // (frag: the transposed matrix has n = source column, k = source row; 8 consecutive k per store)
            const int fn = c0 + col, fk = r0 + rp;
            const int64_t o = frag ? ((int64_t)(fn >> 4) * (D.rows >> 5) + (fk >> 5)) * 512 + ((((fk & 31) >> 3) * 16 + (fn & 15)) << 3)
                                   : (int64_t)fn * D.ld_dstT + fk;
            if (frag && fk >= D.rows) continue;
            *reinterpret_cast<us8*>(Dt + o) = h;
        }
    }
    if (trn && split) {
        // second round through the staging tile: the lo halves (x - hi, both known to the thread that loaded x)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rr + 16 * i;
            tr[r][c4] = cast_bf16(v[i].x - bf16_back(cast_bf16(v[i].x)));
            tr[r][c4 + 1] = cast_bf16(v[i].y - bf16_back(cast_bf16(v[i].y)));
            tr[r][c4 + 2] = cast_bf16(v[i].z - bf16_back(cast_bf16(v[i].z)));
            tr[r][c4 + 3] = cast_bf16(v[i].w - bf16_back(cast_bf16(v[i].w)));
        }
        __syncthreads();
        unsigned short* Dt = dst + D.dstT_off + D.lo_off;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = tid + 256 * i;
            const int col = p >> 3, rp = (p & 7) * 8;
            if (c0 + col >= D.cols || r0 + rp >= rows_w) continue;
            us8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = tr[rp + e][col];
            // (frag: the transposed matrix has n = source column, k = source row; 8 consecutive k per store)
            const int fn = c0 + col, fk = r0 + rp;
            const int64_t o = frag ? ((int64_t)(fn >> 4) * (D.rows >> 5) + (fk >> 5)) * 512 + ((((fk & 31) >> 3) * 16 + (fn & 15)) << 3)
                                   : (int64_t)fn * D.ld_dstT + fk;
            if (frag && fk >= D.rows) continue;
            *reinterpret_cast<us8*>(Dt + o) = h;
        }
    }
    if ((D.flags & GHN3_CAST_COLSUM) && tid < 64 && c0 + tid < D.cols) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += csum[w][tid];
        if (D.flags & GHN3_CAST_COLSUM_PARTS) {
            // deterministic: this row tile's partial sums go to their own slot; GHN3_OP_ROWSET_COLSUM adds the slots of
            // all row tiles (and row sets) of a bias entry in a fixed order
            dbias[D.part_off + (int64_t)(r0 >> 6) * D.cols + c0 + tid] = s;
        } else {
            int c = c0 + tid;
            if (D.bias_q > 0) c = (c / D.bias_q) * D.bias_s + c % D.bias_q;
            atomicAdd(dbias + c + D.bias_off, s);
        }
    }
    __syncthreads();                               // LDS staging is reused by the next work tile
    }
}

__global__ __launch_bounds__(256) void cast16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                     const ghn3_cast_desc* __restrict__ descs, int n_desc,
                                                     float* __restrict__ dbias, int total_items,
                                                     const float* __restrict__ amax) {
    cast16_body<false>(src, dst, descs, n_desc, dbias, total_items, amax, AdamWSrc{});
}
__global__ __launch_bounds__(256) void adamw_cast16_kernel(float* __restrict__ p, unsigned short* __restrict__ dst,
                                                           const ghn3_cast_desc* __restrict__ descs, int n_desc,
                                                           int total_items, AdamWSrc aw) {
    cast16_body<true>(p, dst, descs, n_desc, nullptr, total_items, nullptr, aw);
}

int ghn3_cast16(const float* src, void* dst, const ghn3_cast_desc* d_desc, int n_desc, int total_blocks, float* dbias,
                const float* amax, int grid_cap, hipStream_t s) {
    if (n_desc <= 0 || total_blocks <= 0) return GHN3_OK;
    const int grid = grid_cap > 0 && grid_cap < total_blocks ? grid_cap : total_blocks;
    hipLaunchKernelGGL(cast16_kernel, dim3(grid), dim3(256), 0, s, src, (unsigned short*)dst, d_desc, n_desc,
                       dbias, total_blocks, amax);
    return launch_ok("cast16");
}

// ---------------------------------------------------------------------------------------------------------------
// Trainer step over the flat buffers (trainer.py:356-381): global gradient norm for
// nn.utils.clip_grad_norm_ and a fused AdamW update -- two passes over the flat parameter / gradient / moment buffers
// instead of ~10 ATen launches per parameter tensor.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(float* __restrict__ out, const float* __restrict__ x, int64_t n,
                                                    float* __restrict__ parts) {
    __shared__ float part[4];
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = x4[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = x[(n4 << 2) + threadIdx.x]; acc += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (part[0] + part[1]) + (part[2] + part[3]);
        if (parts) parts[blockIdx.x] = t; else atomicAdd(out, t);     // (parts: summed in block order below)
    }
}
__global__ void sumsq_finish_kernel(float* __restrict__ out, const float* __restrict__ parts, int n) {
    __shared__ float tot[256];
    float acc = 0.f;
    for (int k = threadIdx.x; k < n; k += 256) acc += parts[k];
    tot[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0.f; for (int k = 0; k < 256; ++k) t += tot[k]; out[0] += t; }
}
// plain sums of a slot table, one partial per workgroup (fixed order): parts[b] = sum of the floats block b strides over
__global__ __launch_bounds__(256) void slot_sum_kernel(float* __restrict__ parts, const float* __restrict__ x, int n) {
    __shared__ float part[4];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const float4 v = x4[i];
        a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
    }
    float acc = (a0 + a1) + (a2 + a3);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) acc += x[(n4 << 2) + threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) parts[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
// skip_lo .. skip_hi: a range of x that is left out (its sum of squares arrives as `extra`: partial sums a producer of that
// range left, e.g. the weight-gradient kernel's GHN3_GEMM_SUMSQ slots); needs `parts` (8192 floats then)
int ghn3_sumsq(float* out, const float* x, int64_t n, float* parts, int64_t skip_lo, int64_t skip_hi, const float* extra,
               int n_extra, hipStream_t s) {
    if (n <= 0) return GHN3_OK;
    if (reinterpret_cast<uintptr_t>(x) & 15) { ghn3_set_error("sumsq: buffer must be 16-byte aligned"); return GHN3_E_ARG; }
    const bool skip = skip_hi > skip_lo;
    if (skip || n_extra > 0) {
        if (!parts || skip_lo < 0 || skip_hi > n || (skip_lo & 3) || (skip_hi & 3) || (n_extra > 0 && !extra)) {
            ghn3_set_error("sumsq: a skipped range needs the scratch buffer, 4-float alignment and 0 <= lo <= hi <= n");
            return GHN3_E_ARG;
        }
    }
    auto blocks_of = [](int64_t m) { int64_t b = (m / 4 + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); };
    if (!skip && n_extra <= 0) {
        const int blocks = blocks_of(n);
        hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, s, out, x, n, parts);
        if (parts) hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, s, out, parts, blocks);
        return launch_ok("sumsq");
    }
    // parts: [<= 4096 of the range below the gap | <= 4032 of the range above it | 64 sums of the slot table]
    int b0 = 0, b1 = 0, b2 = 0;
    const int64_t lo = skip ? skip_lo : n, hi = skip ? skip_hi : n;
    if (lo > 0) {
        b0 = blocks_of(lo);
        hipLaunchKernelGGL(sumsq_kernel, dim3(b0), dim3(256), 0, s, out, x, lo, parts);
    }
    if (n > hi) {
        b1 = blocks_of(n - hi);
        if (b1 > 4032) b1 = 4032;
        hipLaunchKernelGGL(sumsq_kernel, dim3(b1), dim3(256), 0, s, out, x + hi, n - hi, parts + b0);
    }
    if (n_extra > 0) {
        if (reinterpret_cast<uintptr_t>(extra) & 15) { ghn3_set_error("sumsq: slot table must be 16-byte aligned"); return GHN3_E_ARG; }
        b2 = 64;
        hipLaunchKernelGGL(slot_sum_kernel, dim3(b2), dim3(256), 0, s, parts + b0 + b1, extra, n_extra);
    }
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, s, out, parts, b0 + b1 + b2);
    return launch_ok("sumsq");
}

// torch.optim.AdamW (decoupled weight decay) with the gradient scaled by clip_grad_norm_'s coefficient
// min(1, max_norm / (||g|| + 1e-6)); `sumsq` holds ||g||^2 (absent: no clipping)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                    const float* __restrict__ sumsq, AdamWArgs a) {
    // Non-finite gradient norm (a NaN / inf anywhere in the -- already rank-averaged -- flat gradient): the update is
    // skipped on every rank alike, like GradScaler's found_inf and the reference trainer's NaN-loss skip
    // (trainer.py:240-257), without a host round trip.
    if (sumsq && !isfinite(*sumsq)) return;
    float clip, step, decay;
    adamw_scalars(a, sumsq, clip, step, decay);
    // 16 bytes per lane (the pass streams 28 bytes per parameter: 4 loads + 3 stores -- HBM-bound); same arithmetic per
    // element as the scalar tail, so results do not depend on the path
    int64_t n4 = 0;
    if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
          reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        n4 = n >> 2;
        const bool nt = a.nt != 0;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            float4 pv = ld4(p + 4 * i, nt), gv = ld4(g + 4 * i, nt), mv = ld4(m + 4 * i, nt), vv = ld4(v + 4 * i, nt);
            float* pe = &pv.x; float* ge = &gv.x; float* me = &mv.x; float* ve = &vv.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) adamw_element(pe[e], ge[e], me[e], ve[e], a, clip, step, decay);
            st4(m + 4 * i, mv, nt); st4(v + 4 * i, vv, nt); st4(p + 4 * i, pv, nt);
        }
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float pi = p[i], mi = m[i], vi = v[i];
        adamw_element(pi, g[i], mi, vi, a, clip, step, decay);
        m[i] = mi; v[i] = vi; p[i] = pi;
    }
}
int ghn3_adamw_cast16(float* p, const float* g, float* m, float* v, void* dst, const ghn3_cast_desc* d_desc, int n_desc,
                      int total_blocks, const float* sumsq, float lr, float beta1, float beta2, float eps, float weight_decay,
                      float bias_corr1, float bias_corr2, float max_norm, float inv_scale, hipStream_t s) {
    if (n_desc <= 0 || total_blocks <= 0) return GHN3_OK;
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v)) & 15) {
        ghn3_set_error("adamw_cast16: buffers must be 16-byte aligned");
        return GHN3_E_ARG;
    }
    AdamWSrc aw{g, m, v, sumsq, AdamWArgs{lr, beta1, beta2, eps, weight_decay, bias_corr1, sqrtf(bias_corr2), max_norm,
                                           inv_scale > 0.f ? inv_scale : 1.f, adamw_nt_env()}};
    hipLaunchKernelGGL(adamw_cast16_kernel, dim3(total_blocks), dim3(256), 0, s, p, (unsigned short*)dst, d_desc, n_desc,
                       total_blocks, aw);
    return launch_ok("adamw_cast16");
}

int ghn3_adamw(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, float lr, float beta1,
               float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, float max_norm,
               float inv_scale, hipStream_t s) {
    if (n <= 0) return GHN3_OK;
    AdamWArgs a{lr, beta1, beta2, eps, weight_decay, bias_corr1, sqrtf(bias_corr2), max_norm,
                inv_scale > 0.f ? inv_scale : 1.f, adamw_nt_env()};
    int64_t blocks = (n / 4 + 255) / 256 + 1;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, n, sumsq, a);
    return launch_ok("adamw");
}
