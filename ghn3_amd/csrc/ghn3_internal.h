// Internal declarations shared by the HIP translation units of libghn3_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ghn3_hip.h"

// Resolved GEMM problem (absolute device pointers); lives in device memory for the launch.
struct GemmProbDev {
    const float* A; const float* B; float* C;
    const float* bias; const float* residual; const float* aux_in; float* aux_out;
    const int* a_gather; const int* b_gather; const int* c_gather;
    int M, N, K, lda, ldb, ldc;
    int a_q, a_s, b_q, b_s, c_q, c_s;
    int bias_q, bias_s, bias_stride;
    int act, dact, flags;
    float alpha;
    int tile_start;      // first tile id of this problem inside its launch (multiple of 8)
    int tiles_m, tiles_n;
    int ksplit, k_chunk; // split-K: K range [z*k_chunk, (z+1)*k_chunk) per replica z < ksplit
    int order;           // 0: m-tiles innermost (B streamed), 1: n-tiles innermost (A streamed)
    int kq, ks;          // 16-bit-operand kernel: k-map of B (kq == 0: identity)
    const int* lim;      // ragged extents per 128 rows (see ghn3_gemm_problem::lim)
    int lim_kind;
    int xcd_cols;        // tile code 25: column groups of the XCD-blocked tile order (1, 2, 4 or 8; see gemm_h16w_kernel)
    // XCD-pinned problems of a 16-bit-operand launch (ghn3_gemm_problem::xcd_pin): they come first in the launch's array,
    // sorted by XCD; tile id t < pin_end belongs to XCD t & 7, local index t >> 3, and tile_start of a pinned problem is
    // its first LOCAL index.  Entry x (x < 8) of the array carries the directory of XCD x; every entry carries the totals.
    int pin;             // 0 or x + 1
    int pin_first, pin_count;   // (entry x) first array index / number of the problems pinned to XCD x
    int pin_end, pin_total;     // ids below pin_end are pinned ids; number of pinned problems
    const float* alpha_amax;   // alpha is divided by ghn3_pow2_scale(*alpha_amax) (operand copies scaled by GHN3_CAST_SCALED)
    int ln_kind;               // row prologue of A (ghn3_gemm_problem::ln_kind), small-problem kernel only
    float ln_eps;
    const float* ln_p[6];
    const void* B2;            // GHN3_GEMM_X3: the bf16 lo copy of B (B itself points at the hi copy)
    const int* mtab;           // tile code 28: row-tile table {m0, mi, extent} x tiles_m (ghn3_gemm_problem::mtiles) or null
};


int ghn3_gemm_init();
// launches one grouped GEMM: all problems share (a_mode, b_mode, tile).  d_probs is device memory.
int ghn3_gemm_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int a_mode, int b_mode,
                     int tile, int ctype, hipStream_t stream);

int ghn3_gemm_h16d_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int tile, int ctype, int grid_cap,
                          hipStream_t stream);
int ghn3_gemm_p8_init();
int ghn3_gemm_p8_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int ctype, int grid_cap,
                        hipStream_t stream);
int ghn3_gemm_p8w_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int ctype, int grid_cap,
                         hipStream_t stream);
int ghn3_gemm_p8d_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int ctype, int grid_cap,
                         hipStream_t stream);
int ghn3_gemm_x3_init();
int ghn3_gemm_x3_tile(int code, int slice, int* bm, int* bn);
int ghn3_gemm_x3_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int code, int slice,
                        hipStream_t stream);
// gemm_x3d.hip: staged split-bf16 GEMMs (tile codes 44 / 45): fragment-major weights, no partial planes, optional
// LayerNorm row prologue
int ghn3_gemm_x3s_init();
int ghn3_gemm_x3s_tile(int code, int K, int ln_kind, int* bm, int* bn);
int ghn3_gemm_x3s_launch(const GemmProbDev* d_probs, const GemmProbDev* h_probs, int n_probs, int total_tiles, int code, int K,
                         int ln_kind, hipStream_t stream);
int ghn3_gemm_wg_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int grid_cap, int tile_edge, hipStream_t stream);
int ghn3_gemm_small_launch(const GemmProbDev* d_probs, int n_probs, int total_tiles, int a_mode, int b_mode, int with_ln,
                           hipStream_t stream);

int ghn3_attn_init();
int ghn3_attn_fwd(float* out, const float* qkv, const float* bias, float* P, const int* n_nodes,
                  int B, int N, int C, int H, hipStream_t s);
int ghn3_attn_bwd(float* dqkv, const float* dO, const float* qkv, const float* P, const float* O, float* amax_out,
                  float* dBias, const int* n_nodes, int B, int N, int C, int H, int general, hipStream_t s);

int ghn3_graph_prologue(const int64_t* A, int* deg_in, int* deg_out, int* dist0, int* pair,
                        int B, int N, int V, hipStream_t s);
int ghn3_embed_nodes(float* x, const int* node_type, const int* shape_idx, const int* n_nodes, const int* node_off,
                     const float* E_type, const float* E_ch, const float* E_sp, const float* E_in,
                     const float* E_out, const float* E_dist, const int* deg_in, const int* deg_out,
                     const int* dist0, int B, int N, int C, hipStream_t s);
int ghn3_embed_bwd(const float* dx, const int* node_type, const int* shape_idx, const int* n_nodes,
                   const int* node_off, float* dE_type, float* dE_ch, float* dE_sp, float* dE_in, float* dE_out,
                   float* dE_dist, const int* deg_in, const int* deg_out, const int* dist0,
                   int B, int N, int C, int n_type, int n_ch, int n_sp, hipStream_t s);
int ghn3_edge_hidden(float* hid, const float* Pfw, const float* Pbw, int V, int C, hipStream_t s);
int ghn3_edge_hidden_bwd(float* dPfw, float* dPbw, float* dhid, const float* hid, int V, int C, hipStream_t s);
int ghn3_bias_gather(float* bias, const float* T, const int* pair, int B, int N, int H, hipStream_t s);
int ghn3_bias_hist(float* dT, const float* dBias, const int* pair, int B, int N, int H, int V, void* scratch,
                   int have_amax, hipStream_t s);
int ghn3_rowset_colsum(float* out, const float* X, const void* sets, int n_sets, int O, int I, hipStream_t s);
int ghn3_layernorm_fwd(float* y, float* x, const float* g, const float* b, float* mean, float* rstd, const float* add,
                       int n_add, int64_t add_stride, int rows, int C, float eps, hipStream_t s);
int ghn3_layernorm_bwd(float* dx, float* dy, const float* x, const float* g, const float* mean,
                       const float* rstd, const float* res, const float* add, int n_add, int64_t add_stride, int rows,
                       int C, hipStream_t s);
int ghn3_ln_param_grad(float* dg, float* db, const float* dy, const float* x, const float* mean,
                       const float* rstd, int rows, int C, int accum, hipStream_t s);
int ghn3_ln_param_grad_batch(float* gbase, const float* abase, const int64_t* table, int n_items, int rows, int C,
                             hipStream_t s);
int ghn3_tile_fwd(float* flat, const float* const* srcs, const ghn3_tile_desc* d_desc, int n_desc,
                  int64_t total, const int64_t* blocks, int lds_bytes, float* sq_parts, float* b_parts, hipStream_t s);
int ghn3_tile_bwd(const float* dflat, const float* const* srcs, float* const* dsrcs,
                  const ghn3_tile_desc* d_desc, int n_desc, int64_t total, const int64_t* blocks, int lds_bytes,
                  float* amax, const float* out, const float* norms, const int* desc_seg, const float* gscale,
                  const int64_t* h16_tab, int h16_bf16, hipStream_t s);
int ghn3_param_norm_fin(float* loss, float* norms, const float* parts, const int* first, int n_seg, const float* b_parts,
                        float* ratio, float* bound, hipStream_t s);
int ghn3_param_norm_fwd(float* loss, const float* flat, const int64_t* seg_off, float* norms, int n_seg,
                        int64_t flat_numel, const int* first_seg, float* parts, hipStream_t s);
int ghn3_param_norm_bwd(float* dflat, const float* flat, const int64_t* seg_off, const float* norms, int n_seg,
                        float g, int64_t flat_numel, const int* first_seg, hipStream_t s);
int ghn3_colsum(float* out, const float* X, int M, int N, int ld, int q, int sdim, int stride, int accum,
                const int* gather, hipStream_t s);
int ghn3_rowseg_sum(float* out, const float* X, const int* seg_ptr, const int* idx, int rows, int C, int ldx,
                    int ldo, int accum, hipStream_t s);
int ghn3_add(float* dst, const float* src, int64_t n, hipStream_t s);
int ghn3_transpose32(float* dst, const float* src, int rows, int cols, int ld_src, int ld_dst, int batch, int64_t sb, int64_t db,
                     hipStream_t s);
int ghn3_wire_pack(void* dst, const void* src, int64_t n, int64_t n_pad, int reverse, hipStream_t s);
int ghn3_rank_reduce(void* out, const void* in, int64_t per, int W, int in16, int out16, float scale, hipStream_t s);
int ghn3_cast16(const float* src, void* dst, const ghn3_cast_desc* d_desc, int n_desc, int total_blocks, float* dbias,
                const float* amax, int grid_cap, hipStream_t s);
int ghn3_sumsq(float* out, const float* x, int64_t n, float* parts, int64_t skip_lo, int64_t skip_hi, const float* extra,
               int n_extra, hipStream_t s);
int ghn3_adamw(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, float lr, float beta1,
               float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, float max_norm,
               float inv_scale, hipStream_t s);
int ghn3_adamw_cast16(float* p, const float* g, float* m, float* v, void* dst, const ghn3_cast_desc* d_desc, int n_desc,
                      int total_blocks, const float* sumsq, float lr, float beta1, float beta2, float eps, float weight_decay,
                      float bias_corr1, float bias_corr2, float max_norm, float inv_scale, hipStream_t s);
int ghn3_dact(float* X, const float* aux, int M, int N, int ld, int dact, float* amax, const float* parts, int n_parts,
              int64_t part_stride, int rows_parts, hipStream_t s);

int ghn3_relu_fix(float* X, const float* U, const float* W, const float* bias, int rows, int cols, int ld, int K, int q,
                  int sdim, float tau_rel, hipStream_t s);

void ghn3_set_error(const char* fmt, ...);

#ifdef __HIPCC__
// Power-of-two scaling of small-magnitude gradients for f16 operand copies: for amax = m * 2^e (1 <= m < 2) the
// scale 2^(11 - e) maps the largest magnitude into [2048, 4096) (16x headroom to the f16 maximum; 2^-14 / 4096 =
// 1.5e-8 of amax still a normal f16 number).  Exact, so the GEMM epilogue undoes it with the exact inverse.
__device__ __forceinline__ int ghn3_amax_exp(float amax) {
    return (int)((__float_as_uint(amax) >> 23) & 0xff) - 127;
}
__device__ __forceinline__ float ghn3_pow2_scale(float amax) {
    const int e = ghn3_amax_exp(amax);
    if (!(amax > 0.f) || e < -100) return 1.f;
    return __uint_as_float((unsigned)(127 + 11 - e) << 23);
}
__device__ __forceinline__ float ghn3_pow2_inv_scale(float amax) {
    const int e = ghn3_amax_exp(amax);
    if (!(amax > 0.f) || e < -100) return 1.f;
    return __uint_as_float((unsigned)(127 - 11 + e) << 23);
}
// running maximum of non-negative floats (bit pattern order == value order); call with all lanes of the wave
__device__ __forceinline__ void ghn3_atomic_amax(float* slot, float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    // (the racy pre-check is safe -- the slot only grows -- and keeps thousands of workgroups from serialising on
    // one address: after the first few, almost no wave exceeds the running maximum)
    if ((threadIdx.x & 63) == 0 && v > *reinterpret_cast<volatile float*>(slot))
        atomicMax(reinterpret_cast<int*>(slot), __float_as_int(v));
}
#endif
