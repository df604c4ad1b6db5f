/*
 * ghn3_hip.h -- C ABI of libghn3_hip.so: the MI355X (gfx950) implementation of the GHN-3
 * parameter-prediction hot path (Graphormer over computational-graph node embeddings + weight-tile
 * decoders, forward and backward).
 *
 * The reference (SamsungSAILMontreal/ghn3) has no FFI layer: its boundary for this path is the Python
 * API `GHN3.forward` (/root/reference/ghn3/nn.py:186-349).  This header therefore defines the ABI that
 * a `GHN3.forward` replacement binds (ghn3_amd/nn.py does so through ctypes; INTEGRATION.md shows the
 * stub a reference maintainer would add).  Every entry point is `extern "C"`, takes plain pointers and
 * sizes, returns 0 on success or a negative GHN3_E_* code, never throws, never allocates device memory
 * and never synchronises the device (except ghn3_event_* helpers, which say so).
 *
 * Execution model: the host compiles one batch of graphs into a flat array of fixed-size `ghn3_op`
 * records (a "program").  Buffers are named by index into a per-call pointer table, so one program is
 * replayed every step with fresh output / gradient buffers.  `ghn3_run` launches the program's HIP
 * kernels on the given stream, in order.  Each op kind replaces a specific sequence of ATen calls in the
 * reference; the citation is on the enum value.
 */
#ifndef GHN3_HIP_H
#define GHN3_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GHN3_ABI_VERSION 19

/* ---- error codes -------------------------------------------------------------------------------- */
#define GHN3_OK            0
#define GHN3_E_ARG        -1   /* malformed op / problem (message via ghn3_last_error) */
#define GHN3_E_LIMIT      -2   /* a documented size limit is exceeded (e.g. N > 1024 nodes)  */
#define GHN3_E_HIP        -3   /* a HIP runtime call failed */
#define GHN3_E_NOCTX      -4

/* ---- buffer references ---------------------------------------------------------------------------- */
/* A reference names `bufs[buf] + off` where off is in BYTES.  buf < 0 means "absent" (NULL). */
typedef struct ghn3_ref {
    int32_t buf;
    int32_t _pad;
    int64_t off;
} ghn3_ref;

/* ---- grouped GEMM --------------------------------------------------------------------------------
 * C(m,n) = epilogue( alpha * sum_k A(m,k) * B(k,n) )           fp32 in HBM (or 16-bit operand copies, see
 *                                                               GHN3_GEMM_OP16), fp32 accumulate.
 *
 * Operand addressing.  An operand is a row-major stored matrix X with leading dimension ld (floats).
 *   ROW mode (k-contiguous):  A(m,k) = X[rmap(m)][k]        B(k,n) = X[rmap(n)][k]
 *   COL mode (k-strided)   :  A(m,k) = X[rmap(k)][m]        B(k,n) = X[rmap(k)][n]
 * rmap(r) = gather ? gather[r] : r ;  then  r -> (r / q) * s + (r % q)  when q > 0.
 * C(m,n) = Y[cmap(m)][n] with the same two-stage map.  ld and the byte offsets must be multiples of 4
 * floats (16 B) for ROW/COL vector loads; M, N, K are arbitrary.
 *
 * Epilogue order: acc*alpha -> +bias[bidx(n)] -> save pre-activation (aux_out) -> activation ->
 *                 *dact(aux_in) -> +residual -> (+= C when GHN3_GEMM_ACCUM) -> store.
 *   bidx(n) = ((n / bias_q) * bias_s + n % bias_q) * bias_stride   (bias_q == 0: n * bias_stride)
 */
enum { GHN3_MODE_ROW = 0, GHN3_MODE_COL = 1 };
enum { GHN3_ACT_NONE = 0, GHN3_ACT_RELU = 1, GHN3_ACT_GELU = 2 };
/* multiply by a derivative evaluated at aux_in(m,n) (same layout/map as C) */
enum { GHN3_DACT_NONE = 0, GHN3_DACT_RELU = 1 /* aux_in > 0 */, GHN3_DACT_GELU = 2 /* gelu'(aux_in) */ };
#define GHN3_GEMM_ACCUM 1u
/* wgrad problems (a_mode == COL): `bias` names the bias GRADIENT; the kernel adds the row sums of A,
 * dbias[cmap(m) * bias_stride] += sum_k A(m,k), and applies no bias to C.  At most one problem of a launch
 * may touch a given dbias element. */
#define GHN3_GEMM_BIASGRAD 2u
/* 16-bit operands ALREADY in HBM (written by GHN3_OP_CAST16): A and B reference f16 or bf16 matrices (the type
 * is the op's compute type, which must be GHN3_CT_F16 / GHN3_CT_BF16), both in ROW mode (k-contiguous).  lda / ldb
 * count 16-bit elements and must be multiples of 8, the byte offsets multiples of 16.  The kernel reads whole
 * 64-wide k-tiles: every operand row must be readable up to round_up(K, 64) and A must hold ZEROS in k >= K.
 * B may carry a k-map: logical k -> physical (k / b_kq) * b_ks + k % b_kq  (b_kq % 8 == 0; 0 = identity) -- the
 * row-subset structure of the decoder W2 weights seen from the reduction side (nn.py:747-749 backward).
 * C, aux_in, aux_out and residual must be 16-byte aligned with ldc % 4 == 0 (row-vector epilogue).
 * Bias, ReLU / dReLU, residual, accumulate, split-K and the C map work as for fp32 operands; GELU and
 * GHN3_GEMM_BIASGRAD are not available (GHN3_OP_CAST16 produces that sum while it writes the transposed copy). */
#define GHN3_GEMM_OP16 4u
/* Split-bf16 operands ("x3"): near-fp32 products on the 16-bit matrix cores for the latency-bound Graphormer linears
 * (graphormer.py:38-44,121,141 and their dgrad) -- v_mfma_f32_16x16x32_bf16 needs 1/16 of the matrix-core cycles of the
 * exact-fp32 v_mfma_f32_32x32x2_f32 per product, three products per term leave ~5x.
 *   A: fp32, ROW mode, optional row gather (a_gather; also c_gather for C), no q / s map; split on the fly into
 *      a = hi + lo (hi = bf16(a), lo = bf16(a - hi)).
 *   B: a weight's persistent bf16 copies written by GHN3_OP_CAST16 with GHN3_CAST_SPLIT: `B` = hi [N][ldb], `B2` = lo
 *      (same layout), both k-contiguous (ROW mode), ldb in 16-bit elements (% 8 == 0), K zero padded to 64.
 *   C = alpha * (hi.hi + hi.lo + lo.hi) (+ epilogue), fp32 accumulate; the dropped lo.lo term is 2^-16 relative.
 * A workgroup keeps a whole K slice (`x3_slice`, a multiple of 64, <= 384 for 32 x 64 tiles) of both operands in LDS
 * and walks the slices of its K range in order.  K splits over workgroups are separate problems (partial planes
 * summed by the consuming LayerNorm op).  Epilogue: bias, ReLU / GELU (+ aux_out), dReLU / dGELU (aux_in), residual;
 * N % 4 == 0, ldc % 4 == 0, 16-byte aligned C / aux / residual / bias.  Tile codes (op.i[2]): 40 = 32 x 64,
 * 41 = 64 x 64, 42 = 32 x 32.
 * Tile codes 44 (32 x 48 tiles, 12 waves) and 45 (16 x 32 tiles, 4 - 16 waves) select the STAGED kernels (gemm_x3d.hip,
 * round 4): B / B2 are FRAGMENT-MAJOR hi / lo copies (GHN3_CAST_FRAG; ldb unused) that every wave loads straight into
 * registers, the workgroup keeps its activation rows over the WHOLE reduction length in LDS (K % 64 == 0, K <= 1536 for
 * 16-row tiles), the waves split K among themselves and add their partial tiles through LDS in a fixed order: no partial
 * planes, x3_slice unused, N % 16 == 0.  These problems may carry the LayerNorm row prologue of A (ln_kind 1 / 2 below,
 * K <= 384, no a_gather): the Graphormer chain then needs no LayerNorm launches (graphormer.py:239-241 and its backward). */
#define GHN3_GEMM_X3 8u
/* Tile codes 29 / 30 only (the persistent weight-gradient kernels): `aux_out` = base of a float slot table; the kernel writes
 * the sum of the squares of what it stored of each output tile to aux_out[8 * t + w] (t = the tile's id inside the launch,
 * w = wave 0..7; slots of ids that name no tile are not written: zero the table first).  The squared gradient norm of
 * clip_grad_norm_ (trainer.py:356-360) then needs no pass over dW2: GHN3_OP_SUMSQ adds the slots.  (ABI v16) */
#define GHN3_GEMM_SUMSQ 16u
/* With GHN3_GEMM_X3 on tile codes 44 / 45: the pieces are F16 instead of bf16 -- a = a_hi + a_lo with 11 + 11 bits of
 * mantissa (~2^-22, fp32-grade) instead of 8 + 8 (~2^-17) on v_mfma_f32_16x16x32_f16, same three products, same rate.  B / B2
 * are then the f16 pieces of W * 2^GHN3_X3F16_WSHIFT written by GHN3_CAST_SPLIT_F16 (weights of ~1e-2 sit in the middle of the
 * f16 range, the lo pieces stay normal numbers) and the caller folds 2^-GHN3_X3F16_WSHIFT into alpha.  f16 has 5 exponent bits:
 * for O(1) operands only -- the FORWARD linears of the Graphormer (LayerNorm / attention / GELU outputs); gradient operands
 * (~1e-7) keep the bf16 pieces.  Why: the node embeddings then carry fp32-grade rounding, and ReLU masks of the decoders on a
 * knife edge flip ~5x less often against the fp32 path (docs/EXPERIMENTS.md, round 5).  (ABI v18) */
#define GHN3_GEMM_X3F16 32u
#define GHN3_X3F16_WSHIFT 6

typedef struct ghn3_gemm_problem {
    ghn3_ref A, B, C;
    ghn3_ref bias, residual, aux_in, aux_out;
    ghn3_ref a_gather, b_gather, c_gather;     /* int32 index arrays */
    int32_t M, N, K;
    int32_t lda, ldb, ldc;
    int32_t a_mode, b_mode;
    int32_t a_q, a_s, b_q, b_s, c_q, c_s;
    int32_t bias_q, bias_s, bias_stride;
    int32_t act, dact, flags;
    float alpha;
    int32_t ksplit;     /* > 1: split the K range into `ksplit` chunks whose partial sums are ADDED atomically to C
                         * (the program must zero C first; no epilogue but alpha is allowed) */
    int32_t b_kq, b_ks; /* GHN3_GEMM_OP16 only: k-map of B (see above) */
    /* GHN3_GEMM_OP16 only: ragged extents.  `lim` = int32 array with one entry per 128 rows of M (rows sorted by
     * decreasing extent).  lim_kind 1: entry = number of valid columns of those rows -- tiles whose first column
     * is beyond the extent of all their rows are skipped (forward of decoder groups stacked along M: every row
     * needs only the W2 rows o' < o of its own group; columns between a row's own extent and its tile's extent are
     * written with don't-care values).  lim_kind 2: entry = valid reduction length of those rows (A holds zeros
     * beyond it): the K loop of a tile stops at the largest extent of its rows (dgrad of the same stacking). */
    ghn3_ref lim;
    int32_t lim_kind;
    /* GHN3_GEMM_OP16 only: 0, or x + 1 to run every tile of this problem on XCD x (workgroup b of a launch runs on XCD
     * b % 8, each XCD has a private 4 MB L2).  Meant for the K chunks of a long reduction (the W2 dgrad): with one chunk
     * per XCD the chunk's slices of A and B are fetched by ONE L2 instead of by all eight.  Pinned problems need
     * ksplit <= 1; a launch honours the pins when it holds at least 8 problems. */
    int32_t xcd_pin;
    /* GHN3_GEMM_OP16 only, optional: device float holding the running max |x| of the fp32 source of an operand copy
     * that GHN3_OP_CAST16 scaled by a power of two (GHN3_CAST_SCALED): alpha is divided by that scale
     * (2^(11 - e) for amax = m 2^e; exact).  f16 copies of ~1e-6 gradients would otherwise be subnormal. */
    ghn3_ref alpha_amax;
    /* Optional row prologue of A (exact-fp32 small-problem kernel: ROW-mode A without gather, tile 32, K <= 4096; staged
     * split-bf16 kernels: GHN3_GEMM_X3 problems with tile code 44 / 45, K <= 384):
     * the LayerNorm that produces A is applied while the operand is staged, so that the latency-bound Graphormer
     * chain needs no separate LayerNorm launch (every workgroup normalises its own 32 rows; the column-tile-0
     * workgroups also write the by-products the backward needs).
     *   ln_kind 1 (forward, graphormer.py:239-241 / F.layer_norm): A' = (A - mean) * rstd * p0 + p1
     *       p0 = gamma[K] p1 = beta[K] p2 = mean out[M] p3 = rstd out[M] p4 = A' out [M][K] (p2..p4 optional)
     *   ln_kind 2 (backward): A = dy, A' = rstd * (dy*gamma - s1 - xhat*s2) + res, s1 = mean_k(dy*gamma),
     *       s2 = mean_k(dy*gamma*xhat), xhat = (x - mean) * rstd
     *       p0 = gamma[K] p1 = x [M][K] p2 = mean[M] p3 = rstd[M] p4 = res [M][K] or absent p5 = A' out [M][K] or absent
     * All row-major with leading dimension lda. */
    ghn3_ref ln_p[6];
    int32_t ln_kind;
    float ln_eps;
    /* GHN3_GEMM_X3 only: the lo copy of B and the K slice a workgroup stages at a time */
    ghn3_ref B2;
    int32_t x3_slice, _pad3;
    /* GHN3_GEMM_OP16, tile code 28 only, optional: row-tile table, `n_mtiles` int32 triples {m0, mi, extent}.  Row tile t
     * covers rows [m0, m0 + 32 * mi) with mi in {2, 4, 6, 7, 8, 9, 10} (64 .. 320 rows; ABI v19 -- until v18 the code counted
     * 64-row units and was 3, 4 or 5; tiles must not overlap, rows beyond M are ignored) and `extent` replaces the per-128-row
     * `lim` entries for it: valid columns (lim_kind 1) or valid reduction length (lim_kind 2) of the tile's rows; lim_kind 0
     * ignores it.  Lets the host cut a stacked decoder family at its extent boundaries with little row padding and give the
     * row tiles of one streamed W2 panel EQUAL heights (533 full-width rows = 288 + 288).
     * Absent: 256-row tiles and `lim` as for the other 16-bit-operand kernels. */
    ghn3_ref mtiles;
    int32_t n_mtiles, _pad4;
} ghn3_gemm_problem;

/* ---- 16-bit operand copies (GHN3_OP_CAST16) -----------------------------------------------------------
 * One descriptor = one fp32 matrix src[rows][cols] (leading dimension ld_src floats) converted to 16 bit:
 *   GHN3_CAST_STRAIGHT    dst [r][c]  = cvt(src[r][c])   r < rows, c < round_up(cols, 64)   (zeros for c >= cols)
 *   GHN3_CAST_TRANSPOSED  dstT[c][r]  = cvt(src[r][c])   c < cols, r < round_up(rows, 64)   (zeros for r >= rows)
 *   GHN3_CAST_COLSUM      dbias[bmap(c)] += sum_r src[r][c]  (fp32, atomics) with bmap(c) = (c / bias_q) * bias_s +
 *                         c % bias_q + bias_off (bias_q == 0: c + bias_off) -- the fused bias gradient of the
 *                         wgrad that reads dstT.
 * Source column map: with src_q > 0 logical column c reads src[r][(c / src_q) * src_s + c % src_q] (src_q, src_s
 * multiples of 4) -- a band of input channels i' in [i_lo, i_lo + src_q) of every output channel o' of a decoder tile
 * row (o' * i + i'), the wgrad operand layout of program.py.
 * GHN3_CAST_TIGHT: the transposed copy writes r < round_up(rows, 8) only (zeros for r >= rows), so that matrices can be
 * concatenated along r at multiples of 8 instead of 64; the caller keeps the remaining padding zero.
 * The type of each copy is f16 unless its *_BF16 flag is set.  Offsets: src_off in floats from r0, dst_off / dstT_off
 * in 16-bit elements from r1; ld_dst >= round_up(cols, 64), ld_dstT >= round_up(rows, 64), both multiples of 8. */
#define GHN3_CAST_STRAIGHT 1u
#define GHN3_CAST_TRANSPOSED 2u
#define GHN3_CAST_STRAIGHT_BF16 4u
#define GHN3_CAST_TRANSPOSED_BF16 8u
#define GHN3_CAST_COLSUM 16u
/* multiply by the power-of-two scale derived from the op's r4 (running max |x|, see ghn3_gemm_problem::alpha_amax)
 * before converting; column sums stay unscaled */
#define GHN3_CAST_SCALED 32u
#define GHN3_CAST_TIGHT 64u
/* bf16 split copies for GHN3_GEMM_X3: the straight and / or transposed copy is written twice, hi = bf16(x) at
 * dst_off / dstT_off and lo = bf16(x - hi) `lo_off` 16-bit elements behind it (both bf16 whatever the *_BF16 flags) */
#define GHN3_CAST_SPLIT 128u
/* with GHN3_CAST_COLSUM: deterministic column sums -- every 64-row tile of the descriptor writes its partial sums to
 * dbias[part_off + tile_row * cols + c] (plain stores, no atomics); GHN3_OP_ROWSET_COLSUM adds them in a fixed order */
#define GHN3_CAST_COLSUM_PARTS 256u
/* with GHN3_CAST_SPLIT: the straight and / or transposed hi / lo copies are written in FRAGMENT-MAJOR order instead of
 * row-major (ld_dst / ld_dstT unused): element (n, k) of the copied [n][k] matrix -- the straight copy has n = source row,
 * k = source column, the transposed one n = source column, k = source row -- lives at
 *     ((n / 16) * (K / 32) + k / 32) * 512 + ((k % 32) / 8 * 16 + n % 16) * 8 + k % 8          (16-bit elements)
 * i.e. the 16 x 32 operand of one v_mfma_f32_16x16x32_bf16 is one contiguous kilobyte in lane order.  rows and cols must be
 * multiples of 32.  Operand layout of the staged split-bf16 kernels (GHN3_OP_GEMM tile codes 44 / 45).  (ABI v14) */
#define GHN3_CAST_FRAG 512u
/* the source is a 16-bit matrix inside the destination buffer: src_off / ld_src in 16-bit elements from r1, element type =
 * type of the copies written; only the transposed copy (and the column sums) may be requested.  With GHN3_CAST_SCALED the
 * source already carries the power-of-two scale of the op's r4: values are re-laid out bit for bit, column sums are divided by
 * the scale.  (The transposed weight-gradient operands made from the 16-bit tile gradient GHN3_OP_TILE_BWD wrote.)  (ABI v15) */
#define GHN3_CAST_SRC16 1024u
/* with GHN3_CAST_SPLIT: the STRAIGHT hi / lo copies are f16 pieces of x * 2^GHN3_X3F16_WSHIFT (operand B of GHN3_GEMM_X3F16
 * problems); the transposed copies stay bf16 pieces of x.  (ABI v18) */
#define GHN3_CAST_SPLIT_F16 2048u
typedef struct ghn3_cast_desc {
    int64_t src_off, dst_off, dstT_off;
    int32_t rows, cols;
    int32_t ld_src, ld_dst, ld_dstT;
    uint32_t flags;
    int32_t bias_q, bias_s;
    int32_t block_start;     /* first workgroup of this descriptor: blocks are 64 x 64 source tiles, column-tile fastest */
    int32_t bias_off;
    int32_t src_q, src_s;    /* source column map (0: identity) */
    int64_t lo_off;          /* GHN3_CAST_SPLIT: distance (16-bit elements) from a hi copy to its lo copy */
    int64_t part_off;        /* GHN3_CAST_COLSUM_PARTS: float offset (from r3) of this descriptor's [row tiles][cols] slots */
} ghn3_cast_desc;

/* ---- tile / normalise descriptors  (nn.py:422-506 _tile_params, 554-592 _normalize, 508-552 _set_params)
 * dst[a,b,c,d] = f( src[src_off + (a%E0)*S0 + (b%E1)*S1 + (c%E2)*S2 + (d%E3)*S3] )
 * f: mode 0  x*scale          (fan-in scaling, nn.py:571-583; scale 1 for positional encodings 566-569)
 *    mode 1  2*sigmoid(x/2)   (1-D weights, nn.py:588)
 *    mode 2  tanh(x/5)        (1-D biases,  nn.py:590)
 */
typedef struct ghn3_tile_desc {
    int64_t dst_off;        /* element offset into the flat predicted-parameter buffer */
    int64_t src_off;        /* element offset into the source buffer */
    int64_t S[4];           /* source strides (elements) */
    int32_t T[4];           /* target extents (leading dims padded with 1) */
    int32_t E[4];           /* modulo extents: E[i] <= T[i]; the consumed source region is E0 x E1 x E2 x E3 */
    int32_t R[4];           /* full source-region extents (>= E): backward writes zeros outside E */
    int32_t src_buf;        /* index into the op's source table r[1..] */
    int32_t mode;
    float scale;
    int32_t _pad;           /* row blocks: number of second-dimension (i) elements per block */
} ghn3_tile_desc;

/* ---- ops ------------------------------------------------------------------------------------------- */
enum ghn3_op_kind {
    GHN3_OP_NOP = 0,
    /* i: first_problem, n_problems, tile (0 auto / 32 / 64 / 128 for fp32 operands; 0 auto / 16 = 128x128 / 24 = 256x256 /
     * 20 = 256x128 with a three-stage ring / 25 = persistent 256x256 for output-heavy PLAIN problems (C = alpha A B^T,
     * optional row map of C: short K, e.g. the W2 weight gradient) / 28 = the 8-phase kernel, (64 .. 320) x 256 tiles in steps
     * the row-tile table names (`mtiles`), no split-K / gathers (problems of the op with fewer than 160 rows or with
     * ksplit > 1 fall back to 16) for
     * GHN3_GEMM_OP16 problems), grid cap for GHN3_GEMM_OP16 launches (0 = one workgroup per tile; > 0: at most
     * that many CUs' worth of persistent workgroups -- a side-stream GEMM that should leave CUs to the chain it runs beside).
     * Tile codes 28 / 29 take the quotient of the row map of C (c_q, c_s) from one 32-bit multiply-high: M * c_q < 2^32, else
     * GHN3_E_ARG.  29 = the persistent 8-phase kernel for the W2 weight gradient (plain problems like 25; GHN3_GEMM_SUMSQ).
     * 30 (ABI v19) = the same stream on 256 x 128 tiles with two accumulator sets: the stores of a finished tile leave during
     * the next tile (contract of 29; the slot table of GHN3_GEMM_SUMSQ is indexed by ITS tile ids, 128-column tiles).
     * fp32 operands additionally: 48 = the split-bf16 weight-gradient kernel (both operands GHN3_MODE_COL fp32 activations,
     * reduction over rows: dW = dY^T X of the Graphormer linears, graphormer.py:208-248 backward; GHN3_GEMM_ACCUM and
     * GHN3_GEMM_BIASGRAD allowed, M % 4 == 0, N % 4 == 0, no gathers / maps / activation / residual / split-K -- problems of
     * the op that do not qualify run on the exact 64 x 64 tiles; ~8e-6 relative like GHN3_GEMM_X3) */
    GHN3_OP_GEMM = 1,                 /* every nn.Linear / F.linear on the path */
    /* graphormer.py:229-237 -- degree counts of A==1, A[0,:], fw/bw pair index
     * r0=A(int64 B,N,N) r1=deg_in r2=deg_out r3=dist0 (int32 B,N) r4=pair (int32 B,N,N); i: B,N,V */
    GHN3_OP_GRAPH_PROLOGUE = 2,
    /* nn.py:248-253 + graphormer.py:230-235: type/shape/centrality/input-dist embeddings, dense pad, mask
     * r0=x out (B,N,C) r1=node_type r2=shape_idx(4 per node) r3=n_nodes r4=node_off (int32)
     * r5..r9 = tables E_type,E_ch,E_sp,E_in,E_out r10=E_dist r11=deg_in r12=deg_out r13=dist0; i: B,N,C */
    GHN3_OP_EMBED_NODES = 3,
    /* graphormer.py:115-117 factorised: hid[fw*V+bw][c] = relu(Pfw[fw][c] + Pbw[bw][c])
     * r0=hid r1=Pfw r2=Pbw ; i: V,C */
    GHN3_OP_EDGE_HIDDEN = 4,
    /* bias[b,h,i,j] = T[pair[b,i,j]][h] ; r0=bias r1=T r2=pair ; i: B,N,H */
    GHN3_OP_BIAS_GATHER = 5,
    /* F.layer_norm ; r0=y r1=x r2=gamma r3=beta r4=mean r5=rstd r6=addend planes or absent ; i: rows,C,n_planes
     * (0 with r6 present = 1), floats between planes ; f0=eps
     * With r6 the normalised rows are x + sum_p r6[p] (summed in plane order) and the sum is written back to x (the
     * planes are the further K slices of the GEMM that produced x, split over several workgroup sets). */
    GHN3_OP_LAYERNORM_FWD = 6,
    /* graphormer.py:121-140 ; r0=out(B*N,C) r1=qkv(B*N,3C) r2=bias(B,H,N,N) r3=P save or absent r4=n_nodes
     * i: B,N,C,H */
    GHN3_OP_ATTN_FWD = 7,
    /* r0=flat out, r1..r6 = sources, r7 = descriptors (device)
     * i: n_desc, n_work_blocks, byte offset from r7 to the int64 (descriptor, start) work-block table, LDS bytes
     * A work block (k, start) with k >= 0 covers TILE_CHUNK (2048) consecutive elements of descriptor k; a block
     * (~k, a0 | i0 << 24) with ~k < 0 is a ROW block: elements [i0, i0 + _pad) of row a0 of descriptor k moved through LDS (4-D
     * kernels with kh * kw > 1, mode 0, S[1] == 1, E[2..3] == T[2..3] == R[2..3]); i3 = max floats * 4 it needs
     * r8 = optional float slots, one per work block: the block's sum of squares of what it wrote (the per-tensor Frobenius
     * norms of trainer.py:288-294 then need no pass over the 346 MB output: GHN3_OP_PARAM_NORM_FIN adds a tensor's slots)
     * r9 = optional second set of slots (with r8): max |value written| * replicas * |scale| of the block for mode-0
     * descriptors of source 0, else 0 -- the ingredients of GHN3_OP_TILE_BWD's a-priori output bound (ABI v15) */
    GHN3_OP_TILE_FWD = 8,
    /* sum over predicted tensors of ||p||_F (trainer.py:97-98,288-294)
     * r0=loss(1 float, accumulated) r1=flat r2=seg_off(int64 (begin,end) pairs, sorted by begin, disjoint)
     * r3=norms(n floats) r4=optional int32 table: first segment whose end lies beyond float 8192 * b, for every
     * 8192-float chunk b of the flat buffer (absent: searched on the device) r5=optional partial-sum slots (n_seg +
     * number of chunks floats; needs r4): per-chunk partial sums added in chunk order instead of with float atomics
     * i: n_seg, extent of the flat buffer in floats (>= the last end; the passes stream it) */
    GHN3_OP_PARAM_NORM_FWD = 9,
    /* r0=dflat r1=flat r2=seg_off r3=norms r4=optional first-segment table ; i: n_seg, flat extent ; f0 = upstream grad */
    GHN3_OP_PARAM_NORM_BWD = 10,
    /* r0=dflat, r1..r6 = source buffers (values), r7=desc, r8..r12 = source-grad buffers, r13 = optional device float
     * that receives the running max |x| of everything written to source-grad buffer 0 (the decoder tiles)
     * i: n_desc, n_work_blocks, byte offset from r7 to the backward work-block table, LDS bytes (row blocks)
     * Fused predicted-parameter-norm loss (trainer.py:97-98,288-294: loss += w sum_t ||p_t||_F): with r14 = per-tensor norms
     * (GHN3_OP_PARAM_NORM_FIN) the upstream gradient of element e of tensor t is  dflat[e] + g * out[e] / norm_t  with
     * r15 = the flat predicted buffer `out`, g = the device float r6 (source slot 5 is never a tile source) and i4 = byte
     * offset from r7 to the int32 descriptor -> tensor table; r0 (dflat) may then be absent: the norm term alone.  The
     * 346 MB gradient of the norm term is never materialised.
     * Direct 16-bit tiles (ABI v15; norm term alone, r0 absent, r13 present): i5 > 0 = byte offset from r7 to a table of three
     * int64 per descriptor {h, rel0, ld32 | ld16 << 32}.  A descriptor of source 0 with h != INT64_MIN writes its gradient
     * not as fp32 but as the scaled 16-bit operand copy (i6 = 0 f16 / 1 bf16) the W2 GEMMs consume: fp32 element rel0 + s of a
     * row-major matrix with row stride ld32 (s = the descriptor's source offset minus src_off) goes to 16-bit element
     * h + (rel / ld32) * ld16 + rel % ld32 behind r8.  The scale is the power of two of ghn3_gemm_problem::alpha_amax for bound = |g| * norms[-1]: the
     * float in front of r14 is GHN3_OP_PARAM_NORM_FIN's r6, an upper bound of every |element| the op writes to source-grad 0
     * divided by |g|.  The op stores `bound` to r13 (instead of a measured maximum), where GHN3_CAST_SCALED copies and
     * ghn3_gemm_problem::alpha_amax look the scale up.  One pass and 2 bytes per element instead of three passes and 8. */
    GHN3_OP_TILE_BWD = 11,
    /* column sums: out[omap(n)] += sum_m X[g(m)][n] ; r0=out r1=X r2=row gather (int32) or absent
     * i: M,N,ld,q,s,stride,accum(must be 1) ; omap(n) = ((n/q)*s + n%q)*stride (q == 0: n*stride) */
    GHN3_OP_COLSUM = 12,
    /* segmented row sum: out[r][:] (+)= sum_{t in seg(r)} X[idx[t]][:]
     * r0=out r1=X r2=seg_ptr(int32 rows+1) r3=idx(int32) ; i: rows,C,ldx,ldo,accum */
    GHN3_OP_ROWSEG_SUM = 13,
    /* r0=dx r1=dy r2=x r3=gamma r4=mean r5=rstd r6=residual grad or absent r7=addend planes of dy or absent
     * (dy + sum_p r7[p] is written back to dy) ; i: rows,C,n_planes (0 with r7 present = 1), floats between planes */
    GHN3_OP_LAYERNORM_BWD = 14,
    /* r0=dgamma r1=dbeta r2=dy r3=x r4=mean r5=rstd ; i: rows,C,accum */
    GHN3_OP_LN_PARAM_GRAD = 15,
    /* r0=dqkv r1=dO r2=qkv r3=P r4=O (saved attention output) r5=optional device float: running max of |dBias| as
     * written by this launch (atomic max on a zeroed slot -- the program passes it to the LAST launch that accumulates into
     * r6, whose values are final: GHN3_OP_BIAS_HIST then needs no pass for its scale; ABI v15) r6=dBias (accumulated)
     * r7=n_nodes ; i: B,N,C,H, i4 = 1: the general kernel (operands straight from memory) also where the LDS-staged kernel for
     * graphs of up to 256 nodes would run -- both give the same bits; tests compare them */
    GHN3_OP_ATTN_BWD = 16,
    /* dT[p][h] += sum_{pair==p} dBias[b,h,i,j] ; r0=dT r1=dBias r2=pair r3=scratch: 8 * V * V * H + 16 bytes, ZEROED by
     * the program (64-bit fixed-point histogram: the order of the atomics does not matter -> deterministic) ; i: B,N,H,V,
     * i4 = 1: the float behind the histogram (r3 + 8 * V * V * H bytes) already holds max |dBias| (GHN3_OP_ATTN_BWD r5) */
    GHN3_OP_BIAS_HIST = 17,
    /* r0=dPfw r1=dPbw r2=dhid(in, masked in place) r3=hid ; i: V,C */
    GHN3_OP_EDGE_HIDDEN_BWD = 18,
    /* scatter-add of dx(B,N,C) into the six embedding tables; refs as EMBED_NODES with r0=dx and
     * r5..r10 = table gradients ; i: B,N,C, rows of E_type, E_ch, E_sp (one workgroup per table row gathers its nodes in
     * node order: deterministic) */
    GHN3_OP_EMBED_BWD = 19,
    /* r0=dst ; i0 = bytes */
    GHN3_OP_MEMSET0 = 20,
    /* r0=dst r1=src ; i0 = n floats ; dst += src */
    GHN3_OP_ADD = 21,
    /* in place: X[m][n] *= dact(aux[m][n]) ; r0=X r1=aux ; i: M,N,ld, dact (GHN3_DACT_*) -- the deferred
     * epilogue of a split-K dgrad GEMM; r2 = optional device float receiving the running max |X| after masking.
     * Plane-wise splits (no atomics, deterministic): r3 = further partial planes, i4 = their number, i5 = floats between
     * planes, i6 = rows that have them: X[m][n] = dact(X[m][n] + sum_p r3[p * i5 + m * N + n]) for m < i6 (N == ld) */
    GHN3_OP_DACT = 22,
    /* fp32 -> f16 / bf16 operand copies for GHN3_GEMM_OP16 problems (straight and / or transposed, zero padded)
     * r0=src base (fp32) r1=dst base (16-bit) r2=ghn3_cast_desc table (device) r3=dbias or absent
     * r4=running max |x| (device float) for descriptors flagged GHN3_CAST_SCALED, or absent
     * i: n_desc, total work tiles, grid cap (0 = one workgroup per tile; > 0: at most that many workgroups stride
     * over the tiles -- side-stream copies that should leave HBM bandwidth to the chain they run under) */
    GHN3_OP_CAST16 = 23,
    /* i0 = 0: the run's stream waits for every GHN3_OPFLAG_SIDE op issued so far (no refs).
     * i0 = 1: MARK -- remember the side-stream work issued so far under id i1 (0..3); i0 = 2: the run's stream waits for the
     * work marked i1 only, later side-stream work keeps running (a short side-stream branch that rejoins the chain while a
     * long one, e.g. the W2 weight gradient, continues).  Marks belong to the context and survive a run that ends in
     * GHN3_OP_DETACH, so the wait may be issued by a LATER run (the parts of a data-parallel backward); a full join clears
     * them, a wait without a pending mark is a no-op.  A run without side-stream ops / JOIN / DETACH leaves the side-stream
     * state of the context untouched.  (ABI v14) */
    GHN3_OP_JOIN = 24,
    /* as the LAST op of a run (GHN3_OP_NOP padding behind it does not count): return without joining the side stream; the pending side work is joined by the next
     * ghn3_run on the context (or observed with ghn3_ctx_side_wait).  Lets a caller split a program in two runs and
     * start the gradient all-reduce of the first part's weight gradients while the second part executes. */
    GHN3_OP_DETACH = 25,
    /* trainer step over the flat buffers (trainer.py:356-381)
     * SUMSQ: r0[0] += sum x^2 (r1 = x, i0 = n floats; r2 = optional 4096 floats of scratch: per-workgroup partial sums
     *        added in a fixed order instead of with float atomics) -- the squared global gradient norm of clip_grad_norm_.
 *        i1 < i2: the floats [i1, i2) of x are left out (multiples of 4; r2 must then hold 8192 floats) and r3 = i3 partial
 *        sums a producer of that range left (GHN3_GEMM_SUMSQ slots of the W2 weight gradient) are added instead (ABI v16)
     * ADAMW: torch.optim.AdamW on r0 = params, r1 = grads, r2 = exp_avg, r3 = exp_avg_sq (i0 = n floats);
     *        r4 = squared gradient norm or absent, f0 = max_norm (<= 0: no clipping): the gradient is scaled by
     *        min(1, max_norm / (norm + 1e-6)); a non-finite r4 skips the whole update (NaN guard, trainer.py:240-257);
     *        f1 = 1 / loss scale (0 = 1): the gradients arrive multiplied by the AMP loss scale (trainer.py:346-352); i1..i7 = IEEE-754 double bit patterns of lr, beta1, beta2, eps,
     *        weight_decay, 1 - beta1^t, 1 - beta2^t */
    GHN3_OP_SUMSQ = 26,
    GHN3_OP_ADAMW = 27,
    /* ReLU of the classifier tiles (nn.py:755-758 `relu(x)` in front of class_layer_predictor) with EXACT masks in the
     * 16-bit modes: X[r][c] = relu(X[r][c]) in place, except that elements the 16-bit GEMM left within f0 * rms of zero
     * (rms over their 1024-element chunk; ~0.4 % of the elements) are first recomputed in fp32,
     * X[r][c] = relu(dot(W[wrow(c)][0..K), U[r][0..K)) + bias[wrow(c)]), wrow(c) = (c / q) * s + c % q.
     * A sign decided by a 3e-4 rounding error would otherwise flip the ReLU mask of that element in the backward -- an
     * O(1) error of a whole dW2 row.  r0=X [rows][ld] r1=U [rows][K] r2=W [.][K] r3=bias or absent
     * i: rows, cols, ld, K, q, s ; f0 = threshold factor (<= 0: plain ReLU) */
    GHN3_OP_RELU_FIX = 28,
    /* bias gradient of a decoder linear from the stacked row sets of its output gradient (deterministic, one writer
     * per output): out[o' * I + i'] += sum_{sets: o' < o_s, i' < i_s} sum_{r < rows_s} X[off_s + r * ld_s + o' * i_s + i']
     * (+ i0_s on the output column); r0=out r1=X (fp32 base) r2=table of {int64 off (floats from r1); int32 rows, o, i,
     * ld, i0, pad} ; i: n_sets, O, I */
    GHN3_OP_ROWSET_COLSUM = 29,
    /* Local passes of the data-parallel gradient exchange (trainer.py:136: DistributedDataParallel's all-reduce of the GHN
     * gradients; here all-to-all + local sum + all-gather on the flat gradient buffer, ghn3_amd/ddp_utils.py).
     * WIRE_PACK:   r0 = dst (i1 16-bit elements) r1 = src (fp32) ; i0 = n valid, i1 = n padded (dst[n..i1) = 0), i2 = 1:
     *              the reverse, dst fp32 [i0] <- src bf16 (the gathered result back into the gradient buffer)
     * RANK_REDUCE: out[e] = f0 * sum_{w < i1} in[w * i0 + e], summed in fp32 in rank order (every rank computes the same
     *              bits); r0 = out r1 = in ; i0 = elements per rank, i1 = ranks W, i2 = 1: `in` is bf16, i3 = 1: `out` is bf16 */
    GHN3_OP_WIRE_PACK = 30,
    GHN3_OP_RANK_REDUCE = 31,
    /* batched fp32 transpose: dst[b][c][r] = src[b][r][c], r < i0 rows, c < i1 cols; i2 = ld_src, i3 = ld_dst, i4 = batch,
     * i5 / i6 = floats between batches in src / dst.  Utility (k-contiguous fp32 copies of k-strided operands; the compiled
     * programs of this round use the 16-bit transposed copies of GHN3_OP_CAST16 instead). */
    GHN3_OP_TRANSPOSE32 = 32,
    /* per-tensor Frobenius norms from the per-work-block sums of squares GHN3_OP_TILE_FWD left (r8 there), added in block
     * order (deterministic): norms[t] = sqrt(sum_{b in [first[t], first[t + 1])} parts[b]); loss[0] = sum_t norms[t] (fixed
     * order).  r0=loss (1 float, overwritten) r1=norms (n floats) r2=parts r3=first (int32, n + 1 entries) ; i0 = n tensors.
     * Replaces GHN3_OP_PARAM_NORM_FWD's pass over the flat output (trainer.py:288-294).  (ABI v14)
     * r4 = optional bound slots of GHN3_OP_TILE_FWD (r9 there), with r5 = scratch (n floats) and r6 = 1 float that receives
     * max_t (max of t's bound slots) / norms[t]  (GHN3_OP_TILE_BWD expects r6 == r1 - 1 float).  (ABI v15) */
    GHN3_OP_PARAM_NORM_FIN = 33,
    /* GHN3_OP_LN_PARAM_GRAD for many LayerNorms in ONE launch (the 2 L LayerNorms of the Graphormer, deferred behind the
     * chain): item t adds  dgamma_t[c] += sum_r dy_t[r][c] xhat_t[r][c],  dbeta_t[c] += sum_r dy_t[r][c].
     * r0 = base of the parameter gradients, r1 = base of the activations, r2 = table of 6 int64 per item: float offsets
     * {dgamma, dbeta} from r0 and {dy, x, mean, rstd} from r1 ; i: n_items, rows, C.  One writer per element, fixed order. */
    GHN3_OP_LN_PARAM_GRAD_BATCH = 34,
    /* GHN3_OP_ADAMW applied to the SOURCE elements of a cast-descriptor table, which are then cast like GHN3_OP_CAST16 does:
     * the 16-bit operand copies of a weight follow its optimizer update (trainer.py:379) without a second pass over it
     * (decoder.conv.2.weight: 1.8 GB read again + a 1 ms side-stream launch in front of every training forward otherwise).
     * r0..r4 as GHN3_OP_ADAMW (the descriptors' src_off are float offsets from r0 AND from r1..r3: congruent flat buffers);
     * r5 = 16-bit destination base, r6 = ghn3_cast_desc table; i0 = n_desc + (work tiles << 32); i1..i7, f0, f1 as
     * GHN3_OP_ADAMW.  Descriptors: fp32 source, cols % 4 == 0, no column map / scale / column sums; every source element
     * belongs to exactly one descriptor.  A non-finite r4 leaves parameters, moments and copies untouched.  Same arithmetic
     * per element as GHN3_OP_ADAMW (bit-identical parameters).  (ABI v16) */
    GHN3_OP_ADAMW_CAST16 = 35,
    GHN3_OP_KIND_COUNT
};

typedef struct ghn3_op {
    int32_t kind;
    int32_t flags;
    int64_t i[8];
    float f[4];
    ghn3_ref r[16];
} ghn3_op;

typedef struct ghn3_ctx ghn3_ctx;

/* Library identity. */
int ghn3_abi_version(void);
const char* ghn3_last_error(void);

/* Context: owns a small pinned+device staging area for resolved GEMM problem tables. */
int ghn3_ctx_create(ghn3_ctx** out);
void ghn3_ctx_destroy(ghn3_ctx* ctx);

/* Arithmetic type of the MFMA operands of GHN3_OP_GEMM (accumulation is always fp32; operands stay fp32
 * in HBM and are converted while staged through LDS).  A GEMM op may override the context default with
 * (op.flags & 0xff) = 1 + GHN3_CT_*; 0 keeps the default. */
enum { GHN3_CT_F32 = 0, GHN3_CT_F16 = 1, GHN3_CT_BF16 = 2 };
int ghn3_ctx_set_compute_type(ghn3_ctx* ctx, int ctype);

/*
 * Run ops[0..n_ops) on `stream`.  `problems` is the host array the GEMM ops index into; `bufs` is the
 * table of device pointers the refs index into.  Replaces the body of GHN3.forward after
 * _map_net_params (nn.py:247-328) and, for backward programs, the autograd graph of the same lines.
 * `stream` is a hipStream_t passed as void*.
 */
int ghn3_run(ghn3_ctx* ctx, const ghn3_op* ops, int n_ops,
             const ghn3_gemm_problem* problems, int n_problems,
             void* const* bufs, int n_bufs, void* stream);

/* Make `stream` wait for everything issued so far on the context's side stream (GHN3_OPFLAG_SIDE ops), e.g. the
 * communication stream that all-reduces weight gradients produced there. */
int ghn3_ctx_side_wait(ghn3_ctx* ctx, void* stream);

/* 1 when the last run that used the side stream ended in GHN3_OP_DETACH and no later run has joined it yet (its
 * side-stream work may still be executing), 0 otherwise, < 0 without a context.  (ABI v19; lets a caller -- and the tests
 * of FusedAdamW.step(overlap=True), optim.py -- verify that a run really returned detached.) */
int ghn3_ctx_side_pending(ghn3_ctx* ctx);

/* ghn3_run keeps the resolved problem tables of its last 16 distinct runs on the device: a run whose ops, problem table and
 * buffer pointers equal (byte for byte) those of a kept one -- every step of a loop once the caller's allocator has settled --
 * skips the host resolve and the table upload.  Counts runs with GEMM ops served from / added to that store since the context
 * was created (either pointer may be NULL).  GHN3_RUN_CACHE=0 in the environment turns the store off.  (ABI v19) */
int ghn3_ctx_cache_stats(ghn3_ctx* ctx, int64_t* hits, int64_t* misses);

/* Timing helper for bench.py: HIP events on `stream` (torch.cuda.Event only sees torch's current stream).
 * ghn3_event_elapsed_ms synchronises on the stop event. */
int ghn3_event_create(void** ev);
int ghn3_event_record(void* ev, void* stream);
int ghn3_event_elapsed_ms(void* start, void* stop, float* ms);
int ghn3_event_destroy(void* ev);

/* Per-op timing with HIP events on the run's stream.
 *   mode 1: every op is bracketed and synchronised (diagnostic; perturbs timing) -> ghn3_profile_read
 *   mode 2: only ops whose flags carry GHN3_OPFLAG_TIMED get an event pair from a pool, nothing is
 *           synchronised until ghn3_profile_read_tags; the tag is (op.flags >> 16) & 255.
 *   mode 3: as mode 2, but GHN3_OPFLAG_SIDE ops run on the caller's stream in program order (nothing co-runs with a timed
 *           kernel, side-stream grid caps are dropped): the kernels' own durations, for roofline figures. */
#define GHN3_OPFLAG_TIMED 0x100
/* Run the op on the context's side stream, concurrently with the following ops of the program: it starts after
 * every earlier op of the program has finished and is waited for by GHN3_OP_JOIN / the end of ghn3_run.  The
 * program must not let later main-stream ops touch what a pending side op reads or writes (used for weight
 * gradients and operand copies that are off the critical path).  GHN3_NO_SIDE_STREAM=1 serialises everything; a serialised
 * side GEMM ignores its grid cap (op.i[3]: the cap only leaves CUs to the stream it would have run beside). */
#define GHN3_OPFLAG_SIDE 0x200
int ghn3_profile_enable(ghn3_ctx* ctx, int mode);
int ghn3_profile_read(ghn3_ctx* ctx, double* ms_per_kind /* [GHN3_OP_KIND_COUNT] */,
                      int64_t* launches_per_kind, int reset);
int ghn3_profile_read_tags(ghn3_ctx* ctx, double* ms_per_tag /* [256] */, int64_t* n_per_tag /* [256] */,
                           int reset);

/*
 * Target-network execution, first native slice: ReLU -> depthwise k x k convolution -> pointwise 1 x 1
 * convolution -> BatchNorm with batch statistics -- `DilConv` and each half of `SepConv` of the DeepNets-1M search space
 * (/root/reference/ghn3/ops.py:198-240, executed at trainer.py:308-319 with the weights the GHN predicted) -- forward and
 * backward on NHWC (torch channels_last) fp32 activations.  Replaces four ATen / MIOpen modules per direction.
 *   x [N][H][W][C_in], z (pre-norm) / out / dout [N][Ho][Wo][C_out], w_dw [C_in][ks][ks], w_pw [C_out][C_in],
 *   gamma / beta [C_out], stats [3 C_out] = mean | 1 / sqrt(var + eps) | biased variance (written by fwd, read by bwd).
 * Weights are read in place (views of the GHN's flat prediction buffer); all five gradients are written densely.
 * w_dw == NULL (with ks = 1, pad = 0; dw_dw then unused): ReLU -> 1 x 1 convolution (stride) -> BatchNorm, the `ReLUConvBN`
 * preprocessing layer of every cell and the `conv_1x1` op (ops.py:180-198).
 * `scratch` = ghn3_dwpw_scratch_floats(desc, backward) floats of device memory owned by the caller (no allocation, no
 * synchronisation inside).  Limits: C_in, C_out multiples of 4 and <= 512, ks <= 7, tensors below 2^31 elements
 * (GHN3_E_LIMIT otherwise: the caller keeps its stock path for such layers).  Deterministic.
 */
typedef struct ghn3_dwpw_desc {
    int32_t N, H, W, C_in, C_out, ks, stride, pad, dil, Ho, Wo;
    float eps;
} ghn3_dwpw_desc;
int64_t ghn3_dwpw_scratch_floats(const ghn3_dwpw_desc* desc, int backward);     /* < 0: bad descriptor (ghn3_last_error) */
int ghn3_dwpw_bn_fwd(const ghn3_dwpw_desc* desc, const float* x, const float* w_dw, const float* w_pw, const float* gamma,
                     const float* beta, float* z, float* out, float* stats, float* scratch, void* stream);
int ghn3_dwpw_bn_bwd(const ghn3_dwpw_desc* desc, const float* dout, const float* x, const float* z, const float* stats,
                     const float* w_dw, const float* w_pw, const float* gamma, float* dx, float* dw_dw, float* dw_pw,
                     float* dgamma, float* dbeta, float* scratch, void* stream);


/* ---- target-network layers, second slice (ABI v19, round 6): [ReLU ->] dense kh x kw convolution -> BatchNorm -------------
 * `ReLUConvBN` with a k x k kernel and its 1 x k / k x 1 halves (/root/reference/ghn3/ops.py:180-198; the `conv_3x3 / 5x5 /
 * 7x7` ops of the DeepNets-1M search space, ops.py:297), forward and backward, NHWC fp32 activations, batch statistics.
 * w [C_out][C_in][kh][kw] is the view of the GHN's flat prediction buffer (read in place; dw is written in the same order),
 * z / out [N Ho Wo][C_out], stats [3 C_out] as for ghn3_dwpw_bn_fwd; relu != 0 applies ReLU to x first (and its mask to dx).
 * Limits (GHN3_E_LIMIT: the caller keeps its stock path): C_in, C_out multiples of 4, C_out <= 512, C_in <= 4096, kernel <= 7 x 7.
 * Deterministic (fixed-order partial sums).
 * relu | GHN3_CONV_NO_NORM: the convolution alone (the first half of the 1 x k / k x 1 pair of ops.py:186-190, which has no norm
 * layer of its own): the forward writes its result to `z` and stops (gamma, beta, out, stats may be NULL); the backward takes
 * `dout` as the gradient of that result and writes dx and dw only (z, stats, gamma, dgamma, dbeta may be NULL). */
#define GHN3_CONV_NO_NORM 2
typedef struct ghn3_conv_desc {
    int32_t N, H, W, C_in, C_out, kh, kw, stride_h, stride_w, pad_h, pad_w, dil, Ho, Wo, relu;
    float eps;
} ghn3_conv_desc;
int64_t ghn3_conv_scratch_floats(const ghn3_conv_desc* desc, int backward);     /* < 0: bad descriptor (ghn3_last_error) */
int ghn3_conv_bn_fwd(const ghn3_conv_desc* desc, const float* x, const float* w, const float* gamma, const float* beta, float* z,
                     float* out, float* stats, float* scratch, void* stream);
int ghn3_conv_bn_bwd(const ghn3_conv_desc* desc, const float* dout, const float* x, const float* z, const float* stats,
                     const float* w, const float* gamma, float* dx, float* dw, float* dgamma, float* dbeta, float* scratch,
                     void* stream);

/* ---- target-network layers, third slice (ABI v19, round 6): squeeze-and-excitation with a hard-swish gate -----------------
 * Replaces `ChannelSELayer.forward` (/root/reference/ghn3/ops.py:239-274) executed at trainer.py:308-319 and torch autograd of it:
 *   s = mean over the pixels of x;  h = relu(W1 s + b1);  a = W2 h + b2;  y = x * hardswish(a)
 * x, y, dy, dx [N][HW][C] fp32 (NHWC); w1 [J][C], b1 [J], w2 [C][J], b2 [C] (the Linear layers' own layouts, read in place);
 * save [N][2 C + J] = (s | a | h) written by the forward for the backward; scratch [N][C + J] floats.
 * One workgroup per sample forward, the same + one parameter-gradient launch backward; fixed summation orders.
 * Limits (GHN3_E_LIMIT): C a multiple of 4, C and J <= 1024. */
int ghn3_se_fwd(int N, int HW, int C, int J, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                float* y, float* save, void* stream);
int ghn3_se_bwd(int N, int HW, int C, int J, const float* dy, const float* x, const float* w1, const float* w2, const float* save,
                float* dx, float* dw1, float* db1, float* dw2, float* db2, float* scratch, void* stream);

/* ---- target-network layers, pooling (ABI v19, round 6): k x k max / average pooling on NHWC activations ---------------------
 * Replaces `nn.MaxPool2d` / `nn.AvgPool2d(count_include_pad=False)` of the search space (/root/reference/ghn3/ops.py:289-291, the stems'
 * MaxPool2d at ops.py:452) executed at trainer.py:308-319, and their autograd.  mode 0 = average over the VALID taps, 1 = max (the first
 * maximum in (kh, kw) order, its tap kept in one byte per output element: idx [N Ho Wo C], forward output / backward input; NULL for
 * mode 0).  Floor mode: Ho = (H + 2 pad - k) / stride + 1.  Limits (GHN3_E_LIMIT): C a multiple of 4, k <= 15, pad <= k / 2. */
typedef struct ghn3_pool_desc { int32_t N, H, W, C, k, stride, pad, Ho, Wo, mode; } ghn3_pool_desc;
int ghn3_pool_fwd(const ghn3_pool_desc* desc, const float* x, float* y, unsigned char* idx, void* stream);
int ghn3_pool_bwd(const ghn3_pool_desc* desc, const float* dy, const unsigned char* idx, float* dx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GHN3_HIP_H */
