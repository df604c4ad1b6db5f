// libghn3_hip.so runtime: context, program executor (ghn3_run), GEMM problem staging, timing helpers.
// The executor is the MI355X-native replacement for the Python/ATen dispatch of GHN3.forward
// (ghn3/nn.py:247-328): one host call launches the whole forward (or backward) as a fixed kernel sequence
// on one HIP stream, with no device synchronisation and no allocation.

#include "ghn3_internal.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>

static thread_local char g_err[512] = "";

void ghn3_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ghn3_last_error(void) { return g_err; }
extern "C" int ghn3_abi_version(void) { return GHN3_ABI_VERSION; }

#define HIPCHK(expr)                                                                  \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            ghn3_set_error("%s failed: %s", #expr, hipGetErrorString(e_));            \
            return GHN3_E_HIP;                                                        \
        }                                                                             \
    } while (0)

static const int kStageSlots = 16;

struct Launch { int a_mode, b_mode, tile, first, count, tiles, with_ln, max_slice; };

// What a staging slot's table was resolved from (copies of the caller's arrays) and the launch groups that came out: a run with
// the same ops, problems and buffer pointers -- every step of a training loop once the allocator has settled -- reuses the
// table already on the device: no host resolve (hundreds of problems, ~0.3 ms at ghn3xlm16) and no 100 KB upload in front
// of the run's first kernel (the blit kernels of that copy sat in the stream: ~50 us per run).
struct SlotCache {
    bool valid = false;
    uint64_t stamp = 0;
    int n_ops = 0, n_problems = 0, n_bufs = 0;
    std::vector<ghn3_op> ops;
    std::vector<ghn3_gemm_problem> problems;
    std::vector<void*> bufs;
    std::vector<std::vector<Launch>> launches;
    hipStream_t up_stream = nullptr;   // the stream the table was uploaded on
};

struct ghn3_ctx {
    // staging slots for resolved GEMM problem tables (least recently used one is overwritten)
    SlotCache* cache;
    uint64_t clock;
    uint64_t cache_hits, cache_misses;
    GemmProbDev* h_stage[kStageSlots];
    GemmProbDev* d_stage[kStageSlots];
    hipEvent_t ev[kStageSlots];        // the slot's upload has completed (host buffer reusable)
    bool ev_used[kStageSlots];
    hipEvent_t ev_done[kStageSlots];   // the run that used the slot has been fully enqueued behind this event
    bool done_used[kStageSlots];
    hipStream_t copy;                  // problem tables are uploaded here, ahead of the stream that will read them
    size_t cap;            // problems per slot
    int ctype;             // compute type for GEMM operands
    // profiling: 0 off, 1 = every op bracketed + synchronised (diagnostic), 2 = only ops carrying
    // GHN3_OPFLAG_TIMED get an event pair from a pool, no synchronisation until ghn3_profile_read
    int profile;
    double ms[GHN3_OP_KIND_COUNT];
    int64_t launches[GHN3_OP_KIND_COUNT];
    hipEvent_t pe0, pe1;
    std::vector<hipEvent_t>* pool;     // pairs
    std::vector<int>* pool_tag;
    size_t pool_used;                  // pairs in use
    double tag_ms[256];
    int64_t tag_n[256];
    // side stream for ops flagged GHN3_OPFLAG_SIDE (work off the critical path of a program)
    hipStream_t side;
    hipEvent_t ev_fork, ev_join, ev_mark[4];
    bool side_enabled;
    bool side_pending;     // a run ended with GHN3_OP_DETACH: its side-stream work has not been joined yet
    bool mark_set[4];      // GHN3_OP_JOIN marks survive a DETACHed run: a later run of the same context may wait for them
};

static int ctx_reserve(ghn3_ctx* c, size_t n) {
    if (n <= c->cap) return GHN3_OK;
    size_t cap = std::max<size_t>(n, 1024);
    for (int i = 0; i < kStageSlots; ++i) {
        if (c->ev_used[i]) { HIPCHK(hipEventSynchronize(c->ev[i])); c->ev_used[i] = false; }
        if (c->done_used[i]) { HIPCHK(hipEventSynchronize(c->ev_done[i])); c->done_used[i] = false; }
        if (c->h_stage[i]) HIPCHK(hipHostFree(c->h_stage[i]));
        if (c->d_stage[i]) HIPCHK(hipFree(c->d_stage[i]));
        c->h_stage[i] = nullptr; c->d_stage[i] = nullptr;
        c->cache[i].valid = false;
        c->cache[i].stamp = 0;
        HIPCHK(hipHostMalloc((void**)&c->h_stage[i], cap * sizeof(GemmProbDev), hipHostMallocDefault));
        HIPCHK(hipMalloc((void**)&c->d_stage[i], cap * sizeof(GemmProbDev)));
    }
    c->cap = cap;
    return GHN3_OK;
}

extern "C" int ghn3_ctx_create(ghn3_ctx** out) {
    if (!out) return GHN3_E_ARG;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { ghn3_set_error("no HIP device"); return GHN3_E_HIP; }
    ghn3_ctx* c = new ghn3_ctx();
    memset(c, 0, sizeof(*c));
    c->pool = new std::vector<hipEvent_t>();
    c->pool_tag = new std::vector<int>();
    c->cache = new SlotCache[kStageSlots];
    for (int i = 0; i < kStageSlots; ++i) {
        HIPCHK(hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming));
    }
    HIPCHK(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&c->pe0));
    HIPCHK(hipEventCreate(&c->pe1));
    {
        // lowest priority: the side stream only fills what the dependent chain on the caller's stream leaves idle
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, least));
    }
    HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreateWithFlags(&c->ev_mark[i], hipEventDisableTiming));
    c->side_enabled = !(getenv("GHN3_NO_SIDE_STREAM") && atoi(getenv("GHN3_NO_SIDE_STREAM")) != 0);
    int rc = ghn3_gemm_init();
    if (rc) return rc;
    rc = ghn3_attn_init();
    if (rc) return rc;
    rc = ghn3_gemm_x3_init();
    if (rc) return rc;
    rc = ghn3_gemm_x3s_init();
    if (rc) return rc;
    rc = ghn3_gemm_p8_init();
    if (rc) return rc;
    rc = ctx_reserve(c, 1024);
    if (rc) return rc;
    *out = c;
    return GHN3_OK;
}

extern "C" void ghn3_ctx_destroy(ghn3_ctx* c) {
    if (!c) return;
    for (int i = 0; i < kStageSlots; ++i) {
        if (c->ev_used[i]) hipEventSynchronize(c->ev[i]);
        if (c->h_stage[i]) hipHostFree(c->h_stage[i]);
        if (c->d_stage[i]) hipFree(c->d_stage[i]);
        hipEventDestroy(c->ev[i]);
        hipEventDestroy(c->ev_done[i]);
    }
    hipStreamDestroy(c->copy);
    hipEventDestroy(c->pe0);
    hipEventDestroy(c->pe1);
    hipStreamSynchronize(c->side);
    hipStreamDestroy(c->side);
    hipEventDestroy(c->ev_fork);
    for (int i = 0; i < 4; ++i) hipEventDestroy(c->ev_mark[i]);
    hipEventDestroy(c->ev_join);
    for (hipEvent_t e : *c->pool) hipEventDestroy(e);
    delete c->pool;
    delete c->pool_tag;
    delete[] c->cache;
    delete c;
}

extern "C" int ghn3_ctx_set_compute_type(ghn3_ctx* c, int ctype) {
    if (!c) return GHN3_E_NOCTX;
    if (ctype < 0 || ctype > 2) { ghn3_set_error("bad compute type %d", ctype); return GHN3_E_ARG; }
    c->ctype = ctype;
    return GHN3_OK;
}

extern "C" int ghn3_profile_enable(ghn3_ctx* c, int mode) {
    if (!c) return GHN3_E_NOCTX;
    c->profile = mode;
    return GHN3_OK;
}
static int drain_pool(ghn3_ctx* c) {
    for (size_t k = 0; k < c->pool_used; ++k) {
        hipEvent_t e0 = (*c->pool)[2 * k], e1 = (*c->pool)[2 * k + 1];
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        const int tag = (*c->pool_tag)[k] & 255;
        c->tag_ms[tag] += ms;
        c->tag_n[tag] += 1;
    }
    c->pool_used = 0;
    return GHN3_OK;
}
extern "C" int ghn3_profile_read(ghn3_ctx* c, double* ms, int64_t* launches, int reset) {
    if (!c) return GHN3_E_NOCTX;
    for (int i = 0; i < GHN3_OP_KIND_COUNT; ++i) {
        if (ms) ms[i] = c->ms[i];
        if (launches) launches[i] = c->launches[i];
        if (reset) { c->ms[i] = 0; c->launches[i] = 0; }
    }
    return GHN3_OK;
}
extern "C" int ghn3_profile_read_tags(ghn3_ctx* c, double* ms256, int64_t* n256, int reset) {
    if (!c) return GHN3_E_NOCTX;
    int rc = drain_pool(c);
    if (rc) return rc;
    for (int i = 0; i < 256; ++i) {
        if (ms256) ms256[i] = c->tag_ms[i];
        if (n256) n256[i] = c->tag_n[i];
        if (reset) { c->tag_ms[i] = 0; c->tag_n[i] = 0; }
    }
    return GHN3_OK;
}

// ---- reference resolution -----------------------------------------------------------------------
struct Resolver {
    void* const* bufs; int n_bufs; bool bad;
    template <typename T> T* get(const ghn3_ref& r) {
        if (r.buf < 0) return nullptr;
        if (r.buf >= n_bufs || bufs[r.buf] == nullptr) { bad = true; return nullptr; }
        return reinterpret_cast<T*>(reinterpret_cast<char*>(bufs[r.buf]) + r.off);
    }
};

// tile code 32 = the latency-optimised small-problem kernel (gemm_small.hip, exact fp32): chosen when a problem
// cannot fill the chip with 64x64 tiles and its K loop is short enough for one workgroup to split four ways.
static int pick_tile(const ghn3_gemm_problem& p, int forced, int64_t op_t64) {
    // split-bf16 kernel (gemm_x3.hip): one instantiation per (tile, K slice) -> composite bucket code
    if (p.flags & GHN3_GEMM_X3) {
        // 44 / 45 = staged kernels (gemm_x3d.hip: fragment-major weights, whole-K activation rows in LDS, optional
        // LayerNorm row prologue; 32 x 48 / 16 x 32 tiles): bucket by (code, ln_kind, K)
        if (forced == 44 || forced == 45 || p.ln_kind)
            return (forced == 45 ? 7000 : 6000) + 100 * (p.ln_kind & 3) + ((p.K / 64) % 100);
        return 4000 + 10 * (((forced >= 40 && forced <= 42) ? forced : 40) - 40) + p.x3_slice / 64;
    }
    if (p.ln_kind) return 32;                       // the row prologue lives in the small-problem kernel
    if (p.flags & GHN3_GEMM_OP16) {
        // tile codes 16 / 24 = the 16-bit-operand kernel with 128 x 128 / 256 x 256 tiles.  The big tile has twice
        // the arithmetic intensity but runs one 512-thread block per CU: it needs enough tiles to fill the chip.
        if (forced == 16 || forced == 24 || forced == 20 || forced == 25 || forced == 29 || forced == 30) return forced;
        // 28 = the 8-phase kernel ((192 | 256 | 320) x 256 tiles): everything it can take that has at least ~a tile of rows;
        // the rest of the op (small families, atomically split problems) stays on the 128 x 128 kernel
        if (forced == 28) {
            const bool kmap_ok = p.b_kq == 0 || p.b_kq % 64 == 0 || 64 % p.b_kq == 0;
            return (p.ksplit <= 1 && p.M >= 160 && (p.N & 3) == 0 && kmap_ok && p.a_gather.buf < 0 && p.b_gather.buf < 0 &&
                    p.c_gather.buf < 0) ? 28 : 16;
        }
        const int64_t t256 = (int64_t)((p.M + 255) / 256) * ((p.N + 255) / 256) * (p.ksplit > 1 ? p.ksplit : 1);
        const double eff = ((double)p.M / (((p.M + 255) / 256) * 256.0)) * ((double)p.N / (((p.N + 255) / 256) * 256.0));
        return (t256 >= 200 && eff >= 0.8) ? 24 : 16;
    }
    // 48 = the split-bf16 weight-gradient kernel (gemm_wg.hip): fp32 activations, reduction over rows (COL / COL); problems
    // of the op it cannot take run on the 64 x 64 fp32 tiles
    if (forced == 48 || forced == 49) {
        const bool ok = p.a_mode == GHN3_MODE_COL && p.b_mode == GHN3_MODE_COL && p.ksplit <= 1 && (p.M & 3) == 0 &&
                        (p.N & 3) == 0 && (p.ldc & 3) == 0 && (p.C.off & 15) == 0 && (p.lda & 3) == 0 && (p.ldb & 3) == 0 &&
                        (p.A.off & 15) == 0 && (p.B.off & 15) == 0 && p.a_gather.buf < 0 &&
                        p.b_gather.buf < 0 && p.c_gather.buf < 0 && !p.a_q && !p.b_q && !p.c_q && !p.bias_q &&
                        p.act == GHN3_ACT_NONE && p.dact == GHN3_DACT_NONE && p.residual.buf < 0 && p.aux_out.buf < 0 &&
                        (p.bias.buf < 0 || (p.flags & GHN3_GEMM_BIASGRAD));
        return ok ? forced : 64;
    }
    if (forced == 32 || forced == 64 || forced == 128) return forced;
    // op_t64 = 64x64 tiles of ALL problems launched together with this one: a grouped launch that already fills
    // the chip keeps the tiled kernels (better operand reuse); a lone small problem takes the latency kernel.
    const double flops = 2.0 * p.M * p.N * (double)p.K;
    if (p.ksplit <= 1 && op_t64 <= 640 && flops <= 4e9 && p.K <= 8192) return 32;
    const int64_t t128 = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128) * (p.ksplit > 1 ? p.ksplit : 1);
    if (p.M >= 96 && p.N >= 96 && t128 >= 192) return 128;
    return 64;
}

static inline bool is16(int tl) { return tl == 16 || tl == 24 || tl == 20 || tl == 25 || tl == 28 || tl == 29 || tl == 30; }   // 16-bit-operand kernels

extern "C" int ghn3_run(ghn3_ctx* c, const ghn3_op* ops, int n_ops, const ghn3_gemm_problem* problems, int n_problems,
                        void* const* bufs, int n_bufs, void* stream_) {
    if (!c) return GHN3_E_NOCTX;
    if (n_ops < 0 || (n_ops > 0 && !ops) || n_bufs < 0) { ghn3_set_error("ghn3_run: bad arguments"); return GHN3_E_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    Resolver R{bufs, n_bufs, false};

    // ---- 1. resolve every GEMM op into launch groups inside one staging slot --------------------
    std::vector<std::vector<Launch>> fresh_launches;
    const std::vector<std::vector<Launch>>* launches = &fresh_launches;
    size_t need = 0;
    for (int k = 0; k < n_ops; ++k)
        if (ops[k].kind == GHN3_OP_GEMM) need += (size_t)ops[k].i[1];
    GemmProbDev* hs = nullptr; GemmProbDev* ds = nullptr;
    int used_slot = -1;
    static const bool use_copy_stream = getenv("GHN3_COPY_STREAM") && atoi(getenv("GHN3_COPY_STREAM")) != 0;
    static const bool use_cache = !(getenv("GHN3_RUN_CACHE") && atoi(getenv("GHN3_RUN_CACHE")) == 0);
    int hit = -1;
    if (need > 0) {
        int rc = ctx_reserve(c, need);
        if (rc) return rc;
        if (n_problems < 0 || (n_problems > 0 && !problems)) { ghn3_set_error("ghn3_run: bad problem table"); return GHN3_E_ARG; }
        for (int i = 0; use_cache && i < kStageSlots && hit < 0; ++i) {
            const SlotCache& e = c->cache[i];
            if (e.valid && e.n_ops == n_ops && e.n_problems == n_problems && e.n_bufs == n_bufs &&
                memcmp(e.bufs.data(), bufs, (size_t)n_bufs * sizeof(void*)) == 0 &&
                memcmp(e.ops.data(), ops, (size_t)n_ops * sizeof(ghn3_op)) == 0 &&
                memcmp(e.problems.data(), problems, (size_t)n_problems * sizeof(ghn3_gemm_problem)) == 0)
                hit = i;
        }
    }
    if (hit >= 0) {
        // the same run as one whose table is still on the device
        SlotCache& e = c->cache[hit];
        e.stamp = ++c->clock;
        c->cache_hits++;
        hs = c->h_stage[hit]; ds = c->d_stage[hit];
        launches = &e.launches;
        if (e.up_stream != stream && c->ev_used[hit]) HIPCHK(hipStreamWaitEvent(stream, c->ev[hit], 0));
        if (use_copy_stream) used_slot = hit;
    } else if (need > 0) {
        int slot = 0;                                    // the least recently used slot (never used ones first)
        for (int i = 1; i < kStageSlots; ++i)
            if (c->cache[i].stamp < c->cache[slot].stamp) slot = i;
        c->cache[slot].valid = false;
        c->cache_misses++;
        if (c->ev_used[slot]) { HIPCHK(hipEventSynchronize(c->ev[slot])); c->ev_used[slot] = false; }
        hs = c->h_stage[slot]; ds = c->d_stage[slot];
        std::vector<std::vector<Launch>>& op_launches = fresh_launches;
        op_launches.resize(n_ops);
        size_t pos = 0;
        for (int k = 0; k < n_ops; ++k) {
            if (ops[k].kind != GHN3_OP_GEMM) continue;
            const int first = (int)ops[k].i[0], cnt = (int)ops[k].i[1], forced = (int)ops[k].i[2];
            if (first < 0 || cnt < 0 || first + cnt > n_problems) {
                ghn3_set_error("op %d: GEMM problem range [%d,%d) outside table of %d", k, first, first + cnt, n_problems);
                return GHN3_E_ARG;
            }
            int64_t op_t64 = 0;
            for (int q = first; q < first + cnt; ++q)
                op_t64 += (int64_t)((problems[q].M + 63) / 64) * ((problems[q].N + 63) / 64);
            // bucket by (a_mode, b_mode, tile): the tile codes that occur among the op's problems, in ascending order
            std::vector<int> codes;
            for (int q = first; q < first + cnt; ++q) {
                if (problems[q].M <= 0 || problems[q].N <= 0) continue;
                const int tc = pick_tile(problems[q], forced, op_t64);
                if (std::find(codes.begin(), codes.end(), tc) == codes.end()) codes.push_back(tc);
            }
            std::sort(codes.begin(), codes.end());
            for (int am = 0; am < 2; ++am)
                for (int bm = 0; bm < 2; ++bm)
                    for (int tl : codes) {
                        Launch L{am, bm, tl, (int)pos, 0, 0, 0, 0};
                        int te = tl == 16 ? 128 : (tl == 24 || tl == 20 || tl == 25 || tl == 28 || tl == 29 || tl == 30) ? 256 : tl == 48 ? 64 : tl == 49 ? 128 : tl;   // tile edge (rows)
                        int te_n = (tl == 20 || tl == 30) ? 128 : te;                                        // (columns)
                        const bool x3 = tl >= 4000;
                        const bool x3old = x3 && tl < 5000;
                        if (x3old && !ghn3_gemm_x3_tile(40 + (tl - 4000) / 10, 64 * (tl % 10), &te, &te_n)) {
                            ghn3_set_error("op %d: no split-bf16 kernel for tile bucket %d (tile code %d, K slice %d)", k, tl,
                                           40 + (tl - 4000) / 10, 64 * (tl % 10));
                            return GHN3_E_LIMIT;
                        }
                        const int x3s_code = tl >= 7000 ? 45 : 44, x3s_ln = (tl / 100) % 10, x3s_k = 64 * (tl % 100);
                        if (tl >= 6000 && !ghn3_gemm_x3s_tile(x3s_code, x3s_k, x3s_ln, &te, &te_n)) {
                            ghn3_set_error("op %d: no staged split-bf16 kernel for tile code %d, K = %d, ln_kind %d "
                                           "(gemm_x3d.hip: K = C <= 384 with a prologue; 16 x 32 tiles also K = 3C / 4C)", k,
                                           x3s_code, x3s_k, x3s_ln);
                            return GHN3_E_LIMIT;
                        }
                        // members of this launch; XCD-pinned problems (16-bit-operand kernel, tile codes 16 / 20) first, by XCD
                        std::vector<int> members, pinned_m;
                        for (int q = first; q < first + cnt; ++q) {
                            const ghn3_gemm_problem& p = problems[q];
                            if (p.M <= 0 || p.N <= 0) continue;
                            if (p.a_mode != am || p.b_mode != bm || pick_tile(p, forced, op_t64) != tl) continue;
                            members.push_back(q);
                        }
                        const bool pin_ok = (tl == 16 || tl == 20 || tl == 28) && members.size() >= 8;
                        if (pin_ok) {
                            for (int x = 0; x < 8; ++x)
                                for (int q : members)
                                    if (problems[q].xcd_pin == x + 1 && problems[q].ksplit <= 1) pinned_m.push_back(q);
                            if (!pinned_m.empty()) {
                                std::vector<int> rest;
                                for (int q : members)
                                    if (!(problems[q].xcd_pin >= 1 && problems[q].xcd_pin <= 8 && problems[q].ksplit <= 1))
                                        rest.push_back(q);
                                members = pinned_m;
                                members.insert(members.end(), rest.begin(), rest.end());
                            }
                        }
                        const int n_pinned = (int)pinned_m.size();
                        int pin_local[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pin_first[8] = {0, 0, 0, 0, 0, 0, 0, 0},
                            pin_count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        int pin_end = 0;
                        if (n_pinned) {                          // local tile counts per XCD -> first unpinned id
                            for (int q : pinned_m) {
                                const ghn3_gemm_problem& p = problems[q];
                                const int tm_ = (tl == 28 && p.mtiles.buf >= 0) ? p.n_mtiles : (p.M + te - 1) / te;
                                pin_local[p.xcd_pin - 1] += tm_ * ((p.N + te_n - 1) / te_n);
                            }
                            for (int x = 0; x < 8; ++x) pin_end = std::max(pin_end, 8 * pin_local[x]);
                            for (int x = 0; x < 8; ++x) pin_local[x] = 0;
                            L.tiles = pin_end;
                        }
                        int member_k = 0;
                        for (int q : members) {
                            const ghn3_gemm_problem& p = problems[q];
                            const int member_i = member_k++;
                            const int pin_x = member_i < n_pinned ? p.xcd_pin - 1 : -1;
                            if (is16(tl) && (p.a_mode != GHN3_MODE_ROW || p.b_mode != GHN3_MODE_ROW || (p.lda & 7) ||
                                             (p.ldb & 7) || (p.b_kq & 7) || (p.flags & GHN3_GEMM_BIASGRAD) ||
                                             p.K >= (1 << 24) || (p.ldc & 3) || (p.C.off & 15) ||
                                             p.act == GHN3_ACT_GELU || p.dact == GHN3_DACT_GELU ||
                                             (p.aux_in.off & 15) || (p.aux_out.off & 15) || (p.residual.off & 15))) {
                                ghn3_set_error("op %d problem %d: 16-bit operands need ROW/ROW modes, ld %% 8 == 0, "
                                               "b_kq %% 8 == 0, K < 2^24, no BIASGRAD / GELU, ldc %% 4 == 0 and "
                                               "16-byte aligned C / aux / residual", k, q);
                                return GHN3_E_ARG;
                            }
                            if ((tl == 25 || tl == 29 || tl == 30) && (p.bias.buf >= 0 || p.residual.buf >= 0 || p.aux_in.buf >= 0 ||
                                             (p.aux_out.buf >= 0 && !((tl == 29 || tl == 30) && (p.flags & GHN3_GEMM_SUMSQ))) ||
                                             p.act != GHN3_ACT_NONE || p.dact != GHN3_DACT_NONE || p.a_gather.buf >= 0 ||
                                             p.b_gather.buf >= 0 || p.c_gather.buf >= 0 || p.a_q || p.b_q || p.b_kq ||
                                             p.lim.buf >= 0 || p.ksplit > 1 || (p.flags & GHN3_GEMM_ACCUM))) {
                                ghn3_set_error("op %d problem %d: tile 25 / 29 / 30 (persistent output-heavy kernels) take plain "
                                               "problems only: C = alpha A B^T with an optional row map of C", k, q);
                                return GHN3_E_ARG;
                            }
                            if ((tl == 29 || tl == 30) && ((p.N & 3) || p.K < 1 || (int64_t)p.M * p.lda * 2 >= (int64_t)0xfff00000 ||
                                             (int64_t)p.N * p.ldb * 2 >= (int64_t)0xfff00000)) {
                                ghn3_set_error("op %d problem %d: tile 29 needs N %% 4 == 0, K >= 1 and operands below 4 GB", k, q);
                                return GHN3_E_ARG;
                            }
                            // (the 8-phase kernels take the quotient of the row map of C from one 32-bit multiply-high: exact while M q < 2^32)
                            if ((tl == 28 || tl == 29 || tl == 30) && p.c_q > 0 && (int64_t)p.M * p.c_q >= ((int64_t)1 << 32)) {
                                ghn3_set_error("op %d problem %d: tiles 28 / 29 need rows x row-map period below 2^32", k, q);
                                return GHN3_E_ARG;
                            }
                            if (tl == 28) {
                                // 32-bit byte offsets inside the kernel: both operands must span less than 4 GB
                                const int64_t a_rows = p.a_q > 0 ? ((int64_t)(p.M - 1) / p.a_q) * p.a_s + p.a_q : p.M;
                                const int64_t b_rows = p.b_q > 0 ? ((int64_t)(p.N - 1) / p.b_q) * p.b_s + p.b_q : p.N;
                                const int64_t k_phys = p.b_kq > 0 ? ((int64_t)(p.K + 63) / p.b_kq + 1) * p.b_ks : p.K + 64;
                                if (p.ksplit > 1 || p.a_gather.buf >= 0 || p.b_gather.buf >= 0 || p.c_gather.buf >= 0 ||
                                    a_rows * p.lda * 2 >= (int64_t)0xfff00000 ||
                                    (b_rows * p.ldb + k_phys) * 2 >= (int64_t)0xfff00000 ||
                                    (p.mtiles.buf >= 0 && p.n_mtiles <= 0)) {
                                    ghn3_set_error("op %d problem %d: tile 28 (8-phase kernel) takes no split-K / gathers and "
                                                   "operands below 4 GB", k, q);
                                    return GHN3_E_ARG;
                                }
                            }
                            if (x3 && (p.a_mode != GHN3_MODE_ROW || p.b_mode != GHN3_MODE_ROW ||
                                       p.b_gather.buf >= 0 || p.a_q || p.b_q || p.c_q || p.bias_q ||
                                       (p.N & 3) || (p.ldc & 3) || (p.ldb & 7) ||
                                       (x3old && (p.x3_slice <= 0 || (p.x3_slice & 63) || (p.K % p.x3_slice))) ||
                                       (!x3old && ((p.K & 63) || (p.N & 15) || (p.ln_kind && p.a_gather.buf >= 0))) || p.ksplit > 1 || p.B2.buf < 0 || (p.C.off & 15) ||
                                       (p.bias.off & 15) || (p.aux_in.off & 15) || (p.aux_out.off & 15) ||
                                       (p.residual.off & 15) || (p.flags & (GHN3_GEMM_ACCUM | GHN3_GEMM_BIASGRAD)) ||
                                       (p.bias.buf >= 0 && p.bias_stride > 1))) {
                                ghn3_set_error("op %d problem %d: split-bf16 (X3) problems need ROW/ROW modes without "
                                               "a B gather / maps, N %% 4 == 0, ldc %% 4 == 0, ldb %% 8 == 0, a K slice that is "
                                               "a multiple of 64 and divides K, the lo copy B2 and 16-byte aligned C / "
                                               "bias / aux / residual", k, q);
                                return GHN3_E_ARG;
                            }
                            if (p.M >= (1 << 24) || p.N >= (1 << 24) || p.K >= (1 << 24)) {
                                ghn3_set_error("op %d problem %d: M, N, K must be below 2^24", k, q);
                                return GHN3_E_LIMIT;
                            }
                            if (p.K < 0 || (p.lda & 3) || (p.ldb & 3) || (p.A.off & 15) || (p.B.off & 15)) {
                                ghn3_set_error("op %d problem %d: operands must be 16-byte aligned with ld %% 4 == 0 "
                                               "(lda=%d ldb=%d)", k, q, p.lda, p.ldb);
                                return GHN3_E_ARG;
                            }
                            GemmProbDev& g = hs[pos++];
                            g.A = R.get<const float>(p.A); g.B = R.get<const float>(p.B); g.C = R.get<float>(p.C);
                            g.bias = R.get<const float>(p.bias); g.residual = R.get<const float>(p.residual);
                            g.aux_in = R.get<const float>(p.aux_in); g.aux_out = R.get<float>(p.aux_out);
                            g.a_gather = R.get<const int>(p.a_gather); g.b_gather = R.get<const int>(p.b_gather);
                            g.c_gather = R.get<const int>(p.c_gather);
                            g.M = p.M; g.N = p.N; g.K = p.K; g.lda = p.lda; g.ldb = p.ldb; g.ldc = p.ldc;
                            g.a_q = p.a_q; g.a_s = p.a_s; g.b_q = p.b_q; g.b_s = p.b_s; g.c_q = p.c_q; g.c_s = p.c_s;
                            g.bias_q = p.bias_q; g.bias_s = p.bias_s; g.bias_stride = p.bias_stride ? p.bias_stride : 1;
                            g.act = p.act; g.dact = p.dact; g.flags = p.flags; g.alpha = p.alpha;
                            if ((p.flags & GHN3_GEMM_BIASGRAD) && (p.a_mode != GHN3_MODE_COL || !g.bias)) {
                                ghn3_set_error("op %d problem %d: BIASGRAD needs a COL-mode A and a bias ref", k, q);
                                return GHN3_E_ARG;
                            }
                            if (!g.A || !g.B || !g.C || (p.dact != GHN3_DACT_NONE && !g.aux_in)) {
                                ghn3_set_error("op %d problem %d: missing A/B/C/aux_in buffer", k, q);
                                return GHN3_E_ARG;
                            }
                            g.tile_start = L.tiles;               // always a multiple of 8 (XCD-aware order)
                            g.tiles_m = (p.M + te - 1) / te;
                            g.tiles_n = (p.N + te_n - 1) / te_n;
                            g.mtab = nullptr;
                            if (tl == 28 && p.mtiles.buf >= 0) {
                                g.mtab = R.get<const int>(p.mtiles);
                                g.tiles_m = p.n_mtiles;
                            }
                            g.pin = pin_x + 1; g.pin_first = g.pin_count = 0;
                            g.pin_end = pin_end; g.pin_total = n_pinned;
                            if (pin_x >= 0) {
                                g.tile_start = pin_local[pin_x];  // first LOCAL index on its XCD
                                pin_local[pin_x] += g.tiles_m * g.tiles_n;
                                if (pin_count[pin_x]++ == 0) pin_first[pin_x] = member_i;
                            }
                            g.kq = is16(tl) ? p.b_kq : 0; g.ks = p.b_ks;
                            g.lim = is16(tl) ? R.get<const int>(p.lim) : nullptr;
                            g.lim_kind = (g.lim || (tl == 28 && p.mtiles.buf >= 0)) ? p.lim_kind : 0;
                            g.alpha_amax = is16(tl) ? R.get<const float>(p.alpha_amax) : nullptr;
                            g.B2 = x3 ? R.get<const void>(p.B2) : nullptr;
                            if (x3old && p.x3_slice > L.max_slice) L.max_slice = p.x3_slice;
                            if (x3 && !x3old) L.max_slice = p.K;     // (one K per bucket: the bucket code carries it)
                            g.ln_kind = p.ln_kind; g.ln_eps = p.ln_eps;
                            if (p.ln_kind) L.with_ln = 1;
                            for (int e = 0; e < 6; ++e) g.ln_p[e] = p.ln_kind ? R.get<const float>(p.ln_p[e]) : nullptr;
                            if (p.ln_kind) {
                                const bool ok = (p.ln_kind == 1 || p.ln_kind == 2) && p.a_mode == GHN3_MODE_ROW &&
                                                p.a_gather.buf < 0 && p.a_q == 0 && p.K <= 4096 && (p.K & 3) == 0 &&
                                                !(p.flags & GHN3_GEMM_OP16) && g.ln_p[0] &&
                                                (p.ln_kind == 1 ? g.ln_p[1] != nullptr
                                                                : (g.ln_p[1] && g.ln_p[2] && g.ln_p[3]));
                                if (!ok) {
                                    ghn3_set_error("op %d problem %d: the LayerNorm row prologue needs a ROW-mode fp32 A "
                                                   "without gather, K %% 4 == 0, K <= 4096 and its parameter refs", k, q);
                                    return GHN3_E_ARG;
                                }
                            }
                            g.order = (int64_t)p.M > (int64_t)p.N ? 1 : 0;   // stream the larger operand once
                            if (tl == 28) g.order = 0;                       // (the 8-phase kernel has one tile order)
                            g.ksplit = p.ksplit > 1 ? p.ksplit : 1;
                            g.k_chunk = x3old ? p.x3_slice : x3 ? p.K : ((p.K + g.ksplit - 1) / g.ksplit + 63) / 64 * 64;
                            if (g.ksplit > 1 && (p.bias.buf >= 0 || p.act || p.dact || p.residual.buf >= 0 ||
                                                 p.aux_out.buf >= 0)) {
                                ghn3_set_error("op %d problem %d: split-K allows no epilogue but alpha", k, q);
                                return GHN3_E_ARG;
                            }
                            g.xcd_cols = 0;
                            if (pin_x >= 0) {
                                // (ids were reserved above: [0, pin_end))
                            } else if (tl == 25 || tl == 29 || tl == 30) {
                                // XCD-blocked order of the persistent kernel: the 8 XCDs form a (8 / G) x G grid; an XCD
                                // works on every (8 / G)-th row tile and on one of G column groups, chosen so that its
                                // share of B stays in its 4 MB L2 while the A tiles stream through once per column group
                                static const int64_t b_budget = getenv("GHN3_XCD_B_BYTES") ? atoll(getenv("GHN3_XCD_B_BYTES"))
                                                                                          : (int64_t)(5 << 19);     // (2.5 MB: two column groups for the W2 weight gradient; r03: 1.07 ms against 1.13 ms with four, same traffic)
                                int G = 1;
                                while (G < 8 && (int64_t)p.N * p.K * 2 / G > b_budget && g.tiles_n >= 2 * G) G *= 2;
                                g.xcd_cols = G;
                                const int npg = (g.tiles_n + G - 1) / G, mpc = (g.tiles_m + 8 / G - 1) / (8 / G);
                                L.tiles += 8 * npg * mpc;
                            } else if (tl == 32 || tl == 48 || tl == 49 || x3) {
                                L.tiles += g.tiles_m * g.tiles_n;          // plain order, no padding
                            } else {
                                const int per_split = g.order ? ((g.tiles_m + 7) / 8 * 8) * g.tiles_n
                                                              : g.tiles_m * ((g.tiles_n + 7) / 8 * 8);
                                L.tiles += per_split * g.ksplit;
                            }
                            L.count++;
                        }
                        if (n_pinned)                            // directory of XCD x in entry x of the launch's array
                            for (int x = 0; x < 8; ++x) {
                                hs[L.first + x].pin_first = pin_first[x];
                                hs[L.first + x].pin_count = pin_count[x];
                            }
                        if (L.count > 0) op_launches[k].push_back(L);
                    }
        }
        if (R.bad) { ghn3_set_error("ghn3_run: a GEMM problem references an absent buffer"); return GHN3_E_ARG; }
        // Upload on the copy stream: stream-ordered behind `stream` the ~100 KB table sat on the critical path of every run
        // (in 4 KB pieces: ~170 us in front of the backward program).  The copy stream only waits for the kernels of the
        // run that used this slot last (kStageSlots runs ago); `stream` waits for the copy.
        // (GHN3_COPY_STREAM=1: upload on a separate copy stream ahead of `stream`; measured 9.11 vs 9.02 ms per step --
        // the host runs far enough ahead that the in-stream copy is never waited for -- so off by default)
        if (use_copy_stream) {
            if (c->done_used[slot]) HIPCHK(hipStreamWaitEvent(c->copy, c->ev_done[slot], 0));
            HIPCHK(hipMemcpyAsync(ds, hs, pos * sizeof(GemmProbDev), hipMemcpyHostToDevice, c->copy));
            HIPCHK(hipEventRecord(c->ev[slot], c->copy));
            c->ev_used[slot] = true;
            HIPCHK(hipStreamWaitEvent(stream, c->ev[slot], 0));
            used_slot = slot;
        } else {
            HIPCHK(hipMemcpyAsync(ds, hs, pos * sizeof(GemmProbDev), hipMemcpyHostToDevice, stream));
            HIPCHK(hipEventRecord(c->ev[slot], stream));
            c->ev_used[slot] = true;
        }
        if (use_cache) {
            SlotCache& e = c->cache[slot];
            e.n_ops = n_ops; e.n_problems = n_problems; e.n_bufs = n_bufs;
            e.ops.assign(ops, ops + n_ops);
            e.problems.assign(problems, problems + n_problems);
            e.bufs.assign(bufs, bufs + n_bufs);
            e.launches.swap(fresh_launches);
            launches = &e.launches;
            e.up_stream = use_copy_stream ? c->copy : stream;
            e.stamp = ++c->clock;
            e.valid = true;
        } else {
            c->cache[slot].stamp = ++c->clock;           // (plain least-recently-used rotation of the slots)
        }
    }

    // ---- 2. launch ops in order ------------------------------------------------------------------
    // Two-stream execution.  Ops flagged GHN3_OPFLAG_SIDE run on the context's side stream: before such an op the
    // side stream waits for everything the main stream has been given so far (program order = dependency order);
    // the main stream waits for the side stream at GHN3_OP_JOIN and at the end of the run.
    hipStream_t const main_stream = stream;
    // A run without side-stream ops or joins (e.g. the local passes of the gradient exchange, issued from the
    // communication stream between two parts of a backward) leaves the pending side work of a DETACHed run alone: it is
    // still joined by the next run that does use the side stream, or observed with ghn3_ctx_side_wait.
    bool touches_side = false;
    for (int k = 0; k < n_ops && !touches_side; ++k)
        touches_side = ((ops[k].flags & GHN3_OPFLAG_SIDE) && c->side_enabled) || ops[k].kind == GHN3_OP_JOIN ||
                       ops[k].kind == GHN3_OP_DETACH;
    bool main_dirty = true, side_dirty = touches_side && c->side_pending;
    bool detach = false;
    bool* const mark_set = c->mark_set;                  // (kept across runs: a MARK in one part, its WAIT in a later one)
    if (touches_side) c->side_pending = false;
    auto join = [&]() -> int {
        if (side_dirty) {
            HIPCHK(hipEventRecord(c->ev_join, c->side));
            HIPCHK(hipStreamWaitEvent(main_stream, c->ev_join, 0));
            side_dirty = false;
        }
        for (int i = 0; i < 4; ++i) mark_set[i] = false;    // everything marked so far has been waited for
        return GHN3_OK;
    };
    for (int k = 0; k < n_ops; ++k) {
        const ghn3_op& o = ops[k];
        int rc = GHN3_OK;
        R.bad = false;
        const bool on_side = (o.flags & GHN3_OPFLAG_SIDE) && c->side_enabled && c->profile != 1 && c->profile != 3;
        // A grid cap of a side-stream GEMM only exists to leave CUs to the chain it runs beside: serialised (no side stream,
        // profile modes 1 / 3) the op takes the whole chip -- what mode 3 and the PMC passes measure is the kernel's own rate.
        const int gemm_cap = ((o.flags & GHN3_OPFLAG_SIDE) && !on_side) ? 0 : (int)o.i[3];
        if (o.kind == GHN3_OP_JOIN) {
            const int mode = (int)o.i[0], id = (int)o.i[1] & 3;
            if (mode == 1) {                             // mark: the side-stream work issued so far
                if (side_dirty) { HIPCHK(hipEventRecord(c->ev_mark[id], c->side)); mark_set[id] = true; }
            } else if (mode == 2) {                      // the run's stream waits for that mark only (later side work keeps running)
                if (mark_set[id]) { HIPCHK(hipStreamWaitEvent(main_stream, c->ev_mark[id], 0)); mark_set[id] = false; }
            } else {
                rc = join();
                if (rc) return rc;
            }
            continue;
        }
        if (o.kind == GHN3_OP_DETACH) { detach = true; continue; }
        if (o.kind == GHN3_OP_NOP) continue;             // (padding of a fixed-size op list: a DETACH in front of it still ends the run)
        detach = false;
        if (on_side) {
            if (main_dirty) {
                HIPCHK(hipEventRecord(c->ev_fork, main_stream));
                HIPCHK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
                main_dirty = false;
            }
            side_dirty = true;
        } else {
            main_dirty = true;
        }
        hipStream_t stream = on_side ? c->side : main_stream;
        const bool timed = (c->profile == 2 || c->profile == 3) && (o.flags & GHN3_OPFLAG_TIMED);
        if (c->profile == 1) HIPCHK(hipEventRecord(c->pe0, stream));
        if (timed) {
            if (c->pool_used >= 8192) { int rc2 = drain_pool(c); if (rc2) return rc2; }
            if (c->pool->size() < 2 * (c->pool_used + 1)) {
                hipEvent_t e0, e1;
                HIPCHK(hipEventCreate(&e0));
                HIPCHK(hipEventCreate(&e1));
                c->pool->push_back(e0); c->pool->push_back(e1);
                c->pool_tag->push_back(0);
            }
            (*c->pool_tag)[c->pool_used] = (o.flags >> 16) & 255;
            HIPCHK(hipEventRecord((*c->pool)[2 * c->pool_used], stream));
        }
        switch (o.kind) {
        case GHN3_OP_NOP: break;
        case GHN3_OP_GEMM:
            for (const Launch& L : (*launches)[k]) {
                if (L.tile >= 6000)
                    rc = ghn3_gemm_x3s_launch(ds + L.first, hs + L.first, L.count, L.tiles, L.tile >= 7000 ? 45 : 44, L.max_slice,
                                              (L.tile / 100) % 10, stream);
                else if (L.tile >= 4000)
                    rc = ghn3_gemm_x3_launch(ds + L.first, L.count, L.tiles, 40 + (L.tile - 4000) / 10,
                                             64 * (L.tile % 10), stream);
                else if (L.tile == 48 || L.tile == 49)
                    rc = ghn3_gemm_wg_launch(ds + L.first, L.count, L.tiles, gemm_cap, L.tile == 49 ? 128 : 64, stream);
                else if (L.tile == 32)
                    rc = ghn3_gemm_small_launch(ds + L.first, L.count, L.tiles, L.a_mode, L.b_mode, L.with_ln, stream);
                else if (L.tile == 30)
                    rc = ghn3_gemm_p8d_launch(ds + L.first, L.count, L.tiles,
                                              (o.flags & 0xff) ? ((o.flags & 0xff) - 1) : c->ctype, gemm_cap, stream);
                else if (L.tile == 29)
                    rc = ghn3_gemm_p8w_launch(ds + L.first, L.count, L.tiles,
                                              (o.flags & 0xff) ? ((o.flags & 0xff) - 1) : c->ctype, gemm_cap, stream);
                else if (L.tile == 28)
                    rc = ghn3_gemm_p8_launch(ds + L.first, L.count, L.tiles,
                                             (o.flags & 0xff) ? ((o.flags & 0xff) - 1) : c->ctype, gemm_cap, stream);
                else if (is16(L.tile))
                    rc = ghn3_gemm_h16d_launch(ds + L.first, L.count, L.tiles,
                                               L.tile == 16 ? 128 : L.tile == 20 ? 20 : L.tile == 25 ? 25 : 256,
                                               (o.flags & 0xff) ? ((o.flags & 0xff) - 1) : c->ctype, gemm_cap,
                                               stream);
                else
                    rc = ghn3_gemm_launch(ds + L.first, L.count, L.tiles, L.a_mode, L.b_mode, L.tile,
                                          (o.flags & 0xff) ? ((o.flags & 0xff) - 1) : c->ctype, stream);
                if (rc) break;
            }
            break;
        case GHN3_OP_GRAPH_PROLOGUE:
            rc = ghn3_graph_prologue(R.get<const int64_t>(o.r[0]), R.get<int>(o.r[1]), R.get<int>(o.r[2]),
                                     R.get<int>(o.r[3]), R.get<int>(o.r[4]), (int)o.i[0], (int)o.i[1], (int)o.i[2],
                                     stream);
            break;
        case GHN3_OP_EMBED_NODES:
            rc = ghn3_embed_nodes(R.get<float>(o.r[0]), R.get<const int>(o.r[1]), R.get<const int>(o.r[2]),
                                  R.get<const int>(o.r[3]), R.get<const int>(o.r[4]), R.get<const float>(o.r[5]),
                                  R.get<const float>(o.r[6]), R.get<const float>(o.r[7]), R.get<const float>(o.r[8]),
                                  R.get<const float>(o.r[9]), R.get<const float>(o.r[10]), R.get<const int>(o.r[11]),
                                  R.get<const int>(o.r[12]), R.get<const int>(o.r[13]), (int)o.i[0], (int)o.i[1],
                                  (int)o.i[2], stream);
            break;
        case GHN3_OP_EMBED_BWD:
            rc = ghn3_embed_bwd(R.get<const float>(o.r[0]), R.get<const int>(o.r[1]), R.get<const int>(o.r[2]),
                                R.get<const int>(o.r[3]), R.get<const int>(o.r[4]), R.get<float>(o.r[5]),
                                R.get<float>(o.r[6]), R.get<float>(o.r[7]), R.get<float>(o.r[8]), R.get<float>(o.r[9]),
                                R.get<float>(o.r[10]), R.get<const int>(o.r[11]), R.get<const int>(o.r[12]),
                                R.get<const int>(o.r[13]), (int)o.i[0], (int)o.i[1], (int)o.i[2], (int)o.i[3], (int)o.i[4],
                                (int)o.i[5], stream);
            break;
        case GHN3_OP_EDGE_HIDDEN:
            rc = ghn3_edge_hidden(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), R.get<const float>(o.r[2]),
                                  (int)o.i[0], (int)o.i[1], stream);
            break;
        case GHN3_OP_EDGE_HIDDEN_BWD:
            rc = ghn3_edge_hidden_bwd(R.get<float>(o.r[0]), R.get<float>(o.r[1]), R.get<float>(o.r[2]),
                                      R.get<const float>(o.r[3]), (int)o.i[0], (int)o.i[1], stream);
            break;
        case GHN3_OP_BIAS_GATHER:
            rc = ghn3_bias_gather(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), R.get<const int>(o.r[2]),
                                  (int)o.i[0], (int)o.i[1], (int)o.i[2], stream);
            break;
        case GHN3_OP_BIAS_HIST:
            rc = ghn3_bias_hist(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), R.get<const int>(o.r[2]),
                                (int)o.i[0], (int)o.i[1], (int)o.i[2], (int)o.i[3], R.get<void>(o.r[3]), (int)o.i[4], stream);
            break;
        case GHN3_OP_ROWSET_COLSUM:
            rc = ghn3_rowset_colsum(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), R.get<const void>(o.r[2]),
                                    (int)o.i[0], (int)o.i[1], (int)o.i[2], stream);
            break;
        case GHN3_OP_LAYERNORM_FWD:
            rc = ghn3_layernorm_fwd(R.get<float>(o.r[0]), R.get<float>(o.r[1]), R.get<const float>(o.r[2]),
                                    R.get<const float>(o.r[3]), R.get<float>(o.r[4]), R.get<float>(o.r[5]),
                                    R.get<const float>(o.r[6]), (int)o.i[2], (int64_t)o.i[3], (int)o.i[0], (int)o.i[1],
                                    o.f[0], stream);
            break;
        case GHN3_OP_LAYERNORM_BWD:
            rc = ghn3_layernorm_bwd(R.get<float>(o.r[0]), R.get<float>(o.r[1]), R.get<const float>(o.r[2]),
                                    R.get<const float>(o.r[3]), R.get<const float>(o.r[4]), R.get<const float>(o.r[5]),
                                    R.get<const float>(o.r[6]), R.get<const float>(o.r[7]), (int)o.i[2], (int64_t)o.i[3],
                                    (int)o.i[0], (int)o.i[1], stream);
            break;
        case GHN3_OP_LN_PARAM_GRAD:
            rc = ghn3_ln_param_grad(R.get<float>(o.r[0]), R.get<float>(o.r[1]), R.get<const float>(o.r[2]),
                                    R.get<const float>(o.r[3]), R.get<const float>(o.r[4]), R.get<const float>(o.r[5]),
                                    (int)o.i[0], (int)o.i[1], (int)o.i[2], stream);
            break;
        case GHN3_OP_LN_PARAM_GRAD_BATCH:
            rc = ghn3_ln_param_grad_batch(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), R.get<const int64_t>(o.r[2]),
                                          (int)o.i[0], (int)o.i[1], (int)o.i[2], stream);
            break;
        case GHN3_OP_ATTN_FWD:
            rc = ghn3_attn_fwd(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), R.get<const float>(o.r[2]),
                               R.get<float>(o.r[3]), R.get<const int>(o.r[4]), (int)o.i[0], (int)o.i[1], (int)o.i[2],
                               (int)o.i[3], stream);
            break;
        case GHN3_OP_ATTN_BWD:
            rc = ghn3_attn_bwd(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), R.get<const float>(o.r[2]),
                               R.get<const float>(o.r[3]), R.get<const float>(o.r[4]), R.get<float>(o.r[5]),
                               R.get<float>(o.r[6]), R.get<const int>(o.r[7]), (int)o.i[0], (int)o.i[1], (int)o.i[2],
                               (int)o.i[3], (int)o.i[4], stream);
            break;
        case GHN3_OP_TILE_FWD: {
            const float* srcs[6];
            for (int j = 0; j < 6; ++j) srcs[j] = R.get<const float>(o.r[1 + j]);
            const ghn3_tile_desc* dd = R.get<const ghn3_tile_desc>(o.r[7]);
            rc = ghn3_tile_fwd(R.get<float>(o.r[0]), srcs, dd, (int)o.i[0], o.i[1],
                               reinterpret_cast<const int64_t*>(reinterpret_cast<const char*>(dd) + o.i[2]),
                               (int)o.i[3], R.get<float>(o.r[8]), R.get<float>(o.r[9]), stream);
            break;
        }
        case GHN3_OP_TILE_BWD: {
            const float* srcs[6]; float* dsrcs[6];
            // (source slot 5 -- r6 -- is never a tile source: with the fused norm loss it carries the device float g)
            for (int j = 0; j < 6; ++j) { srcs[j] = j < 5 ? R.get<const float>(o.r[1 + j]) : nullptr; dsrcs[j] = j < 5 ? R.get<float>(o.r[8 + j]) : nullptr; }
            const ghn3_tile_desc* dd = R.get<const ghn3_tile_desc>(o.r[7]);
            const float* norms = R.get<const float>(o.r[14]);
            rc = ghn3_tile_bwd(R.get<const float>(o.r[0]), srcs, dsrcs, dd, (int)o.i[0], o.i[1],
                               reinterpret_cast<const int64_t*>(reinterpret_cast<const char*>(dd) + o.i[2]),
                               (int)o.i[3], R.get<float>(o.r[13]), R.get<const float>(o.r[15]), norms,
                               norms ? reinterpret_cast<const int*>(reinterpret_cast<const char*>(dd) + o.i[4]) : nullptr,
                               norms ? R.get<const float>(o.r[6]) : nullptr,
                               (norms && o.i[5] > 0) ? reinterpret_cast<const int64_t*>(reinterpret_cast<const char*>(dd) + o.i[5]) : nullptr,
                               (int)o.i[6], stream);
            break;
        }
        case GHN3_OP_PARAM_NORM_FWD:
            if (o.i[1] <= 0) { ghn3_set_error("PARAM_NORM_FWD: i1 (flat extent) missing"); rc = GHN3_E_ARG; break; }
            rc = ghn3_param_norm_fwd(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), R.get<const int64_t>(o.r[2]),
                                     R.get<float>(o.r[3]), (int)o.i[0], (int64_t)o.i[1], R.get<const int>(o.r[4]),
                                     R.get<float>(o.r[5]), stream);
            break;
        case GHN3_OP_PARAM_NORM_FIN:
            rc = ghn3_param_norm_fin(R.get<float>(o.r[0]), R.get<float>(o.r[1]), R.get<const float>(o.r[2]),
                                     R.get<const int>(o.r[3]), (int)o.i[0], R.get<const float>(o.r[4]), R.get<float>(o.r[5]),
                                     R.get<float>(o.r[6]), stream);
            break;
        case GHN3_OP_PARAM_NORM_BWD:
            if (o.i[1] <= 0) { ghn3_set_error("PARAM_NORM_BWD: i1 (flat extent) missing"); rc = GHN3_E_ARG; break; }
            rc = ghn3_param_norm_bwd(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), R.get<const int64_t>(o.r[2]),
                                     R.get<const float>(o.r[3]), (int)o.i[0], o.f[0], (int64_t)o.i[1], R.get<const int>(o.r[4]),
                                     stream);
            break;
        case GHN3_OP_COLSUM:
            rc = ghn3_colsum(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), (int)o.i[0], (int)o.i[1], (int)o.i[2],
                             (int)o.i[3], (int)o.i[4], (int)o.i[5] ? (int)o.i[5] : 1, (int)o.i[6],
                             R.get<const int>(o.r[2]), stream);
            break;
        case GHN3_OP_ROWSEG_SUM:
            rc = ghn3_rowseg_sum(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), R.get<const int>(o.r[2]),
                                 R.get<const int>(o.r[3]), (int)o.i[0], (int)o.i[1], (int)o.i[2], (int)o.i[3],
                                 (int)o.i[4], stream);
            break;
        case GHN3_OP_MEMSET0: {
            void* p = R.get<void>(o.r[0]);
            if (p && o.i[0] > 0) HIPCHK(hipMemsetAsync(p, 0, (size_t)o.i[0], stream));
            break;
        }
        case GHN3_OP_ADD:
            rc = ghn3_add(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), o.i[0], stream);
            break;
        case GHN3_OP_TRANSPOSE32:
            rc = ghn3_transpose32(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), (int)o.i[0], (int)o.i[1], (int)o.i[2],
                                  (int)o.i[3], (int)o.i[4], o.i[5], o.i[6], stream);
            break;
        case GHN3_OP_WIRE_PACK:
            rc = ghn3_wire_pack(R.get<void>(o.r[0]), R.get<const void>(o.r[1]), o.i[0], o.i[1], (int)o.i[2], stream);
            break;
        case GHN3_OP_RANK_REDUCE:
            rc = ghn3_rank_reduce(R.get<void>(o.r[0]), R.get<const void>(o.r[1]), o.i[0], (int)o.i[1], (int)o.i[2],
                                  (int)o.i[3], o.f[0], stream);
            break;
        case GHN3_OP_CAST16:
            rc = ghn3_cast16(R.get<const float>(o.r[0]), R.get<void>(o.r[1]), R.get<const ghn3_cast_desc>(o.r[2]),
                             (int)o.i[0], (int)o.i[1], R.get<float>(o.r[3]), R.get<const float>(o.r[4]), (int)o.i[2],
                             stream);
            break;
        case GHN3_OP_SUMSQ:
            rc = ghn3_sumsq(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), o.i[0], R.get<float>(o.r[2]), o.i[1], o.i[2],
                            R.get<const float>(o.r[3]), (int)o.i[3], stream);
            break;
        case GHN3_OP_ADAMW: {
            double h[7];
            memcpy(h, &o.i[1], sizeof(h));            // i[1..7] carry IEEE-754 double bit patterns
            rc = ghn3_adamw(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), R.get<float>(o.r[2]), R.get<float>(o.r[3]),
                            o.i[0], R.get<const float>(o.r[4]), (float)h[0], (float)h[1], (float)h[2], (float)h[3],
                            (float)h[4], (float)h[5], (float)h[6], o.f[0], o.f[1], stream);
            break;
        }
        case GHN3_OP_ADAMW_CAST16: {
            double h[7];
            memcpy(h, &o.i[1], sizeof(h));
            rc = ghn3_adamw_cast16(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), R.get<float>(o.r[2]), R.get<float>(o.r[3]),
                                   R.get<void>(o.r[5]), R.get<const ghn3_cast_desc>(o.r[6]), (int)(o.i[0] & 0xffffffff),
                                   (int)(o.i[0] >> 32), R.get<const float>(o.r[4]), (float)h[0], (float)h[1], (float)h[2],
                                   (float)h[3], (float)h[4], (float)h[5], (float)h[6], o.f[0], o.f[1], stream);
            break;
        }
        case GHN3_OP_RELU_FIX:
            rc = ghn3_relu_fix(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), R.get<const float>(o.r[2]),
                               R.get<const float>(o.r[3]), (int)o.i[0], (int)o.i[1], (int)o.i[2], (int)o.i[3], (int)o.i[4],
                               (int)o.i[5], o.f[0], stream);
            break;
        case GHN3_OP_DACT:
            rc = ghn3_dact(R.get<float>(o.r[0]), R.get<const float>(o.r[1]), (int)o.i[0], (int)o.i[1], (int)o.i[2],
                           (int)o.i[3], R.get<float>(o.r[2]), R.get<const float>(o.r[3]), (int)o.i[4], (int64_t)o.i[5],
                           (int)o.i[6], stream);
            break;
        default:
            ghn3_set_error("op %d: unknown kind %d", k, o.kind);
            return GHN3_E_ARG;
        }
        if (rc) return rc;
        if (R.bad) { ghn3_set_error("op %d (kind %d): reference to an absent buffer", k, o.kind); return GHN3_E_ARG; }
        if (timed) {
            HIPCHK(hipEventRecord((*c->pool)[2 * c->pool_used + 1], stream));
            c->pool_used++;
        }
        if (c->profile == 1) {
            HIPCHK(hipEventRecord(c->pe1, stream));
            HIPCHK(hipEventSynchronize(c->pe1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, c->pe0, c->pe1));
            c->ms[o.kind] += ms;
            c->launches[o.kind] += 1;
        }
    }
    auto mark_done = [&]() -> int {                  // the slot's table may be overwritten once these kernels are done
        if (used_slot >= 0) {
            if (side_dirty || c->side_pending) {         // (side-stream kernels read the table too)
                HIPCHK(hipEventRecord(c->ev_join, c->side));
                HIPCHK(hipStreamWaitEvent(c->copy, c->ev_join, 0));
            }
            HIPCHK(hipEventRecord(c->ev_done[used_slot], main_stream));
            c->done_used[used_slot] = true;
        }
        return GHN3_OK;
    };
    if (detach && side_dirty) {                          // joined by the next run / ghn3_ctx_side_wait
        c->side_pending = true;
        return mark_done();
    }
    if (!touches_side) {                                 // (side state untouched, see above)
        return mark_done();
    }
    int rc_join = join();
    if (rc_join) return rc_join;
    return mark_done();
}

extern "C" int ghn3_ctx_cache_stats(ghn3_ctx* c, int64_t* hits, int64_t* misses) {
    if (!c) return GHN3_E_NOCTX;
    if (hits) *hits = (int64_t)c->cache_hits;
    if (misses) *misses = (int64_t)c->cache_misses;
    return GHN3_OK;
}

extern "C" int ghn3_ctx_side_wait(ghn3_ctx* c, void* stream) {
    if (!c) return GHN3_E_NOCTX;
    HIPCHK(hipEventRecord(c->ev_join, c->side));
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->ev_join, 0));
    return GHN3_OK;
}

extern "C" int ghn3_ctx_side_pending(ghn3_ctx* c) {
    if (!c) return -1;
    return c->side_pending ? 1 : 0;
}

// ---- timing helpers -----------------------------------------------------------------------------
extern "C" int ghn3_event_create(void** ev) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    *ev = (void*)e;
    return GHN3_OK;
}
extern "C" int ghn3_event_record(void* ev, void* stream) {
    HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return GHN3_OK;
}
extern "C" int ghn3_event_elapsed_ms(void* start, void* stop, float* ms) {
    HIPCHK(hipEventSynchronize((hipEvent_t)stop));
    HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return GHN3_OK;
}
extern "C" int ghn3_event_destroy(void* ev) {
    HIPCHK(hipEventDestroy((hipEvent_t)ev));
    return GHN3_OK;
}
