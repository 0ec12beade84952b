// Model-level orchestration: VQModel encode/decode (reference stage1/vqmodel.py:21-41), the
// stage-2 CondTransformer forward (stage2/transformer.py:80-93) and the MaskGIT sample/generate
// loop (generate.py:159-198), expressed as sequences of the operator-level launches of this
// library on one HIP stream.  No host synchronisation happens inside a forward pass.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>

#include "common.h"

// ------------------------------------------------------------------------------------------------
// error + timing plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

void pm_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

extern "C" const char* pmhip_last_error(void) { return g_err.c_str(); }
extern "C" int pmhip_abi_version(void) { return PMHIP_ABI_VERSION; }

extern "C" int pmhip_device_info(int device, int* cu_count, int* lds_bytes, char* arch, int arch_len) {
    hipDeviceProp_t prop;
    PM_HIP(hipGetDeviceProperties(&prop, device));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
    if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", prop.gcnArchName);
    return PMHIP_OK;
}

// Timing is a process-wide diagnostic that concurrent lanes (one host thread each, generate.py) may run into: the switch is
// atomic and the per-family accumulators are guarded by one mutex (taken only while timing is on).
std::atomic<bool> g_pm_timing_on{false};
namespace {
std::mutex g_fam_mu;
struct FamStat {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    int launches = 0;
    double ms = 0.0;
};
FamStat g_fam[FAM_COUNT];
const char* kFamNames[FAM_COUNT] = {"gemm_plain", "attention", "layernorm", "sample", "vq", "rowops", "gemm_heads", "gemm_swiglu", "gemm_resid",
                                    "gemm_resid2b"};
const int kGemmFams[5] = {FAM_GEMM, FAM_GEMM_HEADS, FAM_GEMM_SWIGLU, FAM_GEMM_RESID, FAM_GEMM_RESID2B};

void drain(FamStat& f) {
    for (auto& pr : f.pending) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            f.ms += ms;
            f.launches += 1;
        }
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    f.pending.clear();
}
}  // namespace

PmTimer::PmTimer(int fam, hipStream_t s) : family(fam), stream(s), e0(nullptr), on(g_pm_timing_on.load(std::memory_order_relaxed)) {
    if (on) {
        if (hipEventCreate(&e0) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(e0, stream);
    }
}
PmTimer::~PmTimer() {
    if (!on) return;
    hipEvent_t e1;
    if (hipEventCreate(&e1) != hipSuccess) return;
    (void)hipEventRecord(e1, stream);
    std::lock_guard<std::mutex> lk(g_fam_mu);
    g_fam[family].pending.emplace_back(e0, e1);
}

extern "C" int pmhip_timing_enable(int on) { g_pm_timing_on.store(on != 0); return PMHIP_OK; }
extern "C" int pmhip_timing_reset(void) {
    std::lock_guard<std::mutex> lk(g_fam_mu);
    for (auto& f : g_fam) { drain(f); f.launches = 0; f.ms = 0.0; }
    return PMHIP_OK;
}
extern "C" int pmhip_timing_get(const char* family, int* launches, double* total_ms) {
    if (family && std::string(family) == "gemm") {           // the whole GEMM family
        std::lock_guard<std::mutex> lk(g_fam_mu);
        int n = 0; double ms = 0.0;
        for (int i : kGemmFams) { drain(g_fam[i]); n += g_fam[i].launches; ms += g_fam[i].ms; }
        if (launches) *launches = n;
        if (total_ms) *total_ms = ms;
        return PMHIP_OK;
    }
    for (int i = 0; i < FAM_COUNT; ++i)
        if (family && std::string(family) == kFamNames[i]) {
            std::lock_guard<std::mutex> lk(g_fam_mu);
            drain(g_fam[i]);
            if (launches) *launches = g_fam[i].launches;
            if (total_ms) *total_ms = g_fam[i].ms;
            return PMHIP_OK;
        }
    pm_set_error("timing_get: unknown family '%s'", family ? family : "(null)");
    return PMHIP_EINVAL;
}

// ------------------------------------------------------------------------------------------------
// workspace: named device buffers that only ever grow (the library's only allocations)
// ------------------------------------------------------------------------------------------------
namespace {

struct Workspace {
    std::map<std::string, std::pair<void*, size_t>> bufs;
    bool frozen = false;   // set while a graph capture is in flight: growing would be a bug
    uint64_t gen = 0;      // bumped by every (re)allocation: a captured graph holds raw buffer pointers and is only
                           // valid for the generation it was captured at
    ~Workspace() {
        for (auto& kv : bufs)
            if (kv.second.first) (void)hipFree(kv.second.first);
    }
    int get(const char* name, size_t bytes, void** out, hipStream_t s) {
        auto& e = bufs[name];
        if (e.second < bytes) {
            if (frozen) { pm_set_error("workspace '%s' would grow during graph capture", name); return PMHIP_ESTATE; }
            if (e.first) {
                PM_HIP(hipStreamSynchronize(s));
                PM_HIP(hipFree(e.first));
                e.first = nullptr; e.second = 0;
            }
            const size_t want = (bytes + 255) & ~(size_t)255;
            hipError_t rc = hipMalloc(&e.first, want);
            if (rc != hipSuccess) {
                pm_set_error("hipMalloc(%zu bytes) for workspace '%s' failed: %s", want, name, hipGetErrorString(rc));
                e.first = nullptr;
                return PMHIP_ENOMEM;
            }
            e.second = want;
            ++gen;
        }
        *out = e.first;
        return PMHIP_OK;
    }
};

#define WS(ws, name, bytes, ptr) PM_TRY((ws).get(name, (size_t)(bytes), (void**)&(ptr), s))

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Stream-ordered copy by a KERNEL (device memory, or pinned host memory through its device alias): the graph-replayed decode loop
// refreshes its per-call parameter block and ids with it, so that the refresh is a queue packet like the graph's own kernels.
// Measured (profiles/r05_c_host_polling.txt, tools/hwtests/graph_dispatch_mode.hip): this did NOT cure the mis-ordered replays
// seen under AMD_DIRECT_DISPATCH=0 -- graph replay itself is broken in that runtime mode on ROCm 7.2, which is why the loop runs
// eagerly there (generate_loop) -- it is kept because it costs nothing and keeps the replay path free of hipMemcpyAsync nodes.
// 16-byte words when size and both pointers allow it, 8-byte words otherwise (ids are int64: any B * tokens is a multiple of 8),
// hipMemcpyAsync for anything else.
template <typename W>
__global__ void copy_words_kernel(W* __restrict__ dst, const W* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
template <typename W>
int copy_words_async(void* dst, const void* src, size_t n, hipStream_t s) {
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(copy_words_kernel<W>, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, (W*)dst, (const W*)src, n);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}
int copy16_async(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return PMHIP_OK;
    const uintptr_t bits = (uintptr_t)dst | (uintptr_t)src | (uintptr_t)bytes;
    if (bits % 16 == 0) return copy_words_async<uint4>(dst, src, bytes / 16, s);
    if (bits % 8 == 0) return copy_words_async<unsigned long long>(dst, src, bytes / 8, s);
    PM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, s));
    return PMHIP_OK;
}
constexpr float kLog2e = 1.4426950408889634f;

struct CrossKV {            // cached cross-attention K / V^T of a static context, per layer
    const void* k = nullptr;
    const void* vt = nullptr;
    int L = 0, Lp = 0;
};

struct TowerBufs {
    float* x = nullptr;     // residual stream, fp32 [M, dim]                      (fp32-verify mode)
    void* xh = nullptr;     // residual stream as a bf16 pair: x = hi + lo, [M, dim] each   (bf16-perf mode).  The hi plane IS
    void* xl = nullptr;     //   bf16(x): the operand of every GEMM that consumes LN(x) with the LayerNorm folded in
    void* y = nullptr;      // normed activations, T [M, dim] (unfolded LayerNorm output)
    void* q = nullptr; void* k = nullptr; void* vt = nullptr;
    void* attn = nullptr;   // T [M, inner]
    void* hid = nullptr;    // T [M, hidden_pad]
    float* split = nullptr; // dim_head != 64 only: f32 scratch of the plain head-split path [M, 3*inner]
    float* coef = nullptr;  // LayerNorm fold: per-row (rstd, -rstd * mean) of the hi plane
    float* parts = nullptr; // ... and the partial row statistics the last residual GEMM left behind, [M, dim/64, 2]
    bool parts_valid = false;   // parts describe the CURRENT hi plane (set by residual_gemm, see tower_coef)
    // centred hi plane (gemm_common.h, GemmParams::center_coef): the producers subtract the row mean the last LayerNorm saw
    bool center = false;        // mode on (fold on and not PMHIP_HILO_CENTER=0)
    bool coef_valid = false;    // coef holds the statistics of the CURRENT hi plane (set by tower_coef, cleared by a producer)
    bool coef_deferred = false; // ... but nobody has WRITTEN them yet: tower_coef left that to the first folded consumer, which
                                // materialises them from `parts` (fold_desc hands it the parts and clears this flag).  A reader of
                                // coef -- the centred producer -- refuses to run while it is set (round-5 advisor: the invariant was implicit)
    float* shift = nullptr;     // per-row running sum of the subtracted means: only where the absolute x is needed again (ViT encoder)
    bool hilo = false;      // bf16 mode
    int fold_rows_cap = 0;  // Switches::fold_rows_cap
    bool stats = true;      // producers leave row statistics (PMHIP_LN_STATS=0: the coefficient pass over the plane, A/B tests)
    bool fold = false;      // hilo and folding not disabled (PMHIP_LN_UNFOLD=1 forces the separate LayerNorm kernel: A/B tests)
};

// where a residual GEMM takes its addend from: fp32 rows (verify mode), or a pair of bf16 planes (bf16 mode)
struct ResSrc {
    const float* f32 = nullptr;
    const void* hi = nullptr; const void* lo = nullptr;
    int ld = 0, rows = 0;      // row stride; rows > 0: the addend row is m % rows (position embedding)
};

// bf16 mode: the LayerNorm that precedes every projection (stage1/layers.py:54-58, stage2/transformer.py:44-49) is folded into
// that projection wherever the 256x256 kernel serves the shape: the consumer multiplies the hi plane by gamma-scaled weights
// and normalises in its epilogue with the row's (rstd, -rstd * mean) from pmhip_ln_coef.  Elsewhere (small batches, the
// decoder's 192-wide projection) pmhip_layernorm_hilo writes LN(x) and the plain GEMM runs.
inline int dh_of(const pmhip_tower_cfg& tc) { return tc.dim_head > 0 ? tc.dim_head : 64; }

int check_dim_head(const char* who, const pmhip_tower_cfg& tc) {
    const int dh = dh_of(tc);
    PM_REQUIRE(tc.heads > 0 && dh >= 16 && dh <= 128 && dh % 16 == 0 && (tc.heads * dh) % 64 == 0,
               "%s: heads=%d dim_head=%d: dim_head must be a multiple of 16 in [16,128] and heads*dim_head a multiple of 64", who, tc.heads, dh);
    return PMHIP_OK;
}

// The bf16 hi/lo residual stream + folded LayerNorm is the DEFAULT of bf16 mode (PMHIP_HILO=0, read when a handle is created, restores
// the fp32 stream + LayerNorm kernel of rounds 1-2): 1.3 % faster on the default workload (same box: 476.5 vs 470.6 images/s),
// as accurate.  It was opt-in while 3-11 of 1800 generate() calls under concurrent lanes were not
// bit-identical; that was a gfx950 packed-FP32 operand-select hazard in the folded epilogue, fixed in
// gemm_common.h and gated by tests/test_isa_hazards.py.
// The PMHIP_* switches are read ONCE, when a handle is created (a handle's graphs, workspace and fold decisions all depend on
// them; a getenv on the hot path is also a data race with a caller that edits the environment from another thread).
struct Switches {
    int overlap_rows = 65536;   // PMHIP_DECODE_OVERLAP_MAX_ROWS: largest B * tokens whose decode loop defers each step's ViT decode
                                // to a side stream beside the next step's tower (0 = never)
    bool hilo = true;       // PMHIP_HILO=0: the fp32 stream + LayerNorm kernel of rounds 1-2
    bool fold = true;       // PMHIP_LN_UNFOLD=1: the hi/lo pair, but the separate LayerNorm kernel
    bool stats = true;      // PMHIP_LN_STATS=0: fold coefficients by a pass over the hi plane
    static Switches from_env() {
        Switches w;
        const char* e = getenv("PMHIP_HILO");
        w.hilo = !(e && atoi(e) == 0);
        e = getenv("PMHIP_LN_UNFOLD");
        w.fold = !(e && atoi(e) != 0);
        e = getenv("PMHIP_LN_STATS");
        w.stats = !(e && atoi(e) == 0);
        e = getenv("PMHIP_FOLD_MAX_ROWS");
        w.fold_rows_cap = e ? atoi(e) : 0;
        e = getenv("PMHIP_HILO_CENTER");
        w.center = !(e && atoi(e) == 0);
        e = getenv("PMHIP_BLOCKING_WAIT");
        w.blocking_wait = e && atoi(e) != 0;
        e = getenv("PMHIP_DECODE_OVERLAP_MAX_ROWS");
        if (e) w.overlap_rows = atoi(e);
        w.logits_stats = pm_dev_knob("PMHIP_LOGITS_STATS", 1) != 0;
        return w;
    }
    int fold_rows_cap = 0;  // PMHIP_FOLD_MAX_ROWS (development / tests): cap on the rows one folded launch takes, see fold_rows()
    bool center = true;     // PMHIP_HILO_CENTER=0: the residual producers do not centre the hi plane (A/B, tests)
    bool logits_stats = true;     // PMHIP_LOGITS_STATS=0 (development builds, profiles/r06_d): the logits GEMM leaves no block statistics, the sampling kernel derives
                                  // them from the rows it then has to read in full (same ids and scores, bit for bit)
    bool blocking_wait = false;   // PMHIP_BLOCKING_WAIT=1: host waits between decode-loop segments sleep instead of spinning
    int key() const { return (hilo ? 2 : 0) + (fold ? 1 : 0) + (stats ? 4 : 0) + (center ? 8 : 0); }
};

// widest residual stream the hi/lo row operators (pmhip_split_hilo, pmhip_layernorm_hilo, pmhip_layernorm_to_hilo, pmhip_ln_coef,
// pmhip_join_hilo: one wave per row, the row in registers) serve; wider bf16 towers keep the fp32 stream + pmhip_layernorm
constexpr int kHiloMaxDim = 1024;

int alloc_tower(Workspace& ws, const Switches& sw, const char* tag, int dtype, const pmhip_tower_cfg& tc, int B, int tokens, TowerBufs& b,
                hipStream_t s) {
    const size_t es = dtype_size(dtype);
    const size_t M = (size_t)B * tokens;
    const int dh = dh_of(tc), inner = tc.heads * dh, Np = round_up(tokens, 64);
    std::string t(tag);
    b.hilo = dtype == PMHIP_BF16 && sw.hilo && tc.dim <= kHiloMaxDim;
    if (b.hilo) {
        WS(ws, (t + ".xh").c_str(), M * tc.dim * 2, b.xh);
        WS(ws, (t + ".xl").c_str(), M * tc.dim * 2, b.xl);
    } else {
        WS(ws, (t + ".x").c_str(), M * tc.dim * 4, b.x);
    }
    WS(ws, (t + ".y").c_str(), M * tc.dim * es, b.y);
    WS(ws, (t + ".q").c_str(), (size_t)B * tc.heads * tokens * dh * es, b.q);
    WS(ws, (t + ".k").c_str(), (size_t)B * tc.heads * Np * dh * es, b.k);
    WS(ws, (t + ".vt").c_str(), (size_t)B * tc.heads * Np * dh * es, b.vt);
    WS(ws, (t + ".attn").c_str(), M * inner * es, b.attn);
    WS(ws, (t + ".hid").c_str(), M * tc.hidden_pad * es, b.hid);
    WS(ws, (t + ".coef").c_str(), M * 2 * 4, b.coef);
    WS(ws, (t + ".parts").c_str(), M * (size_t)round_up(tc.dim, 64) / 64 * 2 * 4, b.parts);
    b.parts_valid = false;
    b.split = nullptr;
    if (dh != 64) WS(ws, (t + ".split").c_str(), M * 3 * inner * 4, b.split);
    b.fold = b.hilo && dh == 64 && sw.fold;
    b.stats = sw.stats;
    b.fold_rows_cap = sw.fold_rows_cap;
    b.center = b.fold && sw.center;
    b.coef_valid = false;
    b.shift = nullptr;
    return PMHIP_OK;
}

// the residual stream itself as the addend (x += ...)
ResSrc res_self(const TowerBufs& b, int dim) {
    ResSrc r;
    r.f32 = b.x; r.hi = b.xh; r.lo = b.xl; r.ld = dim; r.rows = 0;
    return r;
}

// residual GEMM x = A . W^T + bias + addend, written to the tower's residual stream (in place when the addend is the stream)
int residual_gemm(int dtype, TowerBufs& b, const void* A, int lda, const void* W, int ldw, const float* bias, const ResSrc& r,
                  int M, int N, int K, hipStream_t s, float bias_mean = 0.f) {
    if (b.hilo) {
        // with the LayerNorm folded into the consumers the producer's epilogue also leaves the row statistics of the new hi plane
        // (16 bytes per 64 columns): the coefficient pass over the plane (pmhip_ln_coef, 14 us per launch at the bench shape)
        // becomes a combination of 8-16 partials per row.  PMHIP_LN_STATS=0: the pass (A/B).
        b.parts_valid = b.fold && N % 64 == 0 && N <= 1024 && b.stats;   // pmhip_ln_coef_parts combines <= 16 parts
        const bool self = r.hi == b.xh && r.rows == 0;                 // x += ... (not the GEMM that opens the stream)
        const float* cc = (b.center && self && b.coef_valid) ? b.coef : nullptr;
        PM_REQUIRE(!cc || !b.coef_deferred, "residual_gemm: the fold coefficients of this LayerNorm were left to a folded consumer that never ran");
        const int shift_mode = !b.shift ? 0 : (self ? (cc ? 2 : 0) : 1);
        b.coef_valid = false;                                          // the hi plane changes
        if (cc || b.shift)
            return pmhip_gemm_hilo_center(A, lda, W, ldw, bias, r.hi, r.lo, r.ld, r.rows, b.xh, b.xl, N, M, N, K, b.parts_valid ? b.parts : nullptr,
                                          cc, bias_mean, b.shift, shift_mode, s);
        if (b.parts_valid) return pmhip_gemm_hilo_stats(A, lda, W, ldw, bias, r.hi, r.lo, r.ld, r.rows, b.xh, b.xl, N, M, N, K, b.parts, s);
        return pmhip_gemm_hilo(A, lda, W, ldw, bias, r.hi, r.lo, r.ld, r.rows, b.xh, b.xl, N, M, N, K, s);
    }
    return pmhip_gemm(dtype, A, lda, W, ldw, bias, r.f32, r.ld, r.rows, b.x, N, PMHIP_F32, M, N, K, s);
}

// (rstd, -rstd * mean) of the rows of the current hi plane into b.coef.  Where the last residual GEMM left partial row statistics
// behind, the folded consumer itself produces coef from them (pmhip_lnfold::parts: in its own prologue for a small launch, by
// pmhip_ln_coef_parts in front of it otherwise) and nothing is launched here; else the pass over the plane.
int tower_coef(TowerBufs& b, int M, int dim, hipStream_t s) {
    b.coef_valid = true;
    b.coef_deferred = b.parts_valid;
    if (b.parts_valid) return PMHIP_OK;
    return pmhip_ln_coef(b.xh, 1e-5f, b.coef, M, dim, s);
}
// the fold descriptor of the rows [m0, ...) of the tower
// the ONLY place a fold descriptor is built: a consumer handed a descriptor without `parts` would read coefficients nobody wrote
pmhip_lnfold fold_desc(TowerBufs& b, int m0, int dim, const float* c, const float* d) {
    pmhip_lnfold ln{};
    ln.coef = b.coef + (size_t)m0 * 2; ln.c = c; ln.d = d;
    if (b.parts_valid) { ln.parts = b.parts + (size_t)m0 * (dim / 64) * 2; ln.nparts = dim / 64; ln.eps = 1e-5f; }
    b.coef_deferred = false;                                  // the consumer about to be launched writes coef (from parts) or finds it written
    return ln;
}

// LN(x) of the tower's residual stream into b.y (the unfolded path)
int tower_layernorm(int dtype, TowerBufs& b, const float* g, const float* be, int M, int dim, hipStream_t s) {
    if (b.hilo) return pmhip_layernorm_hilo(b.xh, b.xl, g, be, 1e-5f, b.y, dtype, M, dim, s);
    return pmhip_layernorm(b.x, g, be, 1e-5f, b.y, dtype, M, dim, s);
}

// Folded or not must not depend on the BATCH (the two routes differ in rounding, and an image's result may not depend on
// how many images run with it -- lanes, ranks and batch sizes all give bit-identical images): the decision looks at the
// per-image row count and the weight shape only, and the 256x256 kernel then takes the shape whatever its tile count.
// The 256x256 kernel addresses its operands with 32-bit byte offsets (set_lnfold in gemm.hip refuses M * lda * 2 >= 2^31): the
// weight and ONE image's rows must fit -- a larger batch is cut into launches of whole images by fold_rows(), so the decision
// stays a function of the shape and the result stays bit-identical for every batch size.
bool fold_shape_ok(int tokens, int n_out, int dim) {
    return tokens % 256 == 0 && n_out % 256 == 0 && dim % 128 == 0 && dim >= 128 && (unsigned long long)n_out * dim * 2 < (1ull << 31) &&
           (unsigned long long)tokens * dim * 2 < (1ull << 31);
}

// rows one folded launch may take: whole images, below the 32-bit byte-offset limit
int fold_rows(const TowerBufs& b, int M, int tokens, int dim) {
    long long rows = ((1ll << 31) - 1) / ((long long)dim * 2);
    if (b.fold_rows_cap > 0 && b.fold_rows_cap < rows) rows = b.fold_rows_cap;
    long long imgs = rows / tokens;
    if (imgs < 1) imgs = 1;
    return (int)std::min<long long>(M, imgs * tokens);
}

// LN(x) -> head-split projection: folded when the shape is served, else LayerNorm + GEMM
int ln_heads(int dtype, TowerBufs& b, const float* g, const float* be, const void* W, const void* Wf, const float* fc, const float* fd,
             int M, int dim, int heads, int dh, int tokens, int Np, int nparts, const int* kinds, void* const* outs, float q_scale,
             hipStream_t s) {
    if (b.fold && Wf && fold_shape_ok(tokens, nparts * heads * 64, dim)) {
        PM_TRY(tower_coef(b, M, dim, s));
        const int step = fold_rows(b, M, tokens, dim);
        for (int m0 = 0; m0 < M; m0 += step) {
            const int rows = std::min(step, M - m0);
            const size_t b0 = (size_t)(m0 / tokens);
            void* o[3] = {nullptr, nullptr, nullptr};
            for (int i = 0; i < nparts; ++i)
                o[i] = reinterpret_cast<unsigned char*>(outs[i]) + b0 * heads * (kinds[i] == PMHIP_PART_Q ? tokens : Np) * 64 * dtype_size(dtype);
            const pmhip_lnfold ln = fold_desc(b, m0, dim, fc, fd);
            PM_TRY(pmhip_gemm_heads_ln(dtype, reinterpret_cast<const unsigned char*>(b.xh) + (size_t)m0 * dim * 2, dim, Wf, dim, rows, dim, heads,
                                       tokens, Np, nparts, kinds, o, q_scale, &ln, s));
        }
        return PMHIP_OK;
    }
    PM_TRY(tower_layernorm(dtype, b, g, be, M, dim, s));
    return pmhip_gemm_heads_dh(dtype, b.y, dim, W, dim, M, dim, heads, dh, tokens, Np, nparts, kinds, outs, q_scale, b.split, s);
}

// one pre-LN transformer block (stage1/layers.py:54-58; stage2/transformer.py:44-49)
int layer_forward(int dtype, const pmhip_layer_weights& L, const pmhip_tower_cfg& tc, TowerBufs& b, int B, int tokens,
                  bool stage2, const CrossKV* cross, hipStream_t s) {
    const int dh = dh_of(tc);
    const int M = B * tokens, dim = tc.dim, inner = tc.heads * dh, Np = round_up(tokens, 64);
    const bool fast = dtype == PMHIP_BF16;
    const float q_scale = (dh == 64 ? 0.125f : 1.0f / sqrtf((float)dh)) * (fast ? kLog2e : 1.0f);   // dim_head^-0.5, attention.py:31,52
    const int kinds_qkv[3] = {PMHIP_PART_Q, PMHIP_PART_K, PMHIP_PART_V};
    const ResSrc self = res_self(b, dim);

    // x = attn1(norm1(x)) + x
    {
        void* outs[3] = {b.q, b.k, b.vt};
        PM_TRY(ln_heads(dtype, b, L.ln1_g, L.ln1_b, L.wqkv, L.wqkv_f, L.qkv_c, L.qkv_d, M, dim, tc.heads, dh, tokens, Np, 3, kinds_qkv, outs,
                        q_scale, s));
    }
    PM_TRY(pmhip_attention_dh(dtype, b.q, b.k, b.vt, b.attn, inner, B, tc.heads, dh, tokens, tokens, Np, fast, s));
    PM_TRY(residual_gemm(dtype, b, b.attn, inner, L.wo, inner, L.bo, self, M, dim, inner, s, L.bo_mean));

    if (stage2) {
        // x = attn2(norm2(x), context) + x ; context None -> a second self-attention (attention.py:47)
        if (cross && cross->k) {
            const int kind_q[1] = {PMHIP_PART_Q};
            void* outs[1] = {b.q};
            PM_TRY(ln_heads(dtype, b, L.lnx_g, L.lnx_b, L.wqkv2, L.wqkv2_f, L.qkv2_c, L.qkv2_d, M, dim, tc.heads, dh, tokens, Np, 1, kind_q,
                            outs, q_scale, s));
            PM_TRY(pmhip_attention_dh(dtype, b.q, cross->k, cross->vt, b.attn, inner, B, tc.heads, dh, tokens, cross->L, cross->Lp, fast, s));
        } else {
            void* outs[3] = {b.q, b.k, b.vt};
            PM_TRY(ln_heads(dtype, b, L.lnx_g, L.lnx_b, L.wqkv2, L.wqkv2_f, L.qkv2_c, L.qkv2_d, M, dim, tc.heads, dh, tokens, Np, 3, kinds_qkv,
                            outs, q_scale, s));
            PM_TRY(pmhip_attention_dh(dtype, b.q, b.k, b.vt, b.attn, inner, B, tc.heads, dh, tokens, tokens, Np, fast, s));
        }
        PM_TRY(residual_gemm(dtype, b, b.attn, inner, L.wo2, inner, L.bo2, self, M, dim, inner, s, L.bo2_mean));
    }

    // x = ffnet(norm(x)) + x
    if (b.fold && L.w12p_f && fold_shape_ok(tokens, 2 * tc.hidden_pad, dim)) {
        PM_TRY(tower_coef(b, M, dim, s));
        const int step = fold_rows(b, M, tokens, dim);
        for (int m0 = 0; m0 < M; m0 += step) {
            const pmhip_lnfold ln = fold_desc(b, m0, dim, L.w12_c, L.w12_d);
            PM_TRY(pmhip_gemm_swiglu_ln(dtype, reinterpret_cast<const unsigned char*>(b.xh) + (size_t)m0 * dim * 2, dim, L.w12p_f, L.b12p,
                                        reinterpret_cast<unsigned char*>(b.hid) + (size_t)m0 * tc.hidden_pad * dtype_size(dtype), tc.hidden_pad,
                                        std::min(step, M - m0), tc.hidden_pad, dim, &ln, s));
        }
    } else {
        PM_TRY(tower_layernorm(dtype, b, L.ln2_g, L.ln2_b, M, dim, s));
        PM_TRY(pmhip_gemm_swiglu(dtype, b.y, dim, L.w12p, L.b12p, b.hid, tc.hidden_pad, M, tc.hidden_pad, dim, s));
    }
    return residual_gemm(dtype, b, b.hid, tc.hidden_pad, L.w3p, tc.hidden_pad, L.b3, self, M, dim, tc.hidden_pad, s, L.b3_mean);
}

// a position embedding as the addend of the GEMM that opens a residual stream: the fp32 table in verify mode, its hi / lo
// planes (split once per handle, kept in the workspace) in bf16 mode
int pos_source(Workspace& ws, const char* tag, bool hilo, const float* pos, int rows, int dim, bool& done, ResSrc& r, hipStream_t s) {
    r = ResSrc{};
    r.ld = dim; r.rows = rows;
    if (!hilo) { r.f32 = pos; return PMHIP_OK; }
    void* hi; void* lo;
    std::string t(tag);
    WS(ws, (t + ".hi").c_str(), (size_t)rows * dim * 2, hi);
    WS(ws, (t + ".lo").c_str(), (size_t)rows * dim * 2, lo);
    if (!done) {
        PM_TRY(pmhip_split_hilo(pos, hi, lo, rows, dim, s));
        done = true;
    }
    r.hi = hi; r.lo = lo;
    return PMHIP_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// VQModel
// ------------------------------------------------------------------------------------------------
static std::atomic<uint64_t> g_next_vq_uid{1};   // handles are created from lane threads too (Pipeline._lanes clones)
struct pmhip_vqgan {
    uint64_t uid = g_next_vq_uid.fetch_add(1, std::memory_order_relaxed);   // graph keys name the handle by this, never by its (reusable) heap address
    int device = 0, dtype = 0;
    pmhip_vqgan_cfg cfg{};
    pmhip_vqgan_weights w{};
    std::vector<pmhip_layer_weights> enc_layers, dec_layers;
    int grid = 0, tokens = 0, patch_k = 0;
    Workspace ws;
    Switches sw = Switches::from_env();
    bool dec_pos_split = false;     // bf16 mode: the decoder position embedding has been split into hi / lo planes (ws "decpos.*")
};

extern "C" int pmhip_vqgan_create(pmhip_vqgan** out, int device, int dtype, const pmhip_vqgan_cfg* cfg,
                                  const pmhip_vqgan_weights* w) {
    PM_REQUIRE(out && cfg && w, "vqgan_create: null argument");
    PM_REQUIRE(dtype == PMHIP_F32 || dtype == PMHIP_BF16, "vqgan_create: bad dtype");
    PM_REQUIRE(cfg->patch_size > 0 && cfg->image_size % cfg->patch_size == 0, "vqgan_create: image/patch mismatch");
    PM_REQUIRE(cfg->patch_size % 8 == 0, "vqgan_create: patch_size must be a multiple of 8");
    PM_REQUIRE((cfg->channels * cfg->patch_size * cfg->patch_size) % 64 == 0, "vqgan_create: C*P*P must be a multiple of 64");
    PM_REQUIRE(cfg->enc.dim % 64 == 0 && cfg->dec.dim % 64 == 0, "vqgan_create: dim must be a multiple of 64");
    PM_REQUIRE(cfg->enc.hidden_pad % 64 == 0 && cfg->dec.hidden_pad % 64 == 0, "vqgan_create: hidden_pad must be a multiple of 64");
    PM_REQUIRE(cfg->embed_dim <= 64 && cfg->embed_dim % 4 == 0, "vqgan_create: embed_dim must be <= 64 and a multiple of 4");
    PM_TRY(check_dim_head("vqgan_create(enc)", cfg->enc));
    PM_TRY(check_dim_head("vqgan_create(dec)", cfg->dec));
    auto h = std::make_unique<pmhip_vqgan>();
    h->device = device; h->dtype = dtype; h->cfg = *cfg; h->w = *w;
    h->enc_layers.assign(w->enc_layers, w->enc_layers + cfg->enc.depth);
    h->dec_layers.assign(w->dec_layers, w->dec_layers + cfg->dec.depth);
    h->w.enc_layers = h->enc_layers.data();
    h->w.dec_layers = h->dec_layers.data();
    h->grid = cfg->image_size / cfg->patch_size;
    h->tokens = h->grid * h->grid;
    h->patch_k = cfg->channels * cfg->patch_size * cfg->patch_size;
    *out = h.release();
    return PMHIP_OK;
}

extern "C" void pmhip_vqgan_destroy(pmhip_vqgan* h) {
    if (!h) return;
    (void)hipDeviceSynchronize();
    delete h;
}

namespace {

// Encoder.forward (stage1/layers.py:106-112): returns the residual stream in tb.x
int vq_encoder(pmhip_vqgan* h, const float* img, int B, TowerBufs& tb, hipStream_t s) {
    const auto& c = h->cfg;
    const int M = B * h->tokens, dim = c.enc.dim;
    PM_TRY(alloc_tower(h->ws, h->sw, "enc", h->dtype, c.enc, B, h->tokens, tb, s));
    void* pa; float* x0;
    WS(h->ws, "enc.patches", (size_t)M * h->patch_k * dtype_size(h->dtype), pa);
    WS(h->ws, "enc.x0", (size_t)M * dim * 4, x0);
    PM_TRY(pmhip_patchify(img, pa, h->dtype, B, c.channels, c.image_size, c.image_size, c.patch_size, s));
    // conv-as-GEMM (no bias) + position embedding, then norm_pre
    PM_TRY(pmhip_gemm(h->dtype, pa, h->patch_k, h->w.patch_w, h->patch_k, nullptr, h->w.enc_pos, dim, h->tokens, x0, dim,
                      PMHIP_F32, M, dim, h->patch_k, s));
    if (tb.hilo) PM_TRY(pmhip_layernorm_to_hilo(x0, h->w.pre_g, h->w.pre_b, 1e-5f, tb.xh, tb.xl, M, dim, s));
    else PM_TRY(pmhip_layernorm(x0, h->w.pre_g, h->w.pre_b, 1e-5f, tb.x, PMHIP_F32, M, dim, s));
    if (tb.center) {
        // prev_quant (vqmodel.py:23) consumes the ABSOLUTE residual stream, so this tower keeps the running shift of its centred
        // hi plane and folds it back in at the end (the decoder and the stage-2 tower end in a LayerNorm, which never sees it)
        WS(h->ws, "enc.shift", (size_t)M * 4, tb.shift);
        PM_HIP(hipMemsetAsync(tb.shift, 0, (size_t)M * 4, s));
    }
    for (int l = 0; l < c.enc.depth; ++l)
        PM_TRY(layer_forward(h->dtype, h->enc_layers[l], c.enc, tb, B, h->tokens, false, nullptr, s));
    if (tb.shift) PM_TRY(pmhip_unshift_hilo(tb.xh, tb.xl, tb.shift, M, dim, s));
    return PMHIP_OK;
}

// transformer + norm + proj + un-patchify of Decoder.forward (stage1/layers.py:147-150); tb.x holds
// x + position_embedding on entry
int vq_decoder_tower(pmhip_vqgan* h, TowerBufs& tb, int B, float* img_out, bool clamp, hipStream_t s) {
    const auto& c = h->cfg;
    const int M = B * h->tokens, dim = c.dec.dim;
    for (int l = 0; l < c.dec.depth; ++l)
        PM_TRY(layer_forward(h->dtype, h->dec_layers[l], c.dec, tb, B, h->tokens, false, nullptr, s));
    float* yo;
    WS(h->ws, "dec.pixels", (size_t)M * h->patch_k * 4, yo);
    PM_TRY(tower_layernorm(h->dtype, tb, h->w.dn_g, h->w.dn_b, M, dim, s));
    PM_TRY(pmhip_gemm(h->dtype, tb.y, dim, h->w.proj_w, dim, h->w.proj_b, nullptr, 0, 0, yo, h->patch_k, PMHIP_F32, M,
                      h->patch_k, dim, s));
    const float lim = clamp ? 1.0f : INFINITY;
    return pmhip_unpatchify_clamp(yo, img_out, B, c.channels, c.image_size, c.image_size, c.patch_size, -lim, lim, s);
}

// VQModel.decode from zp = T [M,64] latent rows (vqmodel.py:27-30)
int vq_decode_latent(pmhip_vqgan* h, const void* zp, int B, float* img_out, hipStream_t s) {
    const auto& c = h->cfg;
    const int M = B * h->tokens, dim = c.dec.dim;
    TowerBufs tb;
    PM_TRY(alloc_tower(h->ws, h->sw, "dec", h->dtype, c.dec, B, h->tokens, tb, s));
    // post_quant + position embedding fused (vqmodel.py:28, layers.py:146)
    ResSrc pos;
    PM_TRY(pos_source(h->ws, "decpos", tb.hilo, h->w.dec_pos, h->tokens, dim, h->dec_pos_split, pos, s));
    PM_TRY(residual_gemm(h->dtype, tb, zp, 64, h->w.postq_w, 64, h->w.postq_b, pos, M, dim, 64, s));
    return vq_decoder_tower(h, tb, B, img_out, true, s);
}

int vq_decode_indices(pmhip_vqgan* h, const int64_t* idx, int B, float* img_out, hipStream_t s) {
    const int M = B * h->tokens;
    void* zp;
    WS(h->ws, "dec.zp", (size_t)M * 64 * dtype_size(h->dtype), zp);
    // l2norm(embedding(idx)) == a row of the pre-normalised codebook (quantize.py:40-44)
    PM_TRY(pmhip_embed_rows(h->w.codebook_n, idx, zp, h->dtype, 64, M, h->cfg.n_embed, h->cfg.embed_dim, s));
    return vq_decode_latent(h, zp, B, img_out, s);
}

}  // namespace

extern "C" int pmhip_vqgan_encoder_forward(pmhip_vqgan* h, const float* img, int B, float* x_out, pmhip_stream stream) {
    PM_REQUIRE(h && img && x_out && B > 0, "encoder_forward: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    TowerBufs tb;
    PM_TRY(vq_encoder(h, img, B, tb, s));
    if (tb.hilo) return pmhip_join_hilo(tb.xh, tb.xl, x_out, B * h->tokens, h->cfg.enc.dim, s);
    PM_HIP(hipMemcpyAsync(x_out, tb.x, (size_t)B * h->tokens * h->cfg.enc.dim * 4, hipMemcpyDeviceToDevice, s));
    return PMHIP_OK;
}

extern "C" int pmhip_vqgan_encode(pmhip_vqgan* h, const float* img, int B, float* z_out, int64_t* idx_out,
                                  float* loss_out, pmhip_stream stream) {
    PM_REQUIRE(h && img && idx_out && B > 0, "vqgan_encode: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const auto& c = h->cfg;
    const int M = B * h->tokens, dim = c.enc.dim, E = c.embed_dim;
    TowerBufs tb;
    PM_TRY(vq_encoder(h, img, B, tb, s));
    // prev_quant acts on the raw residual stream (vqmodel.py:23): cast it to T when T != f32 (with the hi/lo stream the
    // operand bf16(x) is the hi plane itself)
    const void* xin = tb.hilo ? tb.xh : (const void*)tb.x;
    if (!tb.hilo && h->dtype != PMHIP_F32) {
        PM_TRY(pmhip_convert_pad(tb.x, dim, tb.y, h->dtype, dim, M, s));
        xin = tb.y;
    }
    float* ze; void* scratch;
    WS(h->ws, "enc.ze", (size_t)M * E * 4, ze);
    WS(h->ws, "enc.vq", pmhip_vq_scratch_bytes(M, c.n_embed), scratch);
    PM_TRY(pmhip_gemm(h->dtype, xin, dim, h->w.prevq_w, dim, h->w.prevq_b, nullptr, 0, 0, ze, E, PMHIP_F32, M, E, dim, s));
    return pmhip_vq_quantize(ze, h->w.codebook_n, h->w.codebook_sq, c.beta, z_out, idx_out, loss_out, scratch, M, c.n_embed,
                             E, s);
}

extern "C" int pmhip_vqgan_decode(pmhip_vqgan* h, const float* z, int B, float* img_out, pmhip_stream stream) {
    PM_REQUIRE(h && z && img_out && B > 0, "vqgan_decode: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int M = B * h->tokens;
    void* zp;
    WS(h->ws, "dec.zp", (size_t)M * 64 * dtype_size(h->dtype), zp);
    PM_TRY(pmhip_convert_pad(z, h->cfg.embed_dim, zp, h->dtype, 64, M, s));
    return vq_decode_latent(h, zp, B, img_out, s);
}

extern "C" int pmhip_vqgan_decode_indices(pmhip_vqgan* h, const int64_t* idx, int B, float* img_out,
                                          pmhip_stream stream) {
    PM_REQUIRE(h && idx && img_out && B > 0, "vqgan_decode_indices: bad arguments");
    return vq_decode_indices(h, idx, B, img_out, (hipStream_t)stream);
}

extern "C" int pmhip_vqgan_decoder_forward(pmhip_vqgan* h, const float* x, int B, float* img_out, pmhip_stream stream) {
    PM_REQUIRE(h && x && img_out && B > 0, "decoder_forward: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int M = B * h->tokens;
    TowerBufs tb;
    PM_TRY(alloc_tower(h->ws, h->sw, "dec", h->dtype, h->cfg.dec, B, h->tokens, tb, s));
    if (tb.hilo) {
        float* x0;
        WS(h->ws, "dec.x0", (size_t)M * h->cfg.dec.dim * 4, x0);
        PM_TRY(pmhip_add_rows(x, h->w.dec_pos, h->tokens, x0, M, h->cfg.dec.dim, s));
        PM_TRY(pmhip_split_hilo(x0, tb.xh, tb.xl, M, h->cfg.dec.dim, s));
    } else {
        PM_TRY(pmhip_add_rows(x, h->w.dec_pos, h->tokens, tb.x, M, h->cfg.dec.dim, s));
    }
    return vq_decoder_tower(h, tb, B, img_out, false, s);
}

// ------------------------------------------------------------------------------------------------
// stage 2: CondTransformer + MaskGIT loop
// ------------------------------------------------------------------------------------------------
struct GraphEntry {
    bool warmed = false;            // one eager pass has sized every workspace buffer
    std::vector<hipGraphExec_t> segs;   // one executable graph per segment (a segment ends with a decoded step)
    uint64_t s2_gen = 0, vq_gen = 0;   // workspace generations the graphs' baked-in pointers belong to
    void destroy() {
        for (auto e : segs)
            if (e) (void)hipGraphExecDestroy(e);
        segs.clear();
    }
};

struct pmhip_s2 {
    int device = 0, dtype = 0;
    pmhip_s2_cfg cfg{};
    pmhip_s2_weights w{};
    std::vector<pmhip_layer_weights> layers;
    std::vector<CrossKV> cross;     // per layer, valid after prepare_context
    Workspace ws;
    Switches sw = Switches::from_env();
    bool pos_split = false;         // bf16 mode: position embedding split into hi / lo planes (ws "pos.*")
    std::map<std::string, GraphEntry> graphs;   // captured decode loops, keyed by shape / schedule structure
    hipStream_t capture_stream = nullptr;       // capture never happens on the caller's stream (it may be the NULL stream)
    // small batches: the ViT decode of step t runs on a side stream BESIDE the tower of step t + 1 (fork / join by events, inside
    // the captured graphs too)
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // per-call scalars travel through PINNED host slots (a pageable source makes hipMemcpyAsync stage synchronously);
    // a slot is reused only after the copy that read it has completed
    static constexpr int kParamSlots = 4;
    PmGenParams* params_host = nullptr;
    hipEvent_t params_done[kParamSlots] = {};
    int params_next = 0;
    // device image d complete -> copy stream (one event per image of a call: an event is never re-recorded while a wait on
    // its previous record may still be queued); last D2H complete -> next call
    std::vector<hipEvent_t> img_ready;
    hipEvent_t host_copied = nullptr;
    bool host_copy_pending = false;
    ~pmhip_s2() {
        for (auto& kv : graphs) kv.second.destroy();
        if (capture_stream) (void)hipStreamDestroy(capture_stream);
        if (side_stream) (void)hipStreamDestroy(side_stream);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (params_host) (void)hipHostFree(params_host);
        for (auto e : params_done)
            if (e) (void)hipEventDestroy(e);
        for (auto e : img_ready)
            if (e) (void)hipEventDestroy(e);
        if (host_copied) (void)hipEventDestroy(host_copied);
    }
};

extern "C" int pmhip_s2_create(pmhip_s2** out, int device, int dtype, const pmhip_s2_cfg* cfg, const pmhip_s2_weights* w) {
    PM_REQUIRE(out && cfg && w, "s2_create: null argument");
    PM_REQUIRE(dtype == PMHIP_F32 || dtype == PMHIP_BF16, "s2_create: bad dtype");
    PM_REQUIRE(cfg->tower.dim % 64 == 0 && cfg->tower.hidden_pad % 64 == 0, "s2_create: dim/hidden_pad must be multiples of 64");
    PM_REQUIRE(cfg->embed_dim <= 64 && cfg->embed_dim % 4 == 0, "s2_create: embed_dim must be <= 64 and a multiple of 4");
    PM_TRY(check_dim_head("s2_create", cfg->tower));
    PM_REQUIRE(cfg->context_dim_pad % 64 == 0 && cfg->context_dim_pad >= cfg->context_dim, "s2_create: bad context_dim_pad");
    PM_REQUIRE(w->ctxproj_w || cfg->context_dim == cfg->tower.dim, "s2_create: Identity context_proj needs context_dim == dim");
    PM_REQUIRE(cfg->n_embed % 4 == 0, "s2_create: n_embed must be a multiple of 4");
    auto h = std::make_unique<pmhip_s2>();
    h->device = device; h->dtype = dtype; h->cfg = *cfg; h->w = *w;
    h->layers.assign(w->layers, w->layers + cfg->tower.depth);
    h->w.layers = h->layers.data();
    h->cross.resize(cfg->tower.depth);
    *out = h.release();
    return PMHIP_OK;
}

extern "C" void pmhip_s2_destroy(pmhip_s2* h) {
    if (!h) return;
    (void)hipDeviceSynchronize();
    delete h;
}

namespace {

// context_proj (transformer.py:84-85) and every layer's attn2 to_k / to_v of the projected context
// (attention.py:48-49).  The context is static over a decode loop, so this runs once per loop.
int s2_prepare_context(pmhip_s2* h, const float* context, int L, int B, hipStream_t s) {
    const auto& c = h->cfg;
    const int dim = c.tower.dim, heads = c.tower.heads, dh = dh_of(c.tower), inner = heads * dh;
    const size_t es = dtype_size(h->dtype);
    if (!context) {
        for (auto& ck : h->cross) ck = CrossKV{};
        return PMHIP_OK;
    }
    PM_REQUIRE(L > 0, "s2: context given with L=%d", L);
    const int Mc = B * L, Lp = round_up(L, 64);
    void* cT; void* cp = nullptr;
    WS(h->ws, "ctx.in", (size_t)Mc * c.context_dim_pad * es, cT);
    PM_TRY(pmhip_convert_pad(context, c.context_dim, cT, h->dtype, c.context_dim_pad, Mc, s));
    if (h->w.ctxproj_w) {
        WS(h->ws, "ctx.proj", (size_t)Mc * dim * es, cp);
        PM_TRY(pmhip_gemm(h->dtype, cT, c.context_dim_pad, h->w.ctxproj_w, c.context_dim_pad, nullptr, nullptr, 0, 0, cp, dim,
                          h->dtype, Mc, dim, c.context_dim_pad, s));
    } else {
        cp = cT;
    }
    const size_t per = (size_t)B * heads * Lp * dh * es;
    unsigned char* kv;
    WS(h->ws, "ctx.kv", per * 2 * c.tower.depth, kv);
    float* split = nullptr;
    if (dh != 64) WS(h->ws, "ctx.split", (size_t)Mc * 2 * inner * 4, split);
    const int kinds[2] = {PMHIP_PART_K, PMHIP_PART_V};
    for (int l = 0; l < c.tower.depth; ++l) {
        void* outs[2] = {kv + per * (2 * l), kv + per * (2 * l + 1)};
        const unsigned char* wkv = reinterpret_cast<const unsigned char*>(h->layers[l].wqkv2) + (size_t)inner * dim * es;
        PM_TRY(pmhip_gemm_heads_dh(h->dtype, cp, dim, wkv, dim, Mc, dim, heads, dh, L, Lp, 2, kinds, outs, 1.0f, split, s));
        h->cross[l].k = outs[0]; h->cross[l].vt = outs[1]; h->cross[l].L = L; h->cross[l].Lp = Lp;
    }
    return PMHIP_OK;
}

// token rows (T [M,64]) -> logits fp32 [M,V]  (transformer.py:81-82,87-91)
// use_cross = false: the unconditional branch (context None: every attn2 is a second self-attention, attention.py:47) although a
// context has been prepared -- the second forward of a guided step
// block_stats (optional, n_embed % 64 == 0): the softmax statistics of the logits' 64-column blocks, [M][n_embed/64][2], written by
// the logits GEMM's epilogue for the sampling kernel (gemm_common.h GemmParams::block_stats)
int s2_tower(pmhip_s2* h, const void* tp, int B, float* logits, hipStream_t s, bool use_cross = true, float* block_stats = nullptr) {
    const auto& c = h->cfg;
    const int M = B * c.tokens, dim = c.tower.dim;
    TowerBufs tb;
    PM_TRY(alloc_tower(h->ws, h->sw, "s2", h->dtype, c.tower, B, c.tokens, tb, s));
    ResSrc pos;
    PM_TRY(pos_source(h->ws, "pos", tb.hilo, h->w.pos, c.tokens, dim, h->pos_split, pos, s));
    PM_TRY(residual_gemm(h->dtype, tb, tp, 64, h->w.tokproj_w, 64, h->w.tokproj_b, pos, M, dim, 64, s));
    for (int l = 0; l < c.tower.depth; ++l)
        PM_TRY(layer_forward(h->dtype, h->layers[l], c.tower, tb, B, c.tokens, true, use_cross ? &h->cross[l] : nullptr, s));
    if (tb.fold && h->w.logits_wf && fold_shape_ok(c.tokens, c.n_embed, dim)) {
        PM_TRY(tower_coef(tb, M, dim, s));
        const int step = fold_rows(tb, M, c.tokens, dim);
        for (int m0 = 0; m0 < M; m0 += step) {
            const pmhip_lnfold ln = fold_desc(tb, m0, dim, h->w.logits_c, h->w.logits_d);   // the final norm folded into to_logits
            const void* a = reinterpret_cast<const unsigned char*>(tb.xh) + (size_t)m0 * dim * 2;
            if (block_stats)
                PM_TRY(pmhip_gemm_softmax_stats(h->dtype, a, dim, h->w.logits_wf, dim, h->w.logits_b, logits + (size_t)m0 * c.n_embed, c.n_embed,
                                                std::min(step, M - m0), c.n_embed, dim, &ln, block_stats + (size_t)m0 * (c.n_embed / 64) * 2, s));
            else
                PM_TRY(pmhip_gemm_ln(h->dtype, a, dim, h->w.logits_wf, dim, h->w.logits_b, logits + (size_t)m0 * c.n_embed, c.n_embed, PMHIP_F32,
                                     std::min(step, M - m0), c.n_embed, dim, &ln, s));
        }
        return PMHIP_OK;
    }
    PM_TRY(tower_layernorm(h->dtype, tb, h->w.norm_g, h->w.norm_b, M, dim, s));
    if (block_stats)
        return pmhip_gemm_softmax_stats(h->dtype, tb.y, dim, h->w.logits_w, dim, h->w.logits_b, logits, c.n_embed, M, c.n_embed, dim, nullptr,
                                        block_stats, s);
    return pmhip_gemm(h->dtype, tb.y, dim, h->w.logits_w, dim, h->w.logits_b, nullptr, 0, 0, logits, c.n_embed, PMHIP_F32, M,
                      c.n_embed, dim, s);
}

// Pipeline.sample after the context is prepared (generate.py:161-179)
// guidance != nullptr: the step's logits are uncond + *guidance * (cond - uncond), uncond = the same tower without the context
// (the branch the reference trains by dropping the text, utils/trainer.py:379,387-388); everything after the logits is unchanged
// The step in two halves, so that a caller can put something between the tower and the sampling (the small-batch loop joins the
// previous step's decode there).  step_tower: ids2tokens + the tower(s) -> logits.  step_tail: sampling, the optional decode,
// the optional copies, re-masking.
int step_tower(pmhip_s2* s2, const int64_t* ids, int B, hipStream_t s, const float* guidance) {
    const auto& c = s2->cfg;
    const int M = B * c.tokens;
    void* tp; float* logits;
    WS(s2->ws, "s2.tok", (size_t)M * 64 * dtype_size(s2->dtype), tp);
    WS(s2->ws, "s2.logits", (size_t)M * c.n_embed * 4, logits);
    // ids2tokens: lookup in cat(raw codebook, mask_token) (generate.py:148-157)
    PM_TRY(pmhip_embed_rows(s2->w.tok_table, ids, tp, s2->dtype, 64, M, c.n_embed + 1, c.embed_dim, s));
    // softmax statistics of the logits' 64-column blocks for the sampling kernel: from the logits GEMM, or -- guided -- from the
    // combination, which produces the logits that are sampled
    float* lstats = nullptr;
    if (c.n_embed % 64 == 0 && s2->sw.logits_stats) WS(s2->ws, "s2.lstats", (size_t)M * (c.n_embed / 64) * 8, lstats);
    PM_TRY(s2_tower(s2, tp, B, logits, s, true, guidance ? nullptr : lstats));
    if (guidance) {
        float* uncond;
        WS(s2->ws, "s2.logits_u", (size_t)M * c.n_embed * 4, uncond);
        PM_TRY(s2_tower(s2, tp, B, uncond, s, false));
        if (lstats) PM_TRY(pmhip_guidance_combine_stats(logits, uncond, *guidance, logits, (size_t)M * c.n_embed, lstats, s));
        else PM_TRY(pmhip_guidance_combine(logits, uncond, *guidance, logits, (size_t)M * c.n_embed, s));
    }
    return PMHIP_OK;
}

// the image of the predictions the last step_tail left in the handle's `s2.pred` (decoded from pred at ALL positions, generate.py:165)
int decode_pred(pmhip_s2* s2, pmhip_vqgan* vq, int B, float* img_out, hipStream_t s) {
    PM_REQUIRE(vq, "pipeline_sample: img_out requested without a vqgan handle");
    int64_t* pred;
    WS(s2->ws, "s2.pred", (size_t)B * s2->cfg.tokens * 8, pred);
    return vq_decode_indices(vq, pred, B, img_out, s);
}

int step_tail(pmhip_s2* s2, pmhip_vqgan* vq, int64_t* ids, int B, int topk, float temperature, int num_mask, const float* noise,
              uint64_t seed, uint32_t step, uint64_t image_base, float* img_out, int64_t* pred_out, float* score_out, hipStream_t s,
              const PmGenParams* gp) {
    const auto& c = s2->cfg;
    const int M = B * c.tokens;
    float* logits; int64_t* pred; float* score;
    WS(s2->ws, "s2.logits", (size_t)M * c.n_embed * 4, logits);
    WS(s2->ws, "s2.pred", (size_t)M * 8, pred);
    WS(s2->ws, "s2.score", (size_t)M * 4, score);
    float* lstats = nullptr;                                 // step_tower filled them (same condition, same workspace entry)
    if (c.n_embed % 64 == 0 && s2->sw.logits_stats) WS(s2->ws, "s2.lstats", (size_t)M * (c.n_embed / 64) * 8, lstats);
    PM_TRY(pm_sample_rows(logits, c.n_embed, lstats, ids, (int64_t)c.n_embed, topk, temperature, noise, seed, step,
                          image_base * (uint64_t)c.tokens, pred, ids, score, M, c.n_embed, gp, s));
    if (img_out) PM_TRY(decode_pred(s2, vq, B, img_out, s));
    if (pred_out) PM_HIP(hipMemcpyAsync(pred_out, pred, (size_t)M * 8, hipMemcpyDeviceToDevice, s));
    if (score_out) PM_HIP(hipMemcpyAsync(score_out, score, (size_t)M * 4, hipMemcpyDeviceToDevice, s));
    return pm_remask(ids, score, num_mask, (int64_t)c.n_embed, B, c.tokens, gp, (int)step, s);
}

int sample_step(pmhip_s2* s2, pmhip_vqgan* vq, int64_t* ids, int B, int topk, float temperature, int num_mask,
                const float* noise, uint64_t seed, uint32_t step, uint64_t image_base, float* img_out, int64_t* pred_out,
                float* score_out, hipStream_t s, const PmGenParams* gp = nullptr, const float* guidance = nullptr) {
    PM_TRY(step_tower(s2, ids, B, s, guidance));
    return step_tail(s2, vq, ids, B, topk, temperature, num_mask, noise, seed, step, image_base, img_out, pred_out, score_out, s, gp);
}

}  // namespace

extern "C" int pmhip_s2_forward(pmhip_s2* h, const float* tokens, const float* context, int L, int B, float* logits_out,
                                pmhip_stream stream) {
    PM_REQUIRE(h && tokens && logits_out && B > 0, "s2_forward: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int M = B * h->cfg.tokens;
    PM_TRY(s2_prepare_context(h, context, L, B, s));
    void* tp;
    WS(h->ws, "s2.tok", (size_t)M * 64 * dtype_size(h->dtype), tp);
    PM_TRY(pmhip_convert_pad(tokens, h->cfg.embed_dim, tp, h->dtype, 64, M, s));
    return s2_tower(h, tp, B, logits_out, s);
}

extern "C" int pmhip_pipeline_sample(pmhip_s2* s2, pmhip_vqgan* vq, int64_t* ids, const float* context, int L, int B,
                                     int topk, float temperature, int num_mask, const float* noise, uint64_t seed,
                                     uint32_t step, uint64_t image_base, float* img_out, int64_t* pred_out,
                                     float* score_out, pmhip_stream stream) {
    PM_REQUIRE(s2 && ids && B > 0, "pipeline_sample: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    PM_TRY(s2_prepare_context(s2, context, L, B, s));
    return sample_step(s2, vq, ids, B, topk, temperature, num_mask, noise, seed, step, image_base, img_out, pred_out,
                       score_out, s);
}

extern "C" int pmhip_pipeline_sample_guided(pmhip_s2* s2, pmhip_vqgan* vq, int64_t* ids, const float* context, int L, int B,
                                            int topk, float temperature, int num_mask, const float* noise, uint64_t seed,
                                            uint32_t step, uint64_t image_base, float* img_out, int64_t* pred_out,
                                            float* score_out, float guidance_scale, pmhip_stream stream) {
    PM_REQUIRE(s2 && ids && B > 0, "pipeline_sample_guided: bad arguments");
    PM_REQUIRE(context && L > 0, "pipeline_sample_guided: guidance needs a context (context NULL IS the unconditional branch)");
    hipStream_t s = (hipStream_t)stream;
    PM_TRY(s2_prepare_context(s2, context, L, B, s));
    return sample_step(s2, vq, ids, B, topk, temperature, num_mask, noise, seed, step, image_base, img_out, pred_out,
                       score_out, s, nullptr, &guidance_scale);
}

// process-wide: the runtime reads AMD_DIRECT_DISPATCH once when it starts, so does this
static bool direct_dispatch_off() {
    static const bool off = [] { const char* e = getenv("AMD_DIRECT_DISPATCH"); return e && atoi(e) == 0; }();
    return off;
}

static int pipeline_generate(pmhip_s2* s2, pmhip_vqgan* vq, int64_t* ids, const float* context, int L, int B,
                             int T, const float* temps_host, const int* nmask_host,
                             const unsigned char* decode_host, int topk, uint64_t seed, uint64_t image_base,
                             float* imgs_out, int use_graph, pmhip_stream stream, float* imgs_host,
                             size_t host_stride, pmhip_stream copy_stream, const float* guidance) {
    PM_REQUIRE(s2 && ids && B > 0 && T > 0 && temps_host && nmask_host, "pipeline_generate: bad arguments");
    PM_REQUIRE(!guidance || (context && L > 0), "pipeline_generate_guided: guidance needs a context (context NULL IS the unconditional branch)");
    hipStream_t s = (hipStream_t)stream;
    hipStream_t cs = copy_stream ? (hipStream_t)copy_stream : s;
    PM_TRY(s2_prepare_context(s2, context, L, B, s));         // context projection + cross K/V: once per loop, eager
    size_t img_elems = 0;
    if (vq) img_elems = (size_t)B * vq->cfg.channels * vq->cfg.image_size * vq->cfg.image_size;
    int n_dec = 0;
    for (int t = 0; t < T; ++t) n_dec += (decode_host && decode_host[t]) ? 1 : 0;
    PM_REQUIRE(n_dec == 0 || (vq && (imgs_out || imgs_host)), "pipeline_generate: decode requested without vqgan / an image destination");
    PM_REQUIRE(!imgs_host || host_stride >= img_elems, "pipeline_generate: host_stride smaller than one image batch");
    // AMD_DIRECT_DISPATCH=0 (the runtime mode without the busy-polling helper thread): hipGraph replay is broken there on ROCm 7.2
    // -- tools/hwtests/graph_dispatch_mode.hip, 39 of 40 replays of a chain of dependent kernels wrong with none of this library's
    // code involved -- so the loop stays eager in that mode whatever the caller asked for (same results, bit for bit)
    // pmhip_s2_switches reports the downgrade (bit 5) so that a caller can tell which mode ran
    const bool graph = (use_graph & PMHIP_GENERATE_GRAPH) && !g_pm_timing_on.load() && T <= PM_MAX_STEPS && !direct_dispatch_off();

    if (imgs_host) {
        // PMHIP_BLOCKING_WAIT=1 (read when the handle is created): the lane's host thread SLEEPS in hipEventSynchronize while its
        // segment runs instead of spinning -- frees a core per lane on a host that is short of them, at 0.6 ms of wake-up latency
        // per saved image (measured: the drop-in generate() 151 vs 141 ms per call), hence off by default
        const unsigned evflags = hipEventDisableTiming | (s2->sw.blocking_wait ? hipEventBlockingSync : 0u);
        if (!s2->host_copied) PM_HIP(hipEventCreateWithFlags(&s2->host_copied, evflags));
        while ((int)s2->img_ready.size() < n_dec) {
            hipEvent_t e;
            PM_HIP(hipEventCreateWithFlags(&e, evflags));
            s2->img_ready.push_back(e);
        }
    }
    // images are produced into a handle-owned buffer when a graph bakes the pointer in or when the caller only wants the
    // host copy; that buffer must not be overwritten while the previous call's last device-to-host copy still reads it
    float* gimgs = nullptr;
    if (n_dec && (graph || !imgs_out)) {
        WS(s2->ws, "gen.imgs", (size_t)n_dec * img_elems * 4 + 16, gimgs);
        if (s2->host_copy_pending) { PM_HIP(hipStreamWaitEvent(s, s2->host_copied, 0)); s2->host_copy_pending = false; }
    }
    // Image d is complete on `s` after the step that decodes it.  Device destination: a copy on `s`.  Host destination: the
    // copy must run on the copy stream UNDER the following steps, but it is not made to wait there by a cross-stream event:
    // a barrier packet parked at the head of the copy queue until a whole segment has run starves the other lane's queue
    // on this part (measured: two lanes with such waits run one after the other, 164 vs 138 ms per call).  Instead the HOST
    // paces the loop, like the reference's blocking `img.cpu()` per saved step (generate.py:195-196): once the next
    // segment is queued it waits for image d's event and enqueues a copy that can start at once.  The caller runs
    // concurrent lanes from one thread each (the ctypes call drops the GIL).
    int pending = -1;                                         // decoded image whose host copy has not been enqueued yet
    const float* pending_src = nullptr;
    auto flush_pending = [&]() -> int {
        if (pending < 0) return PMHIP_OK;
        if (cs != s) PM_HIP(hipEventSynchronize(s2->img_ready[pending]));
        PM_HIP(hipMemcpyAsync(imgs_host + (size_t)pending * host_stride, pending_src, img_elems * 4, hipMemcpyDeviceToHost, cs));
        if (pending == n_dec - 1 && gimgs) {
            PM_HIP(hipEventRecord(s2->host_copied, cs));
            s2->host_copy_pending = cs != s;
        }
        pending = -1;
        return PMHIP_OK;
    };
    auto deliver = [&](int d, const float* src) -> int {
        if (imgs_out && src != imgs_out + (size_t)d * img_elems)
            PM_HIP(hipMemcpyAsync(imgs_out + (size_t)d * img_elems, src, img_elems * 4, hipMemcpyDeviceToDevice, s));
        if (imgs_host) {
            if (cs != s) PM_HIP(hipEventRecord(s2->img_ready[d], s));
            pending = d;
            pending_src = src;
        }
        return PMHIP_OK;
    };

    if (!graph) {
        int d = 0;
        for (int t = 0; t < T; ++t) {
            const bool dec = decode_host && decode_host[t];
            float* img = !dec ? nullptr : (gimgs ? gimgs : imgs_out) + (size_t)d * img_elems;
            PM_TRY(sample_step(s2, vq, ids, B, topk, temps_host[t], nmask_host[t], nullptr, seed, (uint32_t)t, image_base, img,
                               nullptr, nullptr, s, nullptr, guidance));
            PM_TRY(flush_pending());                           // the previous image, now that one more step is queued behind it
            if (dec) PM_TRY(deliver(d++, img));
        }
        return flush_pending();
    }

    // ---- hipGraph path.  The T-step loop is a chain of graphs, one per SEGMENT (the steps up to and including a decoded
    // one), whose kernels read the per-call scalars (temperatures, mask counts, seed, row base) from a device parameter
    // block and whose ids / image pointers are handle-owned buffers, so the same executable graphs serve every call with
    // this structure; between two segments the finished image starts its way to the host.
    const size_t ids_bytes = (size_t)B * s2->cfg.tokens * 8;
    int64_t* gids; PmGenParams* gparams;
    WS(s2->ws, "gen.ids", ids_bytes, gids);
    WS(s2->ws, "gen.params", sizeof(PmGenParams), gparams);
    if (!s2->params_host) {
        PM_HIP(hipHostMalloc((void**)&s2->params_host, sizeof(PmGenParams) * pmhip_s2::kParamSlots, hipHostMallocDefault));
        for (auto& e : s2->params_done) PM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int slot = s2->params_next;
    s2->params_next = (slot + 1) % pmhip_s2::kParamSlots;
    PM_HIP(hipEventSynchronize(s2->params_done[slot]));       // the copy that last read this slot (no-op when never recorded)
    PmGenParams& hp = s2->params_host[slot];
    hp.seed = seed;
    hp.row_base = image_base * (uint64_t)s2->cfg.tokens;
    for (int t = 0; t < T; ++t) { hp.temps[t] = temps_host[t]; hp.nmask[t] = nmask_host[t]; }
    {
        void* hp_dev = nullptr;                               // the pinned slot through its device alias
        PM_HIP(hipHostGetDevicePointer(&hp_dev, &hp, 0));
        PM_TRY(copy16_async(gparams, hp_dev, sizeof hp, s));
    }
    PM_HIP(hipEventRecord(s2->params_done[slot], s));
    PM_TRY(copy16_async(gids, ids, ids_bytes, s));

    std::string key = "B" + std::to_string(B) + "T" + std::to_string(T) + "k" + std::to_string(topk) + "L" +
                      std::to_string(context ? L : 0) + "v" + std::to_string(vq ? vq->uid : 0) + "f" +
                      std::to_string(s2->sw.key() * 16 + (vq ? vq->sw.key() : 0));
    if (guidance) {                                           // the scale is a kernel argument of the captured combine: one graph per value
        unsigned bits;
        memcpy(&bits, guidance, 4);
        key += "g" + std::to_string(bits);
    }
    key += "d";
    for (int t = 0; t < T; ++t) key += (decode_host && decode_host[t]) ? '1' : '0';

    // A unit = one executable graph.  Normally a unit is a SEGMENT [t0, t1) (t1 - 1 is a decoded step, or the end of the loop) whose
    // last step decodes in place and whose image is complete when the unit is.  Small batches (B * tokens <= overlap_rows: every
    // kernel is a few dozen workgroups on 256 CUs and the loop is one long dependent chain) DEFER the decode instead: the ViT decode
    // of a segment's last step opens the NEXT unit on a side stream, beside that unit's first tower pass (which only needs the ids),
    // and is joined before the first sampling kernel overwrites the predictions it reads; a last unit without steps decodes the
    // final image.  Same kernels, same inputs: bit-identical images, one unit later.
    struct Unit { int t0, t1, decode_first, delivers; bool decode_inline; };
    // (not when the caller runs other lanes beside this one -- PMHIP_GENERATE_CONCURRENT_LANES: the chip is full then, and a third
    // and fourth stream of kernels costs 5-17 % at 15-16 images per lane, profiles/r05_g_*)
    const bool overlap = vq && n_dec > 0 && s2->sw.overlap_rows > 0 && (long long)B * s2->cfg.tokens <= s2->sw.overlap_rows &&
                         !(use_graph & PMHIP_GENERATE_CONCURRENT_LANES);
    std::vector<Unit> units;
    {
        int d = 0, pend = -1;
        for (int t = 0, t0 = 0; t < T; ++t) {
            const bool dec = decode_host && decode_host[t];
            if (!dec && t != T - 1) continue;
            if (overlap) { units.push_back({t0, t + 1, pend, pend, false}); pend = dec ? d++ : -1; }
            else units.push_back({t0, t + 1, -1, dec ? d++ : -1, true});
            t0 = t + 1;
        }
        if (pend >= 0) units.push_back({T, T, pend, pend, false});
    }
    key += overlap ? "o1" : "o0";
    GraphEntry& ge = s2->graphs[key];
    if (overlap && !s2->side_stream) {
        PM_HIP(hipStreamCreateWithFlags(&s2->side_stream, hipStreamNonBlocking));
        PM_HIP(hipEventCreateWithFlags(&s2->ev_fork, hipEventDisableTiming));
        PM_HIP(hipEventCreateWithFlags(&s2->ev_join, hipEventDisableTiming));
    }

    auto run_unit = [&](hipStream_t on, const Unit& u) -> int {
        bool need_join = false;
        if (u.decode_first >= 0) {
            float* img = gimgs + (size_t)u.decode_first * img_elems;
            if (u.t1 > u.t0) {                                // fork: the pending decode runs beside this unit's first tower pass
                PM_HIP(hipEventRecord(s2->ev_fork, on));
                PM_HIP(hipStreamWaitEvent(s2->side_stream, s2->ev_fork, 0));
                PM_TRY(decode_pred(s2, vq, B, img, s2->side_stream));
                PM_HIP(hipEventRecord(s2->ev_join, s2->side_stream));
                need_join = true;
            } else {
                PM_TRY(decode_pred(s2, vq, B, img, on));
            }
        }
        for (int t = u.t0; t < u.t1; ++t) {
            PM_TRY(step_tower(s2, gids, B, on, guidance));
            if (need_join) { PM_HIP(hipStreamWaitEvent(on, s2->ev_join, 0)); need_join = false; }   // before `s2.pred` is overwritten
            float* img = (u.decode_inline && decode_host && decode_host[t]) ? gimgs + (size_t)u.delivers * img_elems : nullptr;
            PM_TRY(step_tail(s2, vq, gids, B, topk, 0.f, 0, nullptr, 0, (uint32_t)t, 0, img, nullptr, nullptr, on, gparams));
        }
        return PMHIP_OK;
    };

    if (!ge.warmed) {
        for (const Unit& u : units) {                          // eager once: sizes every workspace buffer
            PM_TRY(run_unit(s, u));
            PM_TRY(flush_pending());
            if (u.delivers >= 0) PM_TRY(deliver(u.delivers, gimgs + (size_t)u.delivers * img_elems));
        }
        ge.warmed = true;
    } else {
        // a workspace buffer of either handle was reallocated since the capture (a later call with a larger batch, a
        // longer context, a direct encode/decode on the shared vqgan handle ...): the graphs' pointers are stale
        if (!ge.segs.empty() && (ge.s2_gen != s2->ws.gen || (vq && ge.vq_gen != vq->ws.gen))) {
            PM_HIP(hipStreamSynchronize(s));                  // an earlier replay may still be running
            ge.destroy();
        }
        if (ge.segs.empty()) {
            if (!s2->capture_stream) PM_HIP(hipStreamCreateWithFlags(&s2->capture_stream, hipStreamNonBlocking));
            hipStream_t cap = s2->capture_stream;
            for (const Unit& u : units) {
                hipGraph_t g = nullptr;
                hipGraphExec_t exec = nullptr;
                s2->ws.frozen = true;
                if (vq) vq->ws.frozen = true;
                hipError_t rc = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
                int step_rc = PMHIP_OK;
                if (rc == hipSuccess) {
                    step_rc = run_unit(cap, u);              // records only: nothing executes during capture
                    rc = hipStreamEndCapture(cap, &g);
                }
                s2->ws.frozen = false;
                if (vq) vq->ws.frozen = false;
                if (step_rc == PMHIP_OK && rc == hipSuccess) rc = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
                if (g) (void)hipGraphDestroy(g);
                if (step_rc != PMHIP_OK || rc != hipSuccess) {
                    ge.destroy();
                    if (step_rc != PMHIP_OK) return step_rc;
                    PM_HIP(rc);
                }
                ge.segs.push_back(exec);
            }
            ge.s2_gen = s2->ws.gen;
            ge.vq_gen = vq ? vq->ws.gen : 0;
        }
        for (size_t i = 0; i < units.size(); ++i) {
            PM_HIP(hipGraphLaunch(ge.segs[i], s));
            PM_TRY(flush_pending());
            if (units[i].delivers >= 0) PM_TRY(deliver(units[i].delivers, gimgs + (size_t)units[i].delivers * img_elems));
        }
    }
    PM_TRY(copy16_async(ids, gids, ids_bytes, s));
    return flush_pending();
}

extern "C" int pmhip_pipeline_generate(pmhip_s2* s2, pmhip_vqgan* vq, int64_t* ids, const float* context, int L, int B,
                                       int T, const float* temps_host, const int* nmask_host,
                                       const unsigned char* decode_host, int topk, uint64_t seed, uint64_t image_base,
                                       float* imgs_out, int use_graph, pmhip_stream stream, float* imgs_host,
                                       size_t host_stride, pmhip_stream copy_stream) {
    return pipeline_generate(s2, vq, ids, context, L, B, T, temps_host, nmask_host, decode_host, topk, seed, image_base, imgs_out,
                             use_graph, stream, imgs_host, host_stride, copy_stream, nullptr);
}

extern "C" int pmhip_pipeline_generate_guided(pmhip_s2* s2, pmhip_vqgan* vq, int64_t* ids, const float* context, int L, int B,
                                              int T, const float* temps_host, const int* nmask_host,
                                              const unsigned char* decode_host, int topk, uint64_t seed, uint64_t image_base,
                                              float* imgs_out, int use_graph, pmhip_stream stream, float* imgs_host,
                                              size_t host_stride, pmhip_stream copy_stream, float guidance_scale) {
    return pipeline_generate(s2, vq, ids, context, L, B, T, temps_host, nmask_host, decode_host, topk, seed, image_base, imgs_out,
                             use_graph, stream, imgs_host, host_stride, copy_stream, &guidance_scale);
}

// the PMHIP_* switches a handle latched when it was created (bit 0 fold, 1 hilo, 2 stats, 3 center, 4 blocking_wait)
extern "C" int pmhip_s2_switches(const pmhip_s2* h) {
    return h ? h->sw.key() + (h->sw.blocking_wait ? 16 : 0) + (direct_dispatch_off() ? 32 : 0) : -1;
}
extern "C" int pmhip_vqgan_switches(const pmhip_vqgan* h) { return h ? h->sw.key() + (h->sw.blocking_wait ? 16 : 0) : -1; }
