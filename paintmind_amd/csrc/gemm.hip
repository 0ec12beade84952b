// NT GEMM family for gfx950: out = A[M,K] . W[N,K]^T with fused epilogues (128x128 tile kernel + entry points).
//
// Both operands are K-contiguous, which is exactly the MFMA fragment shape (16 B of k per lane),
// so neither is transposed anywhere.  Tile 128(m) x 128(n) x 128 bytes of k per step
// (64 bf16 / 32 f32), 256 threads = 4 waves as 2(m) x 2(n), each wave 64x64 = 4x4 MFMA tiles.
// Staging is direct-to-LDS (global_load_lds_dwordx4): the LDS image is lane-linear, so the
// bank-conflict XOR swizzle is applied to the per-lane SOURCE address and again on the ds_read
// (cdna_hip_programming.md 5.4 rule 21).  Two LDS stages; the loads of step t+1 are in flight while
// step t is multiplied.  Epilogue: gemm_common.h.  The bf16 SwiGLU / head-split / logits shapes with
// M, N multiples of 256 go to the phase-staggered 256x256 kernel in gemm256.hip instead.
//
// Replaces torch.nn.Linear / aten::addmm at: modules/attention.py:46-49,59; modules/mlp.py:27-31;
// stage1/vqmodel.py:23,28; stage1/layers.py:107 (patch-embed conv as GEMM),149;
// stage2/transformer.py:81,85,91 (reference paths).
#include <stdlib.h>

#include "gemm_common.h"

using namespace pmgemm;

// gemm256.hip
int pm_gemm256_supported(const GemmParams& p, int dtype, int epi, int out_dtype);
int pm_gemm256_launch(const GemmParams& p, int epi, int out_dtype, hipStream_t s);
// gemm2b.hip
int pm_gemm2b_supported(const GemmParams& p, int dtype, int epi, int out_dtype);
int pm_gemm2b_launch(const GemmParams& p, int epi, int out_dtype, hipStream_t s);

namespace {

constexpr int BM = 128, BN = 128;
constexpr int TILE_BYTES = 128 * ROWB;             // one operand tile, 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;        // A tile + W tile
constexpr int THREADS = 256;

template <typename T>
__device__ __forceinline__ void stage_tile(const T* __restrict__ base, int ld, int row0, int rows_total,
                                           int k0, unsigned char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int chunk = wave * 4 + i;                    // 1 KiB = 8 rows x 128 B
        const int r = chunk * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ (r & 7);             // inverse swizzle on the source
        int gr = row0 + r;
        gr = gr < rows_total ? gr : rows_total - 1;        // clamp: guarded rows are never stored
        const unsigned char* src =
            reinterpret_cast<const unsigned char*>(base + (size_t)gr * ld + k0) + slot * 16;
        glds16(src, lds_tile + chunk * 1024);
    }
}

template <typename T, int EPI, typename OutT>
__global__ __launch_bounds__(THREADS) void gemm_nt_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    int tm, tn;
    tile_of_block(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, 8, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const T* A = reinterpret_cast<const T*>(p.A);
    const T* W = reinterpret_cast<const T*>(p.W);
    constexpr int KSTEP = ROWB / (int)sizeof(T);
    const int nk = p.K / KSTEP;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // The residual tile is fetched BEFORE the K loop into registers, in the epilogue's store layout (its HBM
    // latency would otherwise sit, exposed, between the last MFMA and the stores).
    constexpr bool RPRE = (EPI == EPI_STD && sizeof(OutT) == 4);
    float4 rpre[RPRE ? 16 : 1];
    if constexpr (RPRE) {
        if (p.residual) {
            const int ncol = n0 + wn * 64 + (lane & 15) * 4;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    int mm = m0 + wm * 64 + mi * 16 + it * 4 + (lane >> 4);
                    mm = mm < p.M ? mm : p.M - 1;
                    const int nc = ncol < p.N ? ncol : 0;
                    rpre[mi * 4 + it] = *reinterpret_cast<const float4*>(p.residual + (size_t)(mm % p.res_rows) * p.ldr + nc);
                }
        }
    }

    stage_tile<T>(A, p.lda, m0, p.M, 0, lds, wave, lane);
    stage_tile<T>(W, p.ldw, n0, p.N, 0, lds + TILE_BYTES, wave, lane);

    for (int kt = 0; kt < nk; ++kt) {
        // this wave's DMA for step kt has landed; after the barrier so has everybody's, and every
        // wave has finished reading the other stage (its ds_reads fed MFMAs already issued)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned char* cur = lds + (kt & 1) * STAGE_BYTES;
        if (kt + 1 < nk) {
            unsigned char* nxt = lds + ((kt + 1) & 1) * STAGE_BYTES;
            stage_tile<T>(A, p.lda, m0, p.M, (kt + 1) * KSTEP, nxt, wave, lane);
            stage_tile<T>(W, p.ldw, n0, p.N, (kt + 1) * KSTEP, nxt + TILE_BYTES, wave, lane);
        }
        const unsigned char* At = cur;
        const unsigned char* Wt = cur + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 af[4], wf[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                af[f] = read_frag(At, wm * 64 + f * 16 + l15, kk * 4 + g);
                wf[f] = read_frag(Wt, wn * 64 + f * 16 + l15, kk * 4 + g);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) Mma<T>::run(acc[mi][ni], wf[ni], af[mi]);
        }
    }
    // the LDS stage that is idle after the K loop (stage nk&1: its last reads finished before the final barrier)
    // is the epilogue's transposition buffer
    unsigned char* eraw = lds + (nk & 1) * STAGE_BYTES + wave * EPI_WAVE_BYTES;
    if constexpr (EPI == EPI_STD && sizeof(OutT) == 2 && sizeof(T) == 2) {
        if (p.out_lo) { wave_epilogue<EPI, OutT, 4, 1, false, 2>(p, acc, eraw, m0 + wm * 64, n0 + wn * 64, lane, rpre); return; }   // bf16 hi/lo residual stream
    }
    wave_epilogue<EPI, OutT, 4>(p, acc, eraw, m0 + wm * 64, n0 + wn * 64, lane, rpre);
}

template <typename T, int EPI, typename OutT>
int launch(const GemmParams& p, hipStream_t s) {
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    PmTimer tm(gemm_family(p, EPI), s);
    hipLaunchKernelGGL((gemm_nt_kernel<T, EPI, OutT>), dim3(tiles), dim3(THREADS), 0, s, p);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

int g_use256 = -1;      // development switch PMHIP_GEMM256=0 disables the 256x256 kernel

bool use256(const GemmParams& p, int dtype, int epi, int out_dtype) {
    if (g_use256 < 0) {
        const char* e = getenv("PMHIP_GEMM256");
        g_use256 = e ? atoi(e) : 1;
    }
    return g_use256 && pm_gemm256_supported(p, dtype, epi, out_dtype);
}

int g_use2b = -1;       // PMHIP_GEMM2B: 0 = never, 1 = where it wins (default), 2 = wherever it is supported (development)

// The two-workgroups-per-CU kernel (gemm2b.hip) takes the residual GEMMs with a short K loop (attention out-proj,
// K = inner): they are HBM-bound (fp32 residual in, fp32 out) and it streams them at ~4.5 TB/s where the 128x128
// kernel reaches 3.6.  Everything else measured equal or slower than gemm256.hip (tools/gemm_bench.py).
bool use2b(const GemmParams& p, int dtype, int epi, int out_dtype) {
    if (g_use2b < 0) {
        const char* e = getenv("PMHIP_GEMM2B");
        g_use2b = e ? atoi(e) : 1;
    }
    if (!g_use2b || !pm_gemm2b_supported(p, dtype, epi, out_dtype)) return false;
    if (g_use2b >= 2) return true;
    return epi == EPI_STD && p.residual && !pm_gemm256_supported(p, dtype, epi, out_dtype);
}

int check_common(const GemmParams& p, int dtype) {
    PM_REQUIRE(dtype == PMHIP_F32 || dtype == PMHIP_BF16, "gemm: bad dtype %d", dtype);
    PM_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    PM_REQUIRE(p.K % 64 == 0, "gemm: K=%d must be a multiple of 64 (pad on the host)", p.K);
    PM_REQUIRE(p.lda % 8 == 0 && p.ldw % 8 == 0, "gemm: lda/ldw must be multiples of 8 elements");
    PM_REQUIRE(p.A && p.W, "gemm: null operand");
    return PMHIP_OK;
}

}  // namespace

namespace {

// LayerNorm fold (gemm_common.h): checks shared by the three consumer entry points; fills the consumer fields
int set_lnfold(GemmParams& p, const pmhip_lnfold* ln, int dtype, int epi, int out_dtype) {
    if (!ln) return PMHIP_OK;
    PM_REQUIRE(ln->coef && ln->c && ln->d, "gemm_ln: null fold pointer");
    PM_REQUIRE(dtype == PMHIP_BF16, "gemm_ln: the LayerNorm fold exists in bf16 mode only");
    PM_REQUIRE(p.K % 128 == 0, "gemm_ln: K=%d must be a multiple of 128", p.K);
    p.ln_c = ln->c; p.ln_d = ln->d; p.ln_coef = ln->coef;
    // any tile count: whether a LayerNorm is folded must not depend on the batch size (see pmhip_lnfold_supported)
    PM_REQUIRE(p.M % 256 == 0 && p.N % 256 == 0 && (unsigned long long)p.M * p.lda * 2 < (1ull << 31) &&
               (unsigned long long)p.N * p.ldw * 2 < (1ull << 31),
               "gemm_ln: shape M=%d N=%d K=%d is not served by the 256x256 kernel (M, N multiples of 256)", p.M, p.N, p.K);
    return PMHIP_OK;
}

int gemm_impl(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias,
                          const float* residual, int ldr, int res_rows, void* out, int ldo, int out_dtype,
                          int M, int N, int K, const pmhip_lnfold* ln, pmhip_stream stream) {
    GemmParams p{};
    p.A = A; p.W = W; p.bias = bias; p.residual = residual; p.out = out;
    p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.res_rows = res_rows > 0 ? res_rows : M; p.ldo = ldo;
    p.M = M; p.N = N; p.K = K;
    PM_TRY(check_common(p, dtype));
    PM_REQUIRE(out, "gemm: null out");
    PM_REQUIRE(N % 4 == 0 && ldo % 4 == 0, "gemm: N=%d and ldo=%d must be multiples of 4", N, ldo);
    PM_REQUIRE(!residual || ldr % 4 == 0, "gemm: ldr must be a multiple of 4");
    PM_REQUIRE(!residual || out_dtype == PMHIP_F32, "gemm: a residual needs an f32 output (the residual stream is f32)");
    PM_REQUIRE(out_dtype == PMHIP_F32 || out_dtype == dtype, "gemm: out dtype must be f32 or the compute dtype");
    PM_REQUIRE(out_dtype == PMHIP_F32 || ldo % 8 == 0, "gemm: bf16 output needs ldo to be a multiple of 8");
    PM_TRY(set_lnfold(p, ln, dtype, EPI_STD, out_dtype));
    hipStream_t s = (hipStream_t)stream;
    if (ln) return pm_gemm256_launch(p, EPI_STD, out_dtype, s);
    if (use2b(p, dtype, EPI_STD, out_dtype)) return pm_gemm2b_launch(p, EPI_STD, out_dtype, s);
    if (use256(p, dtype, EPI_STD, out_dtype)) return pm_gemm256_launch(p, EPI_STD, out_dtype, s);
    if (dtype == PMHIP_F32) return launch<float, EPI_STD, float>(p, s);
    if (out_dtype == PMHIP_F32) return launch<bf16_t, EPI_STD, float>(p, s);
    return launch<bf16_t, EPI_STD, bf16_t>(p, s);
}

}  // namespace

extern "C" int pmhip_gemm(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias,
                          const float* residual, int ldr, int res_rows, void* out, int ldo, int out_dtype,
                          int M, int N, int K, pmhip_stream stream) {
    return gemm_impl(dtype, A, lda, W, ldw, bias, residual, ldr, res_rows, out, ldo, out_dtype, M, N, K, nullptr, stream);
}

// bf16 hi/lo residual stream: (hi, lo) <- split(A . W^T + bias + (res_hi + res_lo)); in place when the planes coincide.
// row_stats (optional): per-row partial statistics of the new hi plane, [M][N/64][2] (gemm_common.h)
static int gemm_hilo_impl(const void* A, int lda, const void* W, int ldw, const float* bias, const void* res_hi,
                          const void* res_lo, int ldr, int res_rows, void* out_hi, void* out_lo, int ldo, int M, int N, int K,
                          float* row_stats, pmhip_stream stream, const float* center_coef = nullptr, float center_extra = 0.f,
                          float* shift = nullptr, int shift_mode = 0) {
    GemmParams p{};
    p.center_coef = center_coef; p.center_extra = center_coef ? center_extra : 0.f; p.shift = shift; p.shift_mode = shift ? shift_mode : 0;
    PM_REQUIRE(shift_mode >= 0 && shift_mode <= 2, "gemm_hilo_center: shift_mode=%d", shift_mode);
    PM_REQUIRE(!(center_coef || shift_mode == 2) || res_rows <= 0 || res_rows >= M,
               "gemm_hilo_center: centring / shift accumulation needs the residual to be the stream itself (no row modulo)");
    p.A = A; p.W = W; p.bias = bias; p.residual = reinterpret_cast<const float*>(res_hi); p.out = out_hi;
    p.res_lo = reinterpret_cast<const bf16_t*>(res_lo); p.out_lo = reinterpret_cast<bf16_t*>(out_lo);
    p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.res_rows = res_rows > 0 ? res_rows : M; p.ldo = ldo;
    p.M = M; p.N = N; p.K = K;
    p.row_stats = row_stats;
    PM_TRY(check_common(p, PMHIP_BF16));
    PM_REQUIRE(res_hi && res_lo && out_hi && out_lo, "gemm_hilo: null plane");
    PM_REQUIRE(N % 8 == 0 && ldo % 8 == 0 && ldr % 8 == 0, "gemm_hilo: N=%d, ldo=%d, ldr=%d must be multiples of 8", N, ldo, ldr);
    PM_REQUIRE(!row_stats || N % 64 == 0, "gemm_hilo_stats: N=%d must be a multiple of 64", N);
    hipStream_t s = (hipStream_t)stream;
    if (use2b(p, PMHIP_BF16, EPI_STD, PMHIP_BF16)) return pm_gemm2b_launch(p, EPI_STD, PMHIP_BF16, s);
    if (use256(p, PMHIP_BF16, EPI_STD, PMHIP_BF16)) return pm_gemm256_launch(p, EPI_STD, PMHIP_BF16, s);
    return launch<bf16_t, EPI_STD, bf16_t>(p, s);
}

extern "C" int pmhip_gemm_hilo(const void* A, int lda, const void* W, int ldw, const float* bias, const void* res_hi,
                               const void* res_lo, int ldr, int res_rows, void* out_hi, void* out_lo, int ldo, int M, int N, int K,
                               pmhip_stream stream) {
    return gemm_hilo_impl(A, lda, W, ldw, bias, res_hi, res_lo, ldr, res_rows, out_hi, out_lo, ldo, M, N, K, nullptr, stream);
}

extern "C" int pmhip_gemm_hilo_stats(const void* A, int lda, const void* W, int ldw, const float* bias, const void* res_hi,
                                     const void* res_lo, int ldr, int res_rows, void* out_hi, void* out_lo, int ldo, int M, int N,
                                     int K, float* row_stats, pmhip_stream stream) {
    PM_REQUIRE(row_stats, "gemm_hilo_stats: null statistics buffer");
    return gemm_hilo_impl(A, lda, W, ldw, bias, res_hi, res_lo, ldr, res_rows, out_hi, out_lo, ldo, M, N, K, row_stats, stream);
}

// the same producer storing the pair of x - (row mean of the previous hi plane): gemm_common.h, GemmParams::center_coef
extern "C" int pmhip_gemm_hilo_center(const void* A, int lda, const void* W, int ldw, const float* bias, const void* res_hi,
                                      const void* res_lo, int ldr, int res_rows, void* out_hi, void* out_lo, int ldo, int M, int N,
                                      int K, float* row_stats, const float* center_coef, float center_extra, float* shift,
                                      int shift_mode, pmhip_stream stream) {
    return gemm_hilo_impl(A, lda, W, ldw, bias, res_hi, res_lo, ldr, res_rows, out_hi, out_lo, ldo, M, N, K, row_stats, stream, center_coef,
                          center_extra, shift, shift_mode);
}

extern "C" int pmhip_gemm_ln(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, void* out, int ldo,
                             int out_dtype, int M, int N, int K, const pmhip_lnfold* ln, pmhip_stream stream) {
    PM_REQUIRE(ln, "gemm_ln: null fold descriptor");
    return gemm_impl(dtype, A, lda, W, ldw, bias, nullptr, 0, 0, out, ldo, out_dtype, M, N, K, ln, stream);
}

extern "C" int pmhip_lnfold_supported(int dtype, int epi_kind, int M, int N, int K) {
    return dtype == PMHIP_BF16 && K % 128 == 0 && K >= 128 && epi_kind >= 0 && epi_kind <= 2 && M % 256 == 0 && N % 256 == 0 &&
           (unsigned long long)M * K * 2 < (1ull << 31) && (unsigned long long)N * K * 2 < (1ull << 31) ? 1 : 0;
}

static int gemm_swiglu_impl(int dtype, const void* A, int lda, const void* W12p, const float* b12p,
                                 void* out, int ldo, int M, int Hp, int K, const pmhip_lnfold* ln, pmhip_stream stream) {
    GemmParams p{};
    p.A = A; p.W = W12p; p.bias = b12p; p.out = out;
    p.lda = lda; p.ldw = K; p.ldo = ldo; p.M = M; p.N = 2 * Hp; p.K = K;
    p.fast_math = (dtype == PMHIP_BF16);
    PM_TRY(check_common(p, dtype));
    PM_REQUIRE(Hp % 64 == 0, "gemm_swiglu: padded hidden width %d must be a multiple of 64", Hp);
    PM_REQUIRE(b12p && out && ldo % 8 == 0, "gemm_swiglu: bias/out required, ldo multiple of 8");
    PM_TRY(set_lnfold(p, ln, dtype, EPI_SWIGLU, dtype));
    hipStream_t s = (hipStream_t)stream;
    if (ln) return pm_gemm256_launch(p, EPI_SWIGLU, dtype, s);
    if (use2b(p, dtype, EPI_SWIGLU, dtype)) return pm_gemm2b_launch(p, EPI_SWIGLU, dtype, s);
    if (use256(p, dtype, EPI_SWIGLU, dtype)) return pm_gemm256_launch(p, EPI_SWIGLU, dtype, s);
    if (dtype == PMHIP_F32) return launch<float, EPI_SWIGLU, float>(p, s);
    return launch<bf16_t, EPI_SWIGLU, bf16_t>(p, s);
}

extern "C" int pmhip_gemm_swiglu(int dtype, const void* A, int lda, const void* W12p, const float* b12p,
                                 void* out, int ldo, int M, int Hp, int K, pmhip_stream stream) {
    return gemm_swiglu_impl(dtype, A, lda, W12p, b12p, out, ldo, M, Hp, K, nullptr, stream);
}

extern "C" int pmhip_gemm_swiglu_ln(int dtype, const void* A, int lda, const void* W12p, const float* b12p, void* out, int ldo,
                                    int M, int Hp, int K, const pmhip_lnfold* ln, pmhip_stream stream) {
    PM_REQUIRE(ln, "gemm_swiglu_ln: null fold descriptor");
    return gemm_swiglu_impl(dtype, A, lda, W12p, b12p, out, ldo, M, Hp, K, ln, stream);
}

static int gemm_heads_impl(int dtype, const void* A, int lda, const void* W, int ldw, int M, int K,
                                int heads, int tokens, int tokens_pad, int nparts,
                                const int* part_kinds_host, void* const* part_outs_host, float q_scale, const pmhip_lnfold* ln,
                                pmhip_stream stream) {
    GemmParams p{};
    p.A = A; p.W = W; p.lda = lda; p.ldw = ldw; p.M = M; p.K = K;
    p.heads = heads; p.tokens = tokens; p.tokens_pad = tokens_pad; p.inner = heads * 64;
    p.N = nparts * p.inner; p.q_scale = q_scale;
    PM_REQUIRE(nparts >= 1 && nparts <= 3, "gemm_heads: nparts=%d", nparts);
    PM_REQUIRE(heads > 0 && tokens > 0 && tokens_pad >= tokens, "gemm_heads: bad head/token geometry");
    PM_REQUIRE(M % tokens == 0, "gemm_heads: M=%d is not a multiple of tokens=%d", M, tokens);
    for (int i = 0; i < nparts; ++i) {
        p.kinds[i] = part_kinds_host[i];
        p.outs[i] = part_outs_host[i];
        PM_REQUIRE(p.outs[i], "gemm_heads: null output %d", i);
        PM_REQUIRE(p.kinds[i] >= 0 && p.kinds[i] <= 2, "gemm_heads: bad part kind");
    }
    PM_TRY(check_common(p, dtype));
    PM_TRY(set_lnfold(p, ln, dtype, EPI_HEADS, dtype));
    hipStream_t s = (hipStream_t)stream;
    if (ln) return pm_gemm256_launch(p, EPI_HEADS, dtype, s);
    if (use2b(p, dtype, EPI_HEADS, dtype)) return pm_gemm2b_launch(p, EPI_HEADS, dtype, s);
    if (use256(p, dtype, EPI_HEADS, dtype)) return pm_gemm256_launch(p, EPI_HEADS, dtype, s);
    if (dtype == PMHIP_F32) return launch<float, EPI_HEADS, float>(p, s);
    return launch<bf16_t, EPI_HEADS, bf16_t>(p, s);
}

extern "C" int pmhip_gemm_heads(int dtype, const void* A, int lda, const void* W, int ldw, int M, int K,
                                int heads, int tokens, int tokens_pad, int nparts,
                                const int* part_kinds_host, void* const* part_outs_host, float q_scale,
                                pmhip_stream stream) {
    return gemm_heads_impl(dtype, A, lda, W, ldw, M, K, heads, tokens, tokens_pad, nparts, part_kinds_host, part_outs_host, q_scale,
                           nullptr, stream);
}

extern "C" int pmhip_gemm_heads_ln(int dtype, const void* A, int lda, const void* W, int ldw, int M, int K, int heads,
                                   int tokens, int tokens_pad, int nparts, const int* part_kinds_host,
                                   void* const* part_outs_host, float q_scale, const pmhip_lnfold* ln, pmhip_stream stream) {
    PM_REQUIRE(ln, "gemm_heads_ln: null fold descriptor");
    return gemm_heads_impl(dtype, A, lda, W, ldw, M, K, heads, tokens, tokens_pad, nparts, part_kinds_host, part_outs_host, q_scale, ln,
                           stream);
}
