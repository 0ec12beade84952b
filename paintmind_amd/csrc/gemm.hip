// NT GEMM family for gfx950: out = A[M,K] . W[N,K]^T with fused epilogues.
//
// Both operands are K-contiguous, which is exactly the MFMA fragment shape (16 B of k per lane),
// so neither is transposed anywhere.  Tile 128(m) x 128(n) x 128 bytes of k per step
// (64 bf16 / 32 f32), 256 threads = 4 waves as 2(m) x 2(n), each wave 64x64 = 4x4 MFMA tiles.
// Staging is direct-to-LDS (global_load_lds_dwordx4): the LDS image is lane-linear, so the
// bank-conflict XOR swizzle is applied to the per-lane SOURCE address and again on the ds_read
// (cdna_hip_programming.md 5.4 rule 21).  Two LDS stages; the loads of step t+1 are in flight while
// step t is multiplied.
//
// The MFMA is issued with W as the row operand and A as the column operand, so each lane ends up
// with 4 CONSECUTIVE n for one m: bias/residual are float4 loads and stores are 8/16 B wide.
//
// Replaces torch.nn.Linear / aten::addmm at: modules/attention.py:46-49,59; modules/mlp.py:27-31;
// stage1/vqmodel.py:23,28; stage1/layers.py:107 (patch-embed conv as GEMM),149;
// stage2/transformer.py:81,85,91 (reference paths).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 128;                          // bytes of k per tile row per step
constexpr int TILE_BYTES = 128 * ROWB;             // one operand tile, 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;        // A tile + W tile
constexpr int THREADS = 256;

enum { EPI_STD = 0, EPI_SWIGLU = 1, EPI_HEADS = 2 };

struct GemmParams {
    const void* A; const void* W;
    const float* bias; const float* residual;
    void* out;
    int lda, ldw, ldr, res_rows, ldo;
    int M, N, K;
    // EPI_HEADS
    int heads, tokens, tokens_pad, inner;
    int kinds[3];
    void* outs[3];
    float q_scale;
    int fast_math;                                 // SwiGLU: 1 = fast exp (bf16 mode)
};

// XCD-aware, bijective block remap: consecutive virtual ids (which share an A row panel) stay on
// one XCD's L2 (block b is dispatched to XCD b % 8).
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int xcd = bid & 7, local = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

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

__device__ __forceinline__ uint4 read_frag(const unsigned char* lds_tile, int row, int slot) {
    return *reinterpret_cast<const uint4*>(lds_tile + row * ROWB + ((slot ^ (row & 7)) << 4));
}

__device__ __forceinline__ float silu_mul(float x1, float x2, int fast) {
    const float e = fast ? __expf(-x1) : expf(-x1);
    return (x1 / (1.0f + e)) * x2;
}

template <typename T, int EPI, typename OutT>
__global__ __launch_bounds__(THREADS) void gemm_nt_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int nblocks = gridDim.x;
    const int vb = xcd_remap(blockIdx.x, nblocks);
    const int m0 = (vb / tiles_n) * BM;
    const int n0 = (vb % tiles_n) * BN;

    const T* A = reinterpret_cast<const T*>(p.A);
    const T* W = reinterpret_cast<const T*>(p.W);
    constexpr int KSTEP = ROWB / (int)sizeof(T);
    const int nk = p.K / KSTEP;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

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

    // ------------------------------------------------------------------ epilogue
    // lane holds, for tile (mi, ni): m = .. + l15 ; n = .. + 4*g + r (r = 0..3)
    if constexpr (EPI == EPI_STD) {
        OutT* out = reinterpret_cast<OutT*>(p.out);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wm * 64 + mi * 16 + l15;
            if (m >= p.M) continue;
            const float* rrow = p.residual ? p.residual + (size_t)(m % p.res_rows) * p.ldr : nullptr;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int nb = n0 + wn * 64 + ni * 16 + g * 4;
                if (nb >= p.N) continue;
                float v0 = acc[mi][ni][0], v1 = acc[mi][ni][1], v2 = acc[mi][ni][2], v3 = acc[mi][ni][3];
                if (p.bias) {
                    const float4 b = *reinterpret_cast<const float4*>(p.bias + nb);
                    v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;
                }
                if (rrow) {
                    const float4 rr = *reinterpret_cast<const float4*>(rrow + nb);
                    v0 += rr.x; v1 += rr.y; v2 += rr.z; v3 += rr.w;
                }
                store4(out + (size_t)m * p.ldo + nb, v0, v1, v2, v3);
            }
        }
    } else if constexpr (EPI == EPI_SWIGLU) {
        // packed rows: [16 rows of x1 | the same 16 rows of x2] repeated, so tiles (2p, 2p+1) of a
        // wave hold x1 and x2 of the SAME hidden columns in the SAME lanes
        OutT* out = reinterpret_cast<OutT*>(p.out);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wm * 64 + mi * 16 + l15;
            if (m >= p.M) continue;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int n1 = n0 + wn * 64 + (2 * pr) * 16 + g * 4;     // packed row of x1
                const int n2 = n1 + 16;                                   // packed row of x2
                if (n1 >= p.N) continue;
                const float4 b1 = *reinterpret_cast<const float4*>(p.bias + n1);
                const float4 b2 = *reinterpret_cast<const float4*>(p.bias + n2);
                const f32x4_t a1 = acc[mi][2 * pr], a2 = acc[mi][2 * pr + 1];
                const int j = ((n0 + wn * 64) >> 1) + pr * 16 + g * 4;    // hidden column
                store4(out + (size_t)m * p.ldo + j,
                       silu_mul(a1[0] + b1.x, a2[0] + b2.x, p.fast_math),
                       silu_mul(a1[1] + b1.y, a2[1] + b2.y, p.fast_math),
                       silu_mul(a1[2] + b1.z, a2[2] + b2.z, p.fast_math),
                       silu_mul(a1[3] + b1.w, a2[3] + b2.w, p.fast_math));
            }
        }
    } else {  // EPI_HEADS: a wave's 64 n-columns are exactly one head of one part
        const int nw = n0 + wn * 64;
        if (nw < p.N) {
            const int part = nw / p.inner;
            const int h = (nw % p.inner) >> 6;
            const int kind = p.kinds[part];
            OutT* dst = reinterpret_cast<OutT*>(p.outs[part]);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int m = m0 + wm * 64 + mi * 16 + l15;
                if (m >= p.M) continue;
                const int b = m / p.tokens, t = m % p.tokens;
                const size_t bh = (size_t)b * p.heads + h;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int d = ni * 16 + g * 4;
                    const f32x4_t a = acc[mi][ni];
                    if (kind == PMHIP_PART_Q) {
                        store4(dst + (bh * p.tokens + t) * 64 + d, a[0] * p.q_scale, a[1] * p.q_scale,
                               a[2] * p.q_scale, a[3] * p.q_scale);
                    } else if (kind == PMHIP_PART_K) {
                        store4(dst + (bh * p.tokens_pad + t) * 64 + d, a[0], a[1], a[2], a[3]);
                    } else {
                        OutT* vt = dst + (bh * 64 + d) * p.tokens_pad + t;
                        vt[0] = from_f32<OutT>(a[0]);
                        vt[(size_t)p.tokens_pad] = from_f32<OutT>(a[1]);
                        vt[(size_t)2 * p.tokens_pad] = from_f32<OutT>(a[2]);
                        vt[(size_t)3 * p.tokens_pad] = from_f32<OutT>(a[3]);
                    }
                }
            }
        }
    }
}

template <typename T, int EPI, typename OutT>
int launch(const GemmParams& p, hipStream_t s) {
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    PmTimer tm(FAM_GEMM, s);
    hipLaunchKernelGGL((gemm_nt_kernel<T, EPI, OutT>), dim3(tiles), dim3(THREADS), 0, s, p);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
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

extern "C" int pmhip_gemm(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias,
                          const float* residual, int ldr, int res_rows, void* out, int ldo, int out_dtype,
                          int M, int N, int K, pmhip_stream stream) {
    GemmParams p{};
    p.A = A; p.W = W; p.bias = bias; p.residual = residual; p.out = out;
    p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.res_rows = res_rows > 0 ? res_rows : M; p.ldo = ldo;
    p.M = M; p.N = N; p.K = K;
    PM_TRY(check_common(p, dtype));
    PM_REQUIRE(out, "gemm: null out");
    PM_REQUIRE(N % 4 == 0 && ldo % 4 == 0, "gemm: N=%d and ldo=%d must be multiples of 4", N, ldo);
    PM_REQUIRE(!residual || ldr % 4 == 0, "gemm: ldr must be a multiple of 4");
    PM_REQUIRE(out_dtype == PMHIP_F32 || out_dtype == dtype, "gemm: out dtype must be f32 or the compute dtype");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PMHIP_F32) return launch<float, EPI_STD, float>(p, s);
    if (out_dtype == PMHIP_F32) return launch<bf16_t, EPI_STD, float>(p, s);
    return launch<bf16_t, EPI_STD, bf16_t>(p, s);
}

extern "C" int pmhip_gemm_swiglu(int dtype, const void* A, int lda, const void* W12p, const float* b12p,
                                 void* out, int ldo, int M, int Hp, int K, pmhip_stream stream) {
    GemmParams p{};
    p.A = A; p.W = W12p; p.bias = b12p; p.out = out;
    p.lda = lda; p.ldw = K; p.ldo = ldo; p.M = M; p.N = 2 * Hp; p.K = K;
    p.fast_math = (dtype == PMHIP_BF16);
    PM_TRY(check_common(p, dtype));
    PM_REQUIRE(Hp % 64 == 0, "gemm_swiglu: padded hidden width %d must be a multiple of 64", Hp);
    PM_REQUIRE(b12p && out && ldo % 4 == 0, "gemm_swiglu: bias/out required, ldo multiple of 4");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PMHIP_F32) return launch<float, EPI_SWIGLU, float>(p, s);
    return launch<bf16_t, EPI_SWIGLU, bf16_t>(p, s);
}

extern "C" int pmhip_gemm_heads(int dtype, const void* A, int lda, const void* W, int ldw, int M, int K,
                                int heads, int tokens, int tokens_pad, int nparts,
                                const int* part_kinds_host, void* const* part_outs_host, float q_scale,
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
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PMHIP_F32) return launch<float, EPI_HEADS, float>(p, s);
    return launch<bf16_t, EPI_HEADS, bf16_t>(p, s);
}
