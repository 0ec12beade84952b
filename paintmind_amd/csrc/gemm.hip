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
// with 4 consecutive n for one m; the epilogue transposes 16 rows at a time through the idle LDS stage
// so that bias / residual loads and all stores are 128-256 contiguous bytes per row.
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

constexpr int ESTRIDE = 68;                       // floats per row of the per-wave epilogue buffer (64 + pad)

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

    // Tile walk: n-tiles are visited in chunks of <= 8 (<= 1 MiB of W in bf16) with m-tiles varying inside a
    // chunk, so that an XCD's resident blocks share one W chunk + a few A row panels inside its 4 MiB L2.
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nblocks = gridDim.x;
    const int vb = xcd_remap(blockIdx.x, nblocks);
    const int nchunks = (tiles_n + 7) / 8;
    const int cw = (tiles_n + nchunks - 1) / nchunks;              // n-tiles per chunk (last chunk may be narrower)
    const int chunk = vb / (tiles_m * cw);
    const int cw_here = min(cw, tiles_n - chunk * cw);
    const int rem = vb - chunk * tiles_m * cw;
    const int m0 = (rem / cw_here) * BM;
    const int n0 = (chunk * cw + rem % cw_here) * BN;

    const T* A = reinterpret_cast<const T*>(p.A);
    const T* W = reinterpret_cast<const T*>(p.W);
    constexpr int KSTEP = ROWB / (int)sizeof(T);
    const int nk = p.K / KSTEP;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // epilogue geometry (see below): one store instruction = RPI rows x 64 columns of this wave's tile
    constexpr int CPL = 16 / (int)sizeof(OutT);                    // columns per lane per store (16 B)
    constexpr int LPR = 64 / CPL, RPI = 64 / LPR, ITERS = 16 / RPI;
    const int nw = n0 + wn * 64;                                   // first column of this wave
    const int ccol = (lane % LPR) * CPL, ncol = nw + ccol;

    // The residual tile is fetched BEFORE the K loop into registers (its HBM latency would otherwise sit,
    // exposed, between the last MFMA and the stores).  Only the f32-output epilogue carries a residual.
    float4 rpre[EPI == EPI_STD && sizeof(OutT) == 4 ? 16 : 1];
    if constexpr (EPI == EPI_STD && sizeof(OutT) == 4) {
        if (p.residual) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    int mm = m0 + wm * 64 + mi * 16 + it * RPI + lane / LPR;
                    mm = mm < p.M ? mm : p.M - 1;
                    const int nc = ncol < p.N ? ncol : 0;
                    rpre[mi * ITERS + it] = *reinterpret_cast<const float4*>(p.residual + (size_t)(mm % p.res_rows) * p.ldr + nc);
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

    // ------------------------------------------------------------------ epilogue
    // The MFMA result layout gives a lane 4 consecutive n of ONE row, i.e. 16 rows x 64 B per store
    // instruction.  Each wave therefore transposes its tile, 16 rows at a time, through the LDS stage that
    // is idle after the K loop (stage nk&1: its last reads finished before the final barrier), so that
    // 4 lanes cover one row's 64 columns: bias / residual loads and the stores are 256 B (f32) or 128 B
    // (bf16) contiguous per row.
    float* ebuf = reinterpret_cast<float*>(lds + (nk & 1) * STAGE_BYTES) + wave * (16 * ESTRIDE);
    const int erow = lane >> 2;
    float bias_v[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) bias_v[j] = 0.f;
    if constexpr (EPI == EPI_STD) {
        if (p.bias && ncol < p.N) {
#pragma unroll
            for (int j = 0; j < CPL; j += 4) {
                const float4 bb = *reinterpret_cast<const float4*>(p.bias + ncol + j);
                bias_v[j] = bb.x; bias_v[j + 1] = bb.y; bias_v[j + 2] = bb.z; bias_v[j + 3] = bb.w;
            }
        }
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
            *reinterpret_cast<f32x4_t*>(ebuf + l15 * ESTRIDE + ni * 16 + g * 4) = acc[mi][ni];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int mbase = m0 + wm * 64 + mi * 16;
        const int m = mbase + erow;
        if constexpr (EPI == EPI_STD) {
            // one store instruction = RPI rows x 64 columns, LPR adjacent lanes per row (full 128-B lines)
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int r = it * RPI + lane / LPR;
                const int mm = mbase + r;
                if (mm < p.M && ncol < p.N) {
                    float v[CPL];
#pragma unroll
                    for (int j = 0; j < CPL; j += 4) {
                        const float4 t = *reinterpret_cast<const float4*>(ebuf + r * ESTRIDE + ccol + j);
                        v[j] = t.x + bias_v[j]; v[j + 1] = t.y + bias_v[j + 1]; v[j + 2] = t.z + bias_v[j + 2]; v[j + 3] = t.w + bias_v[j + 3];
                    }
                    if constexpr (sizeof(OutT) == 4) {
                        if (p.residual) {
                            const float4 rr = rpre[mi * ITERS + it];
                            v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                        }
                    }
                    store_row(reinterpret_cast<OutT*>(p.out) + (size_t)mm * p.ldo + ncol, v);
                }
            }
        } else if constexpr (EPI == EPI_SWIGLU) {
            // wave columns: [x1 0-15 | x2 0-15 | x1 16-31 | x2 16-31] of 32 hidden columns; lane (erow, q) gates
            // hidden columns q*8 .. q*8+7
            if (m < p.M && nw < p.N) {
                const int q = lane & 3;
                const int c1 = (q >> 1) * 32 + (q & 1) * 8;          // x1 column inside the wave tile
                const float* e1 = ebuf + erow * ESTRIDE + c1;
                const float4 a0 = *reinterpret_cast<const float4*>(e1), a1 = *reinterpret_cast<const float4*>(e1 + 4);
                const float4 g0 = *reinterpret_cast<const float4*>(e1 + 16), g1 = *reinterpret_cast<const float4*>(e1 + 20);
                const float* b1 = p.bias + nw + c1;
                const float4 ba0 = *reinterpret_cast<const float4*>(b1), ba1 = *reinterpret_cast<const float4*>(b1 + 4);
                const float4 bg0 = *reinterpret_cast<const float4*>(b1 + 16), bg1 = *reinterpret_cast<const float4*>(b1 + 20);
                float h[8];
                h[0] = silu_mul(a0.x + ba0.x, g0.x + bg0.x, p.fast_math); h[1] = silu_mul(a0.y + ba0.y, g0.y + bg0.y, p.fast_math);
                h[2] = silu_mul(a0.z + ba0.z, g0.z + bg0.z, p.fast_math); h[3] = silu_mul(a0.w + ba0.w, g0.w + bg0.w, p.fast_math);
                h[4] = silu_mul(a1.x + ba1.x, g1.x + bg1.x, p.fast_math); h[5] = silu_mul(a1.y + ba1.y, g1.y + bg1.y, p.fast_math);
                h[6] = silu_mul(a1.z + ba1.z, g1.z + bg1.z, p.fast_math); h[7] = silu_mul(a1.w + ba1.w, g1.w + bg1.w, p.fast_math);
                OutT* out = reinterpret_cast<OutT*>(p.out) + (size_t)m * p.ldo + (nw >> 1) + q * 8;
                store4(out, h[0], h[1], h[2], h[3]);
                store4(out + 4, h[4], h[5], h[6], h[7]);
            }
        } else {  // EPI_HEADS: the wave's 64 columns are exactly one head of one part
            if (nw < p.N) {
                const int part = nw / p.inner;
                const int h = (nw % p.inner) >> 6;
                const int kind = p.kinds[part];
                OutT* dst = reinterpret_cast<OutT*>(p.outs[part]);
                if (kind != PMHIP_PART_V) {
                    const int tstride = kind == PMHIP_PART_Q ? p.tokens : p.tokens_pad;
                    const float sc = kind == PMHIP_PART_Q ? p.q_scale : 1.0f;
#pragma unroll
                    for (int it = 0; it < ITERS; ++it) {
                        const int r = it * RPI + lane / LPR;
                        const int mm = mbase + r;
                        if (mm < p.M) {
                            const int b = mm / p.tokens, t = mm % p.tokens;
                            float v[CPL];
#pragma unroll
                            for (int j = 0; j < CPL; j += 4) {
                                const float4 t4 = *reinterpret_cast<const float4*>(ebuf + r * ESTRIDE + ccol + j);
                                v[j] = t4.x * sc; v[j + 1] = t4.y * sc; v[j + 2] = t4.z * sc; v[j + 3] = t4.w * sc;
                            }
                            store_row(dst + (((size_t)b * p.heads + h) * tstride + t) * 64 + ccol, v);
                        }
                    }
                } else {
                    // V^T[b,h,d,t]: lane = d, its 16 values are 16 consecutive tokens
                    const int d = lane;
                    const int b0 = mbase / p.tokens, t0 = mbase % p.tokens;
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = ebuf[r * ESTRIDE + d];
                    if (mbase + 15 < p.M && t0 + 15 < p.tokens && (t0 & 7) == 0) {
                        store16(dst + (((size_t)b0 * p.heads + h) * 64 + d) * p.tokens_pad + t0, v);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int mm = mbase + r;
                            if (mm < p.M) {
                                const int b = mm / p.tokens, t = mm % p.tokens;
                                dst[(((size_t)b * p.heads + h) * 64 + d) * p.tokens_pad + t] = from_f32<OutT>(v[r]);
                            }
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next 16 rows overwrite ebuf
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
    PM_REQUIRE(!residual || out_dtype == PMHIP_F32, "gemm: a residual needs an f32 output (the residual stream is f32)");
    PM_REQUIRE(out_dtype == PMHIP_F32 || out_dtype == dtype, "gemm: out dtype must be f32 or the compute dtype");
    PM_REQUIRE(out_dtype == PMHIP_F32 || ldo % 8 == 0, "gemm: bf16 output needs ldo to be a multiple of 8");
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
    PM_REQUIRE(b12p && out && ldo % 8 == 0, "gemm_swiglu: bias/out required, ldo multiple of 8");
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
