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

template <typename T, int CH = 4>                          // CH = 16 / waves: 8-row chunks of the 128-row tile per wave
__device__ __forceinline__ void stage_tile(const T* __restrict__ base, int ld, int row0, int rows_total,
                                           int k0, unsigned char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int chunk = wave * CH + i;                   // 1 KiB = 8 rows x 128 B
        const int r = chunk * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ (r & 7);             // inverse swizzle on the source
        int gr = row0 + r;
        gr = gr < rows_total ? gr : rows_total - 1;        // clamp: guarded rows are never stored
        const unsigned char* src =
            reinterpret_cast<const unsigned char*>(base + (size_t)gr * ld + k0) + slot * 16;
        glds16(src, lds_tile + chunk * 1024);
    }
}

// FOLD (bf16 only): the LayerNorm folded into this GEMM, as in the 256x256 kernel (gemm_common.h: A is the raw hi plane, W carries
// gamma, the epilogue applies rstd * acc - rstd * mean * c[n] + d[n]).  Same MFMA chain per output element (K-tiles in order, the
// two k-halves of a K-tile in order), same ln_apply4 arithmetic: the result is BIT-IDENTICAL to the 256x256 kernel's
// (tests/test_gpu_ops.py), so which of the two serves a shape may depend on the batch size although an image's result may not.
// It takes the folded GEMMs of small batches, where the 256x256 tiling is a handful of workgroups on 256 CUs.
// STAGES = 4 (bf16, launches of at most one workgroup per CU): the LATENCY form of the K loop.  With two stages every K-tile
// waits for the DMA issued one K-tile earlier -- about 1 us per K-tile when nothing else runs on the CU, 22 K-tiles for the
// FFN w3 producer of ONE image.  Four stages keep three K-tiles in flight behind counted vmcnt waits; fragment reads are
// inline-asm ds_read_b128 (hipcc drains vmcnt to 0 in front of an ordinary LDS read while a DMA is in flight).  The MFMA
// order per output element is unchanged (K-tiles in order, k-halves in order): same bits as STAGES = 2 and as gemm256.hip.
// WM = waves along m (x 2 along n).  WM = 4 (the latency form only): EIGHT waves, two per SIMD, each 32 x 64 of the tile: a wave
// issues half of the LDS-DMA instructions per K-tile (with four waves a K-tile cost 0.57 us for 0.23 us of MFMA) and its SIMD
// partner computes meanwhile (the head-split epilogue then transposes V^T in 32-token blocks, gemm_common.h).
template <typename T, int EPI, typename OutT, bool FOLD = false, int STAGES = 2, int WM = 2>
__global__ __launch_bounds__(WM * 128) void gemm_nt_kernel(const GemmParams p) {
    static_assert(STAGES == 2 || (STAGES == 4 && sizeof(T) == 2), "the four-stage K loop is the bf16 small-batch form");
    static_assert(WM == 2 || (WM == 4 && STAGES == 4), "eight waves only in the latency form");
    constexpr int MI = 8 / WM;                          // 16-row accumulator tiles per wave (wave tile = MI*16 x 64)
    constexpr int CH = 8 / WM;                          // 8-row DMA chunks per wave and operand
    __shared__ __attribute__((aligned(1024))) unsigned char lds[STAGES * STAGE_BYTES];

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

    f32x4_t acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // The residual tile is fetched BEFORE the K loop into registers, in the epilogue's store layout (its HBM
    // latency would otherwise sit, exposed, between the last MFMA and the stores).
    constexpr bool RPRE = (EPI == EPI_STD && sizeof(OutT) == 4);
    float4 rpre[RPRE ? MI * 4 : 1];
    if constexpr (RPRE) {
        if (p.residual) {
            const int ncol = n0 + wn * 64 + (lane & 15) * 4;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    int mm = m0 + wm * (MI * 16) + mi * 16 + it * 4 + (lane >> 4);
                    mm = mm < p.M ? mm : p.M - 1;
                    const int nc = ncol < p.N ? ncol : 0;
                    rpre[mi * 4 + it] = *reinterpret_cast<const float4*>(p.residual + (size_t)(mm % p.res_rows) * p.ldr + nc);
                }
        }
    }

    stage_tile<T, CH>(A, p.lda, m0, p.M, 0, lds, wave, lane);
    stage_tile<T, CH>(W, p.ldw, n0, p.N, 0, lds + TILE_BYTES, wave, lane);

    // fold coefficients of this lane's rows / columns, requested before the K loop (a small-batch launch has nothing else to
    // hide their latency behind)
    [[maybe_unused]] float4 fcc[FOLD ? 4 : 1], fdd[FOLD ? 4 : 1];
    [[maybe_unused]] float2 fab[FOLD ? MI : 1];
    [[maybe_unused]] float2 fpa[FOLD && STAGES == 4 ? MI : 1][8], fpb[FOLD && STAGES == 4 ? MI : 1][8];   // raw partial statistics (in-kernel coefficients)
    if constexpr (FOLD) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            fcc[ni] = *reinterpret_cast<const float4*>(p.ln_c + n0 + wn * 64 + ni * 16 + g * 4);
            fdd[ni] = *reinterpret_cast<const float4*>(p.ln_d + n0 + wn * 64 + ni * 16 + g * 4);
        }
        if (STAGES == 4 && p.ln_parts) {                     // launch-uniform: coefficients from the partial statistics, combined after the K loop
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                ln_coef_row_load(reinterpret_cast<const float2*>(p.ln_parts) + (size_t)(m0 + wm * (MI * 16) + mi * 16 + l15) * p.ln_nparts, p.ln_nparts,
                                 fpa[mi], fpb[mi]);
        } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fab[mi] = *reinterpret_cast<const float2*>(p.ln_coef + (size_t)(m0 + wm * (MI * 16) + mi * 16 + l15) * 2);
        }
    }

    if constexpr (STAGES == 4) {
        // K-tiles 1 and 2 follow tile 0 at once; tile kt + 3 is requested when tile kt is entered (its stage held tile kt - 1,
        // which every wave has finished reading once it has passed this K-tile's barrier)
        if (nk > 1) { stage_tile<T, CH>(A, p.lda, m0, p.M, KSTEP, lds + STAGE_BYTES, wave, lane); stage_tile<T, CH>(W, p.ldw, n0, p.N, KSTEP, lds + STAGE_BYTES + TILE_BYTES, wave, lane); }
        if (nk > 2) { stage_tile<T, CH>(A, p.lda, m0, p.M, 2 * KSTEP, lds + 2 * STAGE_BYTES, wave, lane); stage_tile<T, CH>(W, p.ldw, n0, p.N, 2 * KSTEP, lds + 2 * STAGE_BYTES + TILE_BYTES, wave, lane); }
        const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
        // per-lane fragment addresses: row (w * rows + f * 16 + l15), slot (kk * 4 + g) ^ (row & 7); f steps by an immediate
        const unsigned fa0 = lds_base + (unsigned)(wm * (MI * 16) + l15) * ROWB + (unsigned)(((0 + g) ^ (l15 & 7)) << 4);
        const unsigned fa1 = lds_base + (unsigned)(wm * (MI * 16) + l15) * ROWB + (unsigned)(((4 + g) ^ (l15 & 7)) << 4);
        const unsigned fw0 = lds_base + TILE_BYTES + (unsigned)(wn * 64 + l15) * ROWB + (unsigned)(((0 + g) ^ (l15 & 7)) << 4);
        const unsigned fw1 = lds_base + TILE_BYTES + (unsigned)(wn * 64 + l15) * ROWB + (unsigned)(((4 + g) ^ (l15 & 7)) << 4);
#define PM_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
        for (int kt = 0; kt < nk; ++kt) {
            // this wave's 2 * CH DMA instructions of tile kt have landed: at most two younger tiles stay in flight
            const int younger = min(nk - 1 - kt, 2);
            if constexpr (CH == 4) {
                if (younger == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (kt + 3 < nk) {
                unsigned char* nxt = lds + ((kt + 3) & 3) * STAGE_BYTES;
                stage_tile<T, CH>(A, p.lda, m0, p.M, (kt + 3) * KSTEP, nxt, wave, lane);
                stage_tile<T, CH>(W, p.ldw, n0, p.N, (kt + 3) * KSTEP, nxt + TILE_BYTES, wave, lane);
            }
            const unsigned so = (unsigned)(kt & 3) * STAGE_BYTES;
            typedef unsigned frag_t __attribute__((ext_vector_type(4)));   // (a struct type cannot be a tied asm operand)
            frag_t af[2][MI], wf[2][4];
            PM_DSR(af[0][0], fa0 + so, 0 * 2048); PM_DSR(af[0][1], fa0 + so, 1 * 2048);
            if constexpr (MI == 4) { PM_DSR(af[0][2], fa0 + so, 2 * 2048); PM_DSR(af[0][3], fa0 + so, 3 * 2048); }
            PM_DSR(wf[0][0], fw0 + so, 0 * 2048); PM_DSR(wf[0][1], fw0 + so, 1 * 2048); PM_DSR(wf[0][2], fw0 + so, 2 * 2048); PM_DSR(wf[0][3], fw0 + so, 3 * 2048);
            PM_DSR(af[1][0], fa1 + so, 0 * 2048); PM_DSR(af[1][1], fa1 + so, 1 * 2048);
            if constexpr (MI == 4) { PM_DSR(af[1][2], fa1 + so, 2 * 2048); PM_DSR(af[1][3], fa1 + so, 3 * 2048); }
            PM_DSR(wf[1][0], fw1 + so, 0 * 2048); PM_DSR(wf[1][1], fw1 + so, 1 * 2048); PM_DSR(wf[1][2], fw1 + so, 2 * 2048); PM_DSR(wf[1][3], fw1 + so, 3 * 2048);
            if constexpr (MI == 4)
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(wf[0][0]), "+v"(wf[0][1]),
                             "+v"(wf[0][2]), "+v"(wf[0][3]));
            else
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[0][2]), "+v"(wf[0][3]));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) Mma<T>::run(acc[mi][ni], __builtin_bit_cast(uint4, wf[0][ni]), __builtin_bit_cast(uint4, af[0][mi]));
            if constexpr (MI == 4)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]), "+v"(wf[1][0]), "+v"(wf[1][1]),
                             "+v"(wf[1][2]), "+v"(wf[1][3]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[1][0]), "+v"(af[1][1]), "+v"(wf[1][0]), "+v"(wf[1][1]), "+v"(wf[1][2]), "+v"(wf[1][3]));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) Mma<T>::run(acc[mi][ni], __builtin_bit_cast(uint4, wf[1][ni]), __builtin_bit_cast(uint4, af[1][mi]));
        }
#undef PM_DSR
        __builtin_amdgcn_s_barrier();                      // every wave is done with the last K-tile: any stage may serve the epilogue
    } else
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
    unsigned char* eraw = lds + (STAGES == 4 ? 0 : (nk & 1) * STAGE_BYTES) + wave * EPI_WAVE_BYTES;
    if constexpr (FOLD && STAGES == 4) {
        if (p.ln_parts) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                fab[mi] = ln_coef_row(fpa[mi], fpb[mi], p.ln_nparts, p.ln_eps);
                if (tn == 0 && wn == 0 && g == 0) reinterpret_cast<float2*>(p.ln_coef_out)[(size_t)(m0 + wm * (MI * 16) + mi * 16 + l15)] = fab[mi];
            }
        }
    }
    if constexpr (FOLD) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const pk2_t xx = pk_splat(fab[mi].x), yy = pk_splat(fab[mi].y);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) ln_apply4(acc[mi][ni], xx, yy, fcc[ni], fdd[ni]);
        }
    }
    if constexpr (EPI == EPI_STD && sizeof(OutT) == 2 && sizeof(T) == 2) {
        if (p.out_lo) { wave_epilogue<EPI, OutT, MI, 1, false, 2, NoHook, STAGES != 4>(p, acc, eraw, m0 + wm * (MI * 16), n0 + wn * 64, lane, rpre); return; }   // bf16 hi/lo residual stream
    }
    wave_epilogue<EPI, OutT, MI, RPRE ? MI * 4 : 1, false, -1, NoHook, STAGES != 4>(p, acc, eraw, m0 + wm * (MI * 16), n0 + wn * 64, lane, rpre);
}

int env_int(const char* name, int dflt);
// at most one workgroup per CU: nothing overlaps the K loop's DMA latency but the loop itself -> the four-stage form
// (128 KiB of LDS, one workgroup per CU).  PMHIP_GEMM128_DEEP_MAX_TILES (development): 0 = never.
bool deep128(int tiles, int K) {
    static const int deep_max = env_int("PMHIP_GEMM128_DEEP_MAX_TILES", 256);
    return tiles <= deep_max && K >= 4 * 64;
}

template <typename T, int EPI, typename OutT, bool FOLD = false>
int launch(const GemmParams& p, hipStream_t s) {
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    PmTimer tm(gemm_family(p, EPI), s);
    if constexpr (sizeof(T) == 2) {
        if (deep128(tiles, p.K)) {
            hipLaunchKernelGGL((gemm_nt_kernel<T, EPI, OutT, FOLD, 4, 4>), dim3(tiles), dim3(512), 0, s, p);
            PM_HIP(hipGetLastError());
            return PMHIP_OK;
        }
    }
    hipLaunchKernelGGL((gemm_nt_kernel<T, EPI, OutT, FOLD>), dim3(tiles), dim3(THREADS), 0, s, p);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

// process-wide development switches: read once, in the thread-safe initialiser of a function-local static (the first GEMMs of
// a process are launched from several lane threads at the same time)
int env_int(const char* name, int dflt) { return pm_dev_knob(name, dflt); }      // development builds only (common.h)

// A folded GEMM (set_lnfold: M, N multiples of 256) runs on the 128x128 kernel while the 256x256 tiling would leave at least half
// of the CUs without a workgroup: 4x the workgroups, a quarter of the serial K loop each, bit-identical results.
// PMHIP_FOLD128_MAX_TILES (development): largest 256x256 tile count that still takes the small kernel (0 = never).
bool fold_small(const GemmParams& p) {
    static const int max_tiles = env_int("PMHIP_FOLD128_MAX_TILES", 128);
    return (p.M / 256) * (p.N / 256) <= max_tiles;
}

bool use256(const GemmParams& p, int dtype, int epi, int out_dtype) {
    static const int g_use256 = env_int("PMHIP_GEMM256", 1);      // PMHIP_GEMM256=0 disables the 256x256 kernel
    return g_use256 && pm_gemm256_supported(p, dtype, epi, out_dtype);
}

// The two-workgroups-per-CU kernel (gemm2b.hip) takes the residual GEMMs with a short K loop (attention out-proj,
// K = inner): they are HBM-bound (fp32 residual in, fp32 out) and it streams them at ~4.5 TB/s where the 128x128
// kernel reaches 3.6.  Everything else measured equal or slower than gemm256.hip (tools/gemm_bench.py).
bool use2b(const GemmParams& p, int dtype, int epi, int out_dtype) {
    static const int g_use2b = env_int("PMHIP_GEMM2B", 1);   // 0 = never, 1 = where it wins (default), 2 = wherever it is supported (development)
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
int set_lnfold(GemmParams& p, const pmhip_lnfold* ln, int dtype, int epi, int out_dtype, pmhip_stream stream) {
    if (!ln) return PMHIP_OK;
    PM_REQUIRE(ln->coef && ln->c && ln->d, "gemm_ln: null fold pointer");
    PM_REQUIRE(dtype == PMHIP_BF16, "gemm_ln: the LayerNorm fold exists in bf16 mode only");
    PM_REQUIRE(p.K % 128 == 0, "gemm_ln: K=%d must be a multiple of 128", p.K);
    p.ln_c = ln->c; p.ln_d = ln->d; p.ln_coef = ln->coef;
    // any tile count: whether a LayerNorm is folded must not depend on the batch size (see pmhip_lnfold_supported)
    PM_REQUIRE(p.M % 256 == 0 && p.N % 256 == 0 && (unsigned long long)p.M * p.lda * 2 < (1ull << 31) &&
               (unsigned long long)p.N * p.ldw * 2 < (1ull << 31),
               "gemm_ln: shape M=%d N=%d K=%d is not served by the 256x256 kernel (M, N multiples of 256)", p.M, p.N, p.K);
    if (ln->parts) {
        // coef is an OUTPUT too: computed from the producer's partial statistics -- by the GEMM's own prologue where the small
        // kernel takes the launch (one launch less per LayerNorm: 424 launches of 5 us per 8-step generate of one image), by
        // pmhip_ln_coef_parts in front of it otherwise.  Same bits either way (common.h, lnp_*).
        PM_REQUIRE(ln->nparts > 0 && ln->nparts <= 16 && ln->nparts * 64 == p.K, "gemm_ln: nparts=%d does not describe K=%d columns", ln->nparts, p.K);
        if (fold_small(p) && deep128(ceil_div(p.M, BM) * ceil_div(p.N, BN), p.K)) {
            p.ln_parts = ln->parts; p.ln_nparts = ln->nparts; p.ln_eps = ln->eps; p.ln_coef_out = const_cast<float*>(ln->coef);
        } else {
            PM_TRY(pmhip_ln_coef_parts(ln->parts, ln->nparts, ln->eps, const_cast<float*>(ln->coef), p.M, stream));
        }
    }
    return PMHIP_OK;
}

int gemm_impl(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias,
                          const float* residual, int ldr, int res_rows, void* out, int ldo, int out_dtype,
                          int M, int N, int K, const pmhip_lnfold* ln, pmhip_stream stream, float* block_stats = nullptr) {
    GemmParams p{};
    p.A = A; p.W = W; p.bias = bias; p.residual = residual; p.out = out;
    p.block_stats = block_stats;
    PM_REQUIRE(!block_stats || (out_dtype == PMHIP_F32 && !residual && N % 64 == 0),
               "gemm_softmax_stats: block statistics need an f32 output without a residual and N=%d a multiple of 64", N);
    p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.res_rows = res_rows > 0 ? res_rows : M; p.ldo = ldo;
    p.M = M; p.N = N; p.K = K;
    PM_TRY(check_common(p, dtype));
    PM_REQUIRE(out, "gemm: null out");
    PM_REQUIRE(N % 4 == 0 && ldo % 4 == 0, "gemm: N=%d and ldo=%d must be multiples of 4", N, ldo);
    PM_REQUIRE(!residual || ldr % 4 == 0, "gemm: ldr must be a multiple of 4");
    PM_REQUIRE(!residual || out_dtype == PMHIP_F32, "gemm: a residual needs an f32 output (the residual stream is f32)");
    PM_REQUIRE(out_dtype == PMHIP_F32 || out_dtype == dtype, "gemm: out dtype must be f32 or the compute dtype");
    PM_REQUIRE(out_dtype == PMHIP_F32 || ldo % 8 == 0, "gemm: bf16 output needs ldo to be a multiple of 8");
    PM_TRY(set_lnfold(p, ln, dtype, EPI_STD, out_dtype, stream));
    hipStream_t s = (hipStream_t)stream;
    if (ln && fold_small(p)) return out_dtype == PMHIP_F32 ? launch<bf16_t, EPI_STD, float, true>(p, s) : launch<bf16_t, EPI_STD, bf16_t, true>(p, s);
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

// out (f32) = A . W^T + bias, LayerNorm optionally folded (ln may be NULL), plus the softmax statistics of every (row, 64-column
// block) of the result: block_stats [M][N/64][2] = (max, sum of exp), common.h softmax_block_stat.  The logits GEMM of the
// MaskGIT step (transformer.py:91): pmhip_sample_rows_stats then reads the statistics and the top-k blocks of a row only.
extern "C" int pmhip_gemm_softmax_stats(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, float* out, int ldo,
                                        int M, int N, int K, const pmhip_lnfold* ln, float* block_stats, pmhip_stream stream) {
    PM_REQUIRE(block_stats, "gemm_softmax_stats: null statistics buffer");
    return gemm_impl(dtype, A, lda, W, ldw, bias, nullptr, 0, 0, out, ldo, PMHIP_F32, M, N, K, ln, stream, block_stats);
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
    PM_TRY(set_lnfold(p, ln, dtype, EPI_SWIGLU, dtype, stream));
    hipStream_t s = (hipStream_t)stream;
    if (ln && fold_small(p)) return launch<bf16_t, EPI_SWIGLU, bf16_t, true>(p, s);
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
    PM_TRY(set_lnfold(p, ln, dtype, EPI_HEADS, dtype, stream));
    hipStream_t s = (hipStream_t)stream;
    if (ln && fold_small(p)) return launch<bf16_t, EPI_HEADS, bf16_t, true>(p, s);
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
