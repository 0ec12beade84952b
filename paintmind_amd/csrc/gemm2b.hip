// 256x128 bf16 NT GEMM for gfx950, TWO workgroups per CU (the short-K member of the family in gemm.hip).
//
// Why it exists.  With K = 512 a 256x256 tile of gemm256.hip spends ~25 % of its life in the epilogue (LDS
// transposition + 128 KiB of stores per tile) with the matrix pipes idle, and because one workgroup owns a CU and
// all CUs start together, every CU stores at the same moment (an L2 write burst) and then every CU computes.
// Here a CU hosts two independent 4-wave workgroups: each SIMD holds one wave of either, so while one workgroup is
// in its epilogue, at a barrier or waiting for a DMA, the other one feeds the matrix pipe.
//
// Geometry: 256 threads = 4 waves as 2(m) x 2(n); a wave owns 128 x 64 of the output = acc[8][4] (the same wave
// tile as gemm256.hip, so 12 fragment reads feed 32 MFMAs).  K advances in tiles of 64 (128 B per row).
// LDS: 80 KiB per workgroup so that two fit the CU's 160 KiB: the A tile (256 rows, 32 KiB) is double buffered,
// the W tile (128 rows, 16 KiB) is SINGLE buffered and goes through registers: every wave copies its 8 W
// fragments of the K-tile into VGPRs, a barrier later the buffer is refilled by the DMA for the next K-tile.
//
//   per K-tile t:   wait DMA | barrier | DMA A(t+1) -> other A buffer | W fragments -> registers | barrier |
//                   DMA W(t+1) | 2 x { 8 A fragments, 32 MFMAs }
#include <stdlib.h>

#include "gemm_common.h"

using namespace pmgemm;

namespace {

constexpr int BM = 256, BN = 128, THREADS = 256;
constexpr int A_BYTES = BM * ROWB;                 // 32 KiB
constexpr int W_BYTES = BN * ROWB;                 // 16 KiB
constexpr int W_OFF = 2 * A_BYTES;
constexpr int LDS_BYTES = W_OFF + W_BYTES;         // 80 KiB
constexpr int KSTEP = ROWB / 2;                    // 64 bf16

template <int EPI, typename OutT>
__global__ __launch_bounds__(THREADS, 2) void gemm2b_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;

    int tm, tn;
    tile_of_block(xcd_remap(blockIdx.x, gridDim.x), p.M / BM, p.N / BN, p.chunk, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // DMA sources: wave w copies A rows [64w, 64w+64) and W rows [32w, 32w+32) of the tile, 8 rows (1 KiB) per
    // instruction; lane -> (row lane>>3, 16-B slot lane&7), the XOR swizzle is applied to the SOURCE slot
    const size_t lda_b = (size_t)p.lda * 2, ldw_b = (size_t)p.ldw * 2;
    const unsigned lswz = (unsigned)(((lane & 7) ^ ((lane >> 3) & 7)) << 4);
    const unsigned char* asrc = reinterpret_cast<const unsigned char*>(p.A) + ((size_t)m0 + wave * 64) * lda_b +
                                (size_t)(lane >> 3) * lda_b + lswz;
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.W) + ((size_t)n0 + wave * 32) * ldw_b +
                                (size_t)(lane >> 3) * ldw_b + lswz;
    unsigned char* adst = lds + wave * 64 * ROWB;
    unsigned char* wdst = lds + W_OFF + wave * 32 * ROWB;

#define ISSUE_A(kt)                                                                                           \
    {                                                                                                         \
        const unsigned char* s_ = asrc + (size_t)(kt) * ROWB;                                                 \
        unsigned char* d_ = adst + ((kt) & 1) * A_BYTES;                                                      \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) glds16(s_ + (size_t)i * 8 * lda_b, d_ + i * 8 * ROWB);  \
    }
#define ISSUE_W(kt)                                                                                           \
    {                                                                                                         \
        const unsigned char* s_ = wsrc + (size_t)(kt) * ROWB;                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) glds16(s_ + (size_t)i * 8 * ldw_b, wdst + i * 8 * ROWB); \
    }

    // fragment read addresses (LDS byte addresses): row l15 (mod 8) fixes the swizzled slot, the fragment index is
    // an immediate offset
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned fa0 = lds_base + (unsigned)((wm * 128 + l15) * ROWB) + ((unsigned)((0 + g) ^ (l15 & 7)) << 4);
    const unsigned fa1 = lds_base + (unsigned)((wm * 128 + l15) * ROWB) + ((unsigned)((4 + g) ^ (l15 & 7)) << 4);
    const unsigned fw0 = lds_base + (unsigned)(W_OFF + (wn * 64 + l15) * ROWB) + ((unsigned)((0 + g) ^ (l15 & 7)) << 4);
    const unsigned fw1 = lds_base + (unsigned)(W_OFF + (wn * 64 + l15) * ROWB) + ((unsigned)((4 + g) ^ (l15 & 7)) << 4);

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // inline-asm LDS reads: hipcc would put `s_waitcnt vmcnt(0)` in front of every LDS read while a DMA is in flight
#define DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LGKM0 asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0)
#define RD_A(half)                                                                                  \
    DSR(a[0][0], ba0, ((half) * 64 + 0) * ROWB);  DSR(a[0][1], ba1, ((half) * 64 + 0) * ROWB);      \
    DSR(a[1][0], ba0, ((half) * 64 + 16) * ROWB); DSR(a[1][1], ba1, ((half) * 64 + 16) * ROWB);     \
    DSR(a[2][0], ba0, ((half) * 64 + 32) * ROWB); DSR(a[2][1], ba1, ((half) * 64 + 32) * ROWB);     \
    DSR(a[3][0], ba0, ((half) * 64 + 48) * ROWB); DSR(a[3][1], ba1, ((half) * 64 + 48) * ROWB);
#define MMA(half)                                                                                   \
    {                                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                              \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int f = 0; f < 4; ++f) \
            _Pragma("unroll") for (int h = 0; h < 4; ++h)                                           \
                Mma<bf16_t>::run(acc[(half) * 4 + f][h], w[h][kk], a[f][kk]);                       \
        __builtin_amdgcn_s_setprio(0);                                                              \
    }

#ifdef PM_ABL_NO_KLOOP
    const int nk = 1;
#else
    const int nk = p.K / KSTEP;
#endif
    ISSUE_A(0)
    ISSUE_W(0)
    for (int kt = 0; kt < nk; ++kt) {
        uint4 a[4][2], w[4][2];
        const unsigned ba0 = fa0 + (unsigned)(kt & 1) * A_BYTES, ba1 = fa1 + (unsigned)(kt & 1) * A_BYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // A(kt), W(kt) visible; nobody still reads the other A buffer
        if (kt + 1 < nk) ISSUE_A(kt + 1)
        DSR(w[0][0], fw0, 0 * ROWB);  DSR(w[0][1], fw1, 0 * ROWB);
        DSR(w[1][0], fw0, 16 * ROWB); DSR(w[1][1], fw1, 16 * ROWB);
        DSR(w[2][0], fw0, 32 * ROWB); DSR(w[2][1], fw1, 32 * ROWB);
        DSR(w[3][0], fw0, 48 * ROWB); DSR(w[3][1], fw1, 48 * ROWB);
        RD_A(0)
        LGKM0;
        __builtin_amdgcn_s_barrier();                      // every wave holds W(kt) in registers
        if (kt + 1 < nk) ISSUE_W(kt + 1)
        MMA(0)
        RD_A(1)
        LGKM0;
        MMA(1)
    }
    __builtin_amdgcn_s_barrier();                          // all fragment reads done: LDS is free for the epilogue
#undef DSR
#undef LGKM0
#undef RD_A
#undef MMA
#undef ISSUE_A
#undef ISSUE_W

    const float4 no_pre[1] = {};
    unsigned char* eraw = lds + wave * EPI_WAVE_BYTES;
    if constexpr (EPI == EPI_STD && sizeof(OutT) == 4) {
        if (p.residual) wave_epilogue<EPI, OutT, 8, 1, true, 1>(p, acc, eraw, m0 + wm * 128, n0 + wn * 64, lane, no_pre);
        else wave_epilogue<EPI, OutT, 8, 1, true, 0>(p, acc, eraw, m0 + wm * 128, n0 + wn * 64, lane, no_pre);
    } else if constexpr (EPI == EPI_STD) {
        if (p.out_lo) wave_epilogue<EPI, OutT, 8, 1, true, 2>(p, acc, eraw, m0 + wm * 128, n0 + wn * 64, lane, no_pre);   // bf16 hi/lo residual stream
        else wave_epilogue<EPI, OutT, 8, 1, true, 0>(p, acc, eraw, m0 + wm * 128, n0 + wn * 64, lane, no_pre);
    } else {
        wave_epilogue<EPI, OutT, 8, 1, true, 0>(p, acc, eraw, m0 + wm * 128, n0 + wn * 64, lane, no_pre);
    }
}

template <int EPI, typename OutT>
int launch2b(const GemmParams& p0, hipStream_t s) {
    // PMHIP_CHUNK2B (development builds, common.h pm_dev_knob): read once, in the thread-safe initialiser of a function-local static
    static const int g_chunk2b = pm_dev_knob("PMHIP_CHUNK2B", 12);
    GemmParams p = p0;
    p.chunk = g_chunk2b;
    const int tiles = (p.M / BM) * (p.N / BN);
    PmTimer tm(gemm_family(p, EPI, true), s);
    hipLaunchKernelGGL((gemm2b_kernel<EPI, OutT>), dim3(tiles), dim3(THREADS), 0, s, p);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

}  // namespace

int pm_gemm2b_supported(const GemmParams& p, int dtype, int epi, int out_dtype) {
    if (dtype != PMHIP_BF16) return 0;
    if (p.M % BM || p.N % BN || p.K % KSTEP) return 0;
    if ((p.M / BM) * (p.N / BN) < 192) return 0;
    (void)epi; (void)out_dtype;
    return 1;
}

int pm_gemm2b_launch(const GemmParams& p, int epi, int out_dtype, hipStream_t s) {
    if (epi == EPI_SWIGLU) return launch2b<EPI_SWIGLU, bf16_t>(p, s);
    if (epi == EPI_HEADS) return launch2b<EPI_HEADS, bf16_t>(p, s);
    if (out_dtype == PMHIP_F32) return launch2b<EPI_STD, float>(p, s);
    return launch2b<EPI_STD, bf16_t>(p, s);
}
