// 256x256 phase-staggered bf16 NT GEMM for gfx950 (the large-shape member of the family in gemm.hip).
//
// Geometry: 512 threads = 8 waves as 2(m) x 4(n); a wave owns 128 x 64 of the output = acc[8][4] MFMA tiles
// (128 accumulator registers), so every 16-byte fragment read from LDS feeds 4-8 MFMAs and every byte DMA'd
// into LDS feeds 4x the MFMAs of the 128x128 kernel.  K advances in tiles of 64; LDS holds two K-tile buffers of
// 64 KiB, one workgroup per CU.
//
// LDS image of a K-tile: four 16 KiB PIECES, [k-half h][operand A | W][256 rows][64 B] -- one piece is one operand's
// 32 k-values, exactly the K of one v_mfma_f32_16x16x32_bf16.  The 16-byte slot of (row, k-chunk g) is
// g ^ perm[(row >> 2) & 3], perm = {0,3,2,1}: the 16 lanes the LDS serves together for a ds_read_b128 then touch 16
// different 16-byte bank groups (rows are 64 B, so four rows share a 256-byte bank line).  The swizzle is applied on the
// DMA's per-lane SOURCE address (the LDS side of global_load_lds is lane-linear).
//
// Schedule.  A K-tile is consumed in 2 phases, one k-half each: 12 fragment reads (8 A + 4 W) and 32 independent MFMAs
// (one per accumulator tile).  The next K-tile's pieces of the same k-half are issued in that phase (2 pieces = 4 DMA
// instructions per wave), so a piece is needed 2 phases after it was issued and loads stay in flight across barriers:
// every wait is `s_waitcnt vmcnt(4)` (this phase's pieces may still be flying), never 0, except in the last K-tile.
// A piece is read one barrier after the wait that retired it and overwritten 2 phases after its last read.
// Each phase is two barrier-separated slots.  Waves with wm = 0 do {fragment reads + DMA issue | MFMAs}; waves with
// wm = 1 run one slot behind, {MFMAs of the previous phase | fragment reads + DMA issue}.  A SIMD hosts one wave of each
// kind, so its matrix pipe and the LDS / DMA issue are busy in the same slot instead of alternating.  A read slot does
// not wait for its reads: their latency runs under the DMA issue and the barrier, the lgkmcnt(0) heads the compute slot.
// (Round-2 history: 4 phases of 16 MFMAs per K-tile, i.e. 8 barriers, ran at 1.25 PFLOP/s on 8192^3; moving the
// lgkmcnt behind the barrier 1.31; DMA issue behind the MFMAs instead of behind the reads 1.24-1.28.)
#include <stdlib.h>

#include "gemm_common.h"

using namespace pmgemm;

namespace {

constexpr int BM = 256, BN = 256, THREADS = 512;
constexpr int PIECE_BYTES = 256 * 64;              // one operand's k-half: 16 KiB
constexpr int HALF_BYTES = 2 * PIECE_BYTES;        // A + W of one k-half
constexpr int BUF_BYTES = 2 * HALF_BYTES;          // one K-tile
constexpr int KSTEP = 64;                          // bf16 per K-tile

// slot swizzle: perm[(row >> 2) & 3]
__device__ __forceinline__ unsigned slot_perm(unsigned rowblk) { return (0x1230u >> (4 * (rowblk & 3))) & 3u; }   // {0,3,2,1}

// One k-half of the next K-tile: piece A then piece W, 2 DMA instructions per wave each (16 rows x 64 B = 1 KiB).  The
// source address splits into a wave-uniform part (operand base at the tile origin, chunk row, k offset) and ONE per-lane
// byte offset per operand (row-in-chunk * ld + swizzled 16-B chunk), so no per-piece address VGPRs are kept alive.
__device__ __forceinline__ void issue_half(int h, const unsigned char* __restrict__ Ab, const unsigned char* __restrict__ Wb,
                                           size_t lda_b, size_t ldw_b, unsigned laneoffA, unsigned laneoffW, int k0,
                                           unsigned char* buf, int wave) {
    unsigned char* dst = buf + h * HALF_BYTES;
    const size_t kb = (size_t)(k0 + h * 32) * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row0 = (wave * 2 + i) * 16;                      // wave-uniform
        glds16(Ab + (size_t)row0 * lda_b + kb + laneoffA, dst + row0 * 64);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row0 = (wave * 2 + i) * 16;
        glds16(Wb + (size_t)row0 * ldw_b + kb + laneoffW, dst + PIECE_BYTES + row0 * 64);
    }
}

struct LoopCtx {
    const unsigned char* Ab; const unsigned char* Wb;   // operand bases at the tile origin rows
    size_t lda_b, ldw_b;
    unsigned laneoffA, laneoffW;                        // per-lane DMA source offsets
    unsigned fa, fw;                                    // per-lane fragment read offsets inside a k-half
    unsigned lds_base;                                  // LDS byte address of the staging buffers
    int nk, wave;
};

// The K loop for one wave.  LEAD = true: waves with wm = 0 ({reads | MFMAs}); LEAD = false: waves with wm = 1, one
// slot behind ({MFMAs of the previous phase | reads}).  Both versions execute the SAME sequence of barriers, DMA
// issues and vmcnt waits; they are separate straight-line loops so that no register is merged across roles.
template <bool LEAD>
__device__ __forceinline__ void k_loop(const LoopCtx& c, unsigned char* lds, f32x4_t (&acc)[8][4], int pb) {
    uint4 a[8], w[4];
    // Fragment reads are inline-asm ds_read_b128: hipcc would otherwise put `s_waitcnt vmcnt(0)` in front of every
    // LDS read while a DMA (an LDS write on the VM counter) is in flight and drain the pipeline each phase.  The
    // counted waits + barriers below are what orders a read after the DMA that produced its data.  Offsets are
    // literal immediates (the "n" constraint needs constants, hence the macro expansion).
#define DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define RD(h)                                                                                    \
    {                                                                                            \
        const unsigned ba = c.lds_base + boff + (h) * HALF_BYTES + c.fa, bw = c.lds_base + boff + (h) * HALF_BYTES + c.fw; \
        DSR(w[0], bw, 0 * 1024); DSR(w[1], bw, 1 * 1024); DSR(w[2], bw, 2 * 1024); DSR(w[3], bw, 3 * 1024); \
        DSR(a[0], ba, 0 * 1024); DSR(a[1], ba, 1 * 1024); DSR(a[2], ba, 2 * 1024); DSR(a[3], ba, 3 * 1024); \
        DSR(a[4], ba, 4 * 1024); DSR(a[5], ba, 5 * 1024); DSR(a[6], ba, 6 * 1024); DSR(a[7], ba, 7 * 1024); \
    }
#define MMA()                                                                                    \
    {                                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                           \
        _Pragma("unroll") for (int f = 0; f < 8; ++f) _Pragma("unroll") for (int j = 0; j < 4; ++j) \
            Mma<bf16_t>::run(acc[f][j], w[j], a[f]);                                             \
        __builtin_amdgcn_s_setprio(0);                                                           \
    }
    // the wait is invisible to the scheduler too: pin everything behind it (cdna_hip_programming.md 5.4 rule 18)
#define LGKM0 asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0)
#define BAR __builtin_amdgcn_s_barrier()
#define ISSUE(h) \
    if (has_next) issue_half(h, c.Ab, c.Wb, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, (kt + 1) * KSTEP, nxt, c.wave);
    // end of k-half h: the pieces of the NEXT phase must have landed.  Steady state: everything but this phase's own 4
    // DMA instructions; last K-tile: after h = 0 its own second half (the 4 youngest), after h = 1 nothing is in flight
#define WAIT(h)                                                                                  \
    if (has_next) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                               \
    else if ((h) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    for (int kt = 0; kt < c.nk; ++kt) {
        const bool has_next = kt + 1 < c.nk;
        const unsigned boff = (unsigned)((kt + pb) & 1) * BUF_BYTES;   // pb: buffer of K-tile 0
        unsigned char* nxt = lds + ((kt + 1 + pb) & 1) * BUF_BYTES;
        // ---------------- k-half 0
        if constexpr (LEAD) { RD(0) ISSUE(0) } else { if (kt > 0) { LGKM0; MMA() } }
        BAR;
        if constexpr (LEAD) { LGKM0; MMA() } else { RD(0) ISSUE(0) }
        WAIT(0)
        BAR;
        // ---------------- k-half 1
        if constexpr (LEAD) { RD(1) ISSUE(1) } else { LGKM0; MMA() }
        BAR;
        if constexpr (LEAD) { LGKM0; MMA() } else { RD(1) ISSUE(1) }
        WAIT(1)
        BAR;
    }
    if constexpr (!LEAD) { LGKM0; MMA() }
#undef DSR
#undef RD
#undef MMA
#undef LGKM0
#undef BAR
#undef ISSUE
#undef WAIT
}

// FOLD: LayerNorm folded into this GEMM (gemm_common.h): A is the raw bf16 residual row, the epilogue normalises.
template <int EPI, typename OutT, bool FOLD = false>
__global__ __launch_bounds__(THREADS) void gemm256_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * BUF_BYTES + (FOLD ? 8 * 2048 : 0)];   // + 2 KiB per wave for the fold

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, g = lane >> 4;

    // PERSISTENT: the grid holds at most one workgroup per CU; workgroup w walks tiles w, w + grid, ... of the same XCD /
    // L2-aware order.  While a tile's epilogue runs, the first K-tile of the NEXT tile is already in flight (issued from
    // the epilogue's hook into the LDS buffer the epilogue does not stage through), so only the first tile of a
    // workgroup pays the DMA latency of its prologue, and a tile's stores drain under the next tile's K loop.
    const int ntiles = (p.M / BM) * (p.N / BN);
    const int tiles_m = p.M / BM, tiles_n = p.N / BN;
    LoopCtx c;
    c.lda_b = (size_t)p.lda * 2; c.ldw_b = (size_t)p.ldw * 2;
    // DMA: lane -> (row-in-chunk lane >> 2, LDS slot lane & 3); chunk rows start at multiples of 16, so the row block of
    // the swizzle is (lane >> 4) & 3 and the k-chunk this lane must fetch for its slot is slot ^ perm (an involution)
    const unsigned lswz = (((unsigned)lane & 3u) ^ slot_perm((unsigned)lane >> 4)) << 4;
    c.laneoffA = (unsigned)(lane >> 2) * (unsigned)c.lda_b + lswz;
    c.laneoffW = (unsigned)(lane >> 2) * (unsigned)c.ldw_b + lswz;
    // fragment addresses: every fragment row is l15 (mod 16), so the swizzled slot depends on the lane only; the
    // fragment index is a compile-time offset that folds into the ds_read immediate
    const unsigned fslot = ((unsigned)g ^ slot_perm((unsigned)l15 >> 2)) << 4;
    c.fa = (unsigned)((wm * 128 + l15) * 64) + fslot;
    c.fw = (unsigned)(PIECE_BYTES + (wn * 64 + l15) * 64) + fslot;
    c.nk = p.K / KSTEP;
    c.wave = wave;
    c.lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;

    auto origin = [&](int tile, int& m0, int& n0) {
        int tm, tn;
        tile_of_block(xcd_remap(tile, ntiles), tiles_m, tiles_n, p.chunk, tm, tn);
        m0 = tm * BM; n0 = tn * BN;
    };
    int tile = blockIdx.x, m0, n0;
    origin(tile, m0, n0);
    c.Ab = reinterpret_cast<const unsigned char*>(p.A) + (size_t)m0 * c.lda_b;
    c.Wb = reinterpret_cast<const unsigned char*>(p.W) + (size_t)n0 * c.ldw_b;

    // prologue of the FIRST tile: its whole first K-tile, in need-order; the first k-half must have landed before phase 1
    LnLoads lnl;
    if constexpr (FOLD) ln_stats_issue(p, m0 + wm * 128, n0 + wn * 64, lane, lnl);      // before the DMA: these return first
    issue_half(0, c.Ab, c.Wb, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, lds, wave);
    issue_half(1, c.Ab, c.Wb, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, lds, wave);
    float fa[FOLD ? 8 : 1], fb[FOLD ? 8 : 1];
    float* fscr = reinterpret_cast<float*>(lds + 2 * BUF_BYTES + (FOLD ? wave * 2048 : 0));
    if constexpr (FOLD) ln_row_coeffs<8>(p, lane, fscr, lnl, fa, fb);        // 4 pieces x 2 DMA instructions stay in flight
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int pb = 0;                                    // LDS buffer that holds K-tile 0 of the current tile
    for (;;) {
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        if (wm == 0) k_loop<true>(c, lds, acc, pb); else k_loop<false>(c, lds, acc, pb);

        // every fragment read finished before the last barrier: both LDS buffers are free.  The epilogue stages through
        // buffer 0 (8 KiB per wave); the next tile's first K-tile goes to buffer 1.
        const int next = FOLD ? ntiles : tile + (int)gridDim.x;       // the fold variant is launched one tile per workgroup
        const bool has_next = next < ntiles;
        int m1 = 0, n1 = 0;
        if (has_next) origin(next, m1, n1);
        const unsigned char* Ab1 = reinterpret_cast<const unsigned char*>(p.A) + (size_t)m1 * c.lda_b;
        const unsigned char* Wb1 = reinterpret_cast<const unsigned char*>(p.W) + (size_t)n1 * c.ldw_b;
        auto prefetch = [&]() {
            if (has_next) {
                unsigned char* b1 = lds + BUF_BYTES;
                issue_half(0, Ab1, Wb1, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, b1, wave);
                issue_half(1, Ab1, Wb1, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, b1, wave);
            }
        };
        if constexpr (FOLD) ln_apply<8>(fscr, acc, lane, fa, fb);
        const float4 no_pre[1] = {};
        unsigned char* eraw = lds + wave * EPI_WAVE_BYTES;
        const int mw = m0 + wm * 128, nw = n0 + wn * 64;
        if constexpr (EPI == EPI_STD && sizeof(OutT) == 4) {
            // residual variants load through the whole epilogue (pipelined residual rows): no early prefetch there
            if (p.residual && p.xb_out) { wave_epilogue<EPI, OutT, 8, 1, true, 1, true>(p, acc, eraw, mw, nw, lane, no_pre); prefetch(); }
            else if (p.residual) { wave_epilogue<EPI, OutT, 8, 1, true, 1>(p, acc, eraw, mw, nw, lane, no_pre); prefetch(); }
            else wave_epilogue<EPI, OutT, 8, 1, true, 0, false>(p, acc, eraw, mw, nw, lane, no_pre, prefetch);
        } else {
            wave_epilogue<EPI, OutT, 8, 1, true, 0, false>(p, acc, eraw, mw, nw, lane, no_pre, prefetch);
        }
        if (!has_next) break;
        // the 8 DMA instructions of the prefetch are older than every store issued after the hook (>= 8 per wave in every
        // epilogue), and vmcnt retires in issue order: with at most 4 operations left in flight those are stores, i.e. the
        // whole K-tile has landed, while the youngest stores keep draining under the next K loop
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // every wave is done staging through buffer 0; the K-tile is visible
        tile = next; m0 = m1; n0 = n1;
        c.Ab = Ab1; c.Wb = Wb1;
        pb = 1;
    }
}

int g_chunk256 = -1;

template <int EPI, typename OutT>
int launch256(const GemmParams& p0, hipStream_t s) {
    if (g_chunk256 < 0) {
        const char* e = getenv("PMHIP_CHUNK256");
        g_chunk256 = e ? atoi(e) : 6;
    }
    GemmParams p = p0;
    p.chunk = g_chunk256;
    const int tiles = (p.M / BM) * (p.N / BN);
    static int persist = -1;                     // PMHIP_PERSIST256: workgroups of the persistent grid (0 = one per tile)
    if (persist < 0) { const char* e = getenv("PMHIP_PERSIST256"); persist = e ? atoi(e) : 256; }
    const int grid = (persist > 0 && tiles > persist) ? persist : tiles;
    PmTimer tm(FAM_GEMM, s);
    if (p.ln_stats) hipLaunchKernelGGL((gemm256_kernel<EPI, OutT, true>), dim3(tiles), dim3(THREADS), 0, s, p);
    else hipLaunchKernelGGL((gemm256_kernel<EPI, OutT, false>), dim3(grid), dim3(THREADS), 0, s, p);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

}  // namespace

int pm_gemm256_supported(const GemmParams& p, int dtype, int epi, int out_dtype) {
    if (dtype != PMHIP_BF16) return 0;
    if (p.M % BM || p.N % BN || p.K % KSTEP) return 0;
    if ((p.M / BM) * (p.N / BN) < 96) return 0;             // too few tiles: the small kernel has 4x the workgroups
    if (epi == EPI_STD && p.residual) {
        // residual GEMMs: the 128x128 kernel prefetches the residual tile and overlaps two workgroups per CU, which wins
        // while the GEMM is HBM-bound (small K); with a long K loop the faster main loop of this kernel wins
        static int kmin = -1;
        if (kmin < 0) { const char* e = getenv("PMHIP_G256_RES_KMIN"); kmin = e ? atoi(e) : 1024; }
        if (p.K < kmin) return 0;
    }
    (void)out_dtype;
    return 1;
}

int pm_gemm256_launch(const GemmParams& p, int epi, int out_dtype, hipStream_t s) {
    if (epi == EPI_SWIGLU) return launch256<EPI_SWIGLU, bf16_t>(p, s);
    if (epi == EPI_HEADS) return launch256<EPI_HEADS, bf16_t>(p, s);
    if (out_dtype == PMHIP_F32) return launch256<EPI_STD, float>(p, s);
    return launch256<EPI_STD, bf16_t>(p, s);
}
