#define ABL_NOBAR 1
#define ABL_NODMA 1
// 256x256 phase-staggered bf16 NT GEMM for gfx950 (the large-shape member of the family in gemm.hip).
//
// Geometry: 512 threads = 8 waves as 2(m) x 4(n); a wave owns 128 x 64 of the output = acc[8][4] MFMA tiles
// (128 accumulator registers), so every 16-byte fragment read from LDS feeds 4-8 MFMAs and every byte DMA'd
// into LDS feeds 4x the MFMAs of the 128x128 kernel.  K advances in tiles of 64 (128 B per row); LDS holds two
// K-tile buffers of 64 KiB (A 256 x 128 B + W 256 x 128 B), one workgroup per CU.
//
// Schedule.  A K-tile is consumed in 4 phases, one 64 x 32 output quadrant of the wave each (16 MFMAs):
//     phase 1: A rows [0,64)  x W rows [0,32)      reads 8 A + 4 W fragments
//     phase 2: A rows [0,64)  x W rows [32,64)     reads 4 W
//     phase 3: A rows [64,128) x W rows [32,64)    reads 8 A
//     phase 4: A rows [64,128) x W rows [0,32)     reads 4 W (again: cheaper than 16 more live registers)
// The next K-tile arrives as four 16 KiB DMA pieces (global_load_lds, 2 per thread), one issued per phase in
// the order it will be needed: PA0 (A rows for phase 1), PW0, PW1, PA1.  A piece is needed 3 phases after it was
// issued, so loads stay in flight across barriers: every wait is `s_waitcnt vmcnt(4)` (the two youngest pieces may
// still be flying), never 0, except in the last K-tile.  A piece is read one barrier after the wait that
// retired it, and overwritten >= 4 phases after its last read.
// Each phase is two barrier-separated slots.  Waves with wm = 0 do {fragment reads | MFMAs}; waves with wm = 1 run
// one slot behind, {MFMAs of the previous phase | fragment reads}.  A SIMD hosts one wave of each kind, so its
// matrix pipe and the LDS pipe are busy in the same slot instead of alternating.  DMA issue (even slots) and
// vmcnt waits (odd slots) are slot-aligned for all waves, which keeps the vmcnt arithmetic identical.
#include <stdlib.h>

#include "gemm_common.h"

#ifndef ABL_NODMA
#define ABL_NODMA 0
#endif
#ifndef ABL_NOREAD
#define ABL_NOREAD 0
#endif
#ifndef ABL_NOBAR
#define ABL_NOBAR 0
#endif
using namespace pmgemm;

namespace {

constexpr int BM = 256, BN = 256, THREADS = 512;
constexpr int OPER_BYTES = 256 * ROWB;             // one operand's K-tile: 32 KiB
constexpr int BUF_BYTES = 2 * OPER_BYTES;          // A + W
constexpr int KSTEP = ROWB / 2;                    // 64 bf16

enum { PA0 = 0, PW0 = 1, PW1 = 2, PA1 = 3 };

// first tile row of 8-row chunk c (0..15) of piece `piece`
__device__ __forceinline__ int piece_row(int piece, int c) {
    const int pr = c * 8;                                          // row inside the 128-row piece
    if (piece == PA0) return pr < 64 ? pr : 128 + (pr - 64);
    if (piece == PA1) return pr < 64 ? 64 + pr : 192 + (pr - 64);
    const int grp = pr >> 5, within = pr & 31;
    return grp * 64 + within + (piece == PW1 ? 32 : 0);
}

// A piece = 16 chunks of 8 rows; wave w DMAs chunks 2w, 2w+1.  The source address splits into a wave-uniform
// part (operand base, tile origin, chunk row, k offset: SGPRs) and ONE per-lane byte offset per operand
// (row-in-chunk * ld + swizzled 16-B slot), so no per-piece address VGPRs are kept alive.
__device__ __forceinline__ void issue_piece(int piece, const unsigned char* __restrict__ Ab, const unsigned char* __restrict__ Wb,
                                            size_t lda_b, size_t ldw_b, unsigned laneoffA, unsigned laneoffW, int k0,
                                            unsigned char* buf, int wave, int i0 = 0, int i1 = 2) {
    const bool isA = (piece == PA0 || piece == PA1);
    const unsigned char* base = isA ? Ab : Wb;                     // already offset to the tile origin row
    const size_t ld_b = isA ? lda_b : ldw_b;
    const unsigned laneoff = isA ? laneoffA : laneoffW;
    unsigned char* tile = buf + (isA ? 0 : OPER_BYTES);
#pragma unroll
    for (int i = i0; i < i1; ++i) {
        const int row0 = piece_row(piece, wave * 2 + i);           // wave-uniform
        const unsigned char* sbase = base + (size_t)row0 * ld_b + (size_t)k0 * 2;
        glds16(sbase + laneoff, tile + row0 * ROWB);
    }
}

struct LoopCtx {
    const unsigned char* Ab; const unsigned char* Wb;   // operand bases at the tile origin rows
    size_t lda_b, ldw_b;
    unsigned laneoffA, laneoffW;                        // per-lane DMA source offsets
    unsigned fa0, fa1, fw0, fw1;                        // per-lane fragment read offsets (kk = 0, 1)
    unsigned lds_base;                                  // LDS byte address of the staging buffers
    int nk, wave;
};

// The K loop for one wave.  LEAD = true: waves with wm = 0 ({reads | MFMAs}); LEAD = false: waves with wm = 1, one
// slot behind ({MFMAs of the previous phase | reads}).  Both versions execute the SAME sequence of barriers, DMA
// issues and vmcnt waits; they are separate straight-line loops so that no register is merged across roles.
template <bool LEAD>
__device__ __forceinline__ void k_loop(const LoopCtx& c, unsigned char* lds, f32x4_t (&acc)[8][4], int pb) {
    uint4 a[4][2] = {}, w[2][2] = {};                        // [fragment][kk]
    // Fragment reads are inline-asm ds_read_b128: hipcc would otherwise put `s_waitcnt vmcnt(0)` in front of every
    // LDS read while a DMA (an LDS write on the VM counter) is in flight and drain the pipeline each phase.  The
    // counted waits + barriers below are what orders a read after the DMA that produced its data.  Offsets are
    // literal immediates (the "n" constraint needs constants, hence the macro expansion).
#if ABL_NOREAD
#define DSR(dst, addr, off) asm volatile("; no read %0 %1" : "=v"(dst) : "v"(addr))
#else
#define DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#endif
#define RD_A(half)                                                  \
    DSR(a[0][0], ba0, ((half) * 64 + 0) * ROWB);  DSR(a[0][1], ba1, ((half) * 64 + 0) * ROWB);  \
    DSR(a[1][0], ba0, ((half) * 64 + 16) * ROWB); DSR(a[1][1], ba1, ((half) * 64 + 16) * ROWB); \
    DSR(a[2][0], ba0, ((half) * 64 + 32) * ROWB); DSR(a[2][1], ba1, ((half) * 64 + 32) * ROWB); \
    DSR(a[3][0], ba0, ((half) * 64 + 48) * ROWB); DSR(a[3][1], ba1, ((half) * 64 + 48) * ROWB);
#define RD_W(half)                                                  \
    DSR(w[0][0], bw0, ((half) * 32 + 0) * ROWB);  DSR(w[0][1], bw1, ((half) * 32 + 0) * ROWB);  \
    DSR(w[1][0], bw0, ((half) * 32 + 16) * ROWB); DSR(w[1][1], bw1, ((half) * 32 + 16) * ROWB);
#define MMA(mhalf, nhalf)                                                                        \
    {                                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                           \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int f = 0; f < 4; ++f) \
            _Pragma("unroll") for (int h = 0; h < 2; ++h)                                        \
                Mma<bf16_t>::run(acc[(mhalf) * 4 + f][(nhalf) * 2 + h], w[h][kk], a[f][kk]);     \
        __builtin_amdgcn_s_setprio(0);                                                           \
    }
    // the wait is invisible to the scheduler too: pin everything behind it (cdna_hip_programming.md 5.4 rule 18)
#define LGKM0 asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0)
#if ABL_NOBAR
#define BAR __builtin_amdgcn_sched_barrier(0)
#else
#define BAR __builtin_amdgcn_s_barrier()
#endif
#ifndef PM_DMA_SPLIT
#define PM_DMA_SPLIT 0
#endif
    // The two DMA instructions of a wave's piece: PM_DMA_SPLIT of them go behind the MFMAs of the wave's compute slot, the
    // rest behind the fragment reads of its read slot.  ISSUE_A / ISSUE_B = first / second slot of the phase (lead: read
    // slot then compute slot; lag: compute slot then read slot).
#define ISSUE_RANGE(piece, lo, hi) \
    if (has_next && (lo) < (hi) && !ABL_NODMA) issue_piece(piece, c.Ab, c.Wb, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, (kt + 1) * KSTEP, nxt, c.wave, lo, hi);
#define ISSUE_A(piece) ISSUE_RANGE(piece, 0, (LEAD ? 2 - PM_DMA_SPLIT : PM_DMA_SPLIT))
#define ISSUE_B(piece) ISSUE_RANGE(piece, (LEAD ? 2 - PM_DMA_SPLIT : PM_DMA_SPLIT), 2)
#define WAIT(last_n)                                                                             \
    if (has_next) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                               \
    else if ((last_n) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                     \
    else if ((last_n) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    for (int kt = 0; kt < c.nk; ++kt) {
        const bool has_next = kt + 1 < c.nk;
        const unsigned boff = (unsigned)((kt + pb) & 1) * BUF_BYTES;   // LDS byte addresses of this K-tile's fragments (pb: buffer of K-tile 0)
        const unsigned ba0 = c.lds_base + boff + c.fa0, ba1 = c.lds_base + boff + c.fa1;
        const unsigned bw0 = c.lds_base + boff + c.fw0, bw1 = c.lds_base + boff + c.fw1;
        unsigned char* nxt = lds + ((kt + 1 + pb) & 1) * BUF_BYTES;
        // Every wave issues its DMA piece in ITS fragment-read slot (lead: first slot, lag: second slot), i.e. while the
        // other wave of the SIMD runs MFMAs: a DMA instruction holds the issuing wave for ~60 cycles, which in front of
        // the lag wave's MFMAs (with the lead wave busy reading) would idle the matrix pipe.  The piece still precedes
        // the phase's WAIT, so the vmcnt arithmetic is the same for both roles.
        // ---------------- phase 1: A rows [0,64) x W rows [0,32)
        // A reader slot is {fragment reads, DMA issue} with NO wait: the reads' LDS latency runs under the DMA issue and
        // the barrier, and the lgkmcnt(0) sits at the head of the wave's NEXT slot, right before the MFMAs that consume them.
        // Within a phase the order of a wave's two DMA instructions and the phase's WAIT is the same for both roles.
        if constexpr (LEAD) { RD_A(0) RD_W(0) ISSUE_A(PA0) } else { if (kt > 0) { LGKM0; MMA(1, 0) } ISSUE_A(PA0) }
        BAR;
        if constexpr (LEAD) { LGKM0; MMA(0, 0) ISSUE_B(PA0) } else { RD_A(0) RD_W(0) ISSUE_B(PA0) }
        WAIT(2)
        BAR;
        // ---------------- phase 2: A rows [0,64) x W rows [32,64)
        if constexpr (LEAD) { RD_W(1) ISSUE_A(PW0) } else { LGKM0; MMA(0, 0) ISSUE_A(PW0) }
        BAR;
        if constexpr (LEAD) { LGKM0; MMA(0, 1) ISSUE_B(PW0) } else { RD_W(1) ISSUE_B(PW0) }
        WAIT(0)
        BAR;
        // ---------------- phase 3: A rows [64,128) x W rows [32,64)
        if constexpr (LEAD) { RD_A(1) ISSUE_A(PW1) } else { LGKM0; MMA(0, 1) ISSUE_A(PW1) }
        BAR;
        if constexpr (LEAD) { LGKM0; MMA(1, 1) ISSUE_B(PW1) } else { RD_A(1) ISSUE_B(PW1) }
        WAIT(-1)
        BAR;
        // ---------------- phase 4: A rows [64,128) x W rows [0,32) (re-read: cheaper than 16 more live registers)
        if constexpr (LEAD) { RD_W(0) ISSUE_A(PA1) } else { LGKM0; MMA(1, 1) ISSUE_A(PA1) }
        BAR;
        if constexpr (LEAD) { LGKM0; MMA(1, 0) ISSUE_B(PA1) } else { RD_W(0) ISSUE_B(PA1) }
        WAIT(-1)
        BAR;
    }
    if constexpr (!LEAD) { LGKM0; MMA(1, 0) }
#undef DSR
#undef RD_A
#undef RD_W
#undef MMA
#undef LGKM0
#undef BAR
#undef ISSUE_RANGE
#undef ISSUE_A
#undef ISSUE_B
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
    const unsigned lswz = (unsigned)(((lane & 7) ^ ((lane >> 3) & 7)) << 4);       // chunk rows start at multiples of 8
    c.laneoffA = (unsigned)(lane >> 3) * (unsigned)c.lda_b + lswz;
    c.laneoffW = (unsigned)(lane >> 3) * (unsigned)c.ldw_b + lswz;
    // fragment addresses: every fragment row is l15 (mod 8), so the swizzled slot depends on kk only; everything
    // else is a compile-time offset that folds into the ds_read immediate
    c.fa0 = (unsigned)((wm * 128 + l15) * ROWB) + ((unsigned)((0 + g) ^ (l15 & 7)) << 4);
    c.fa1 = (unsigned)((wm * 128 + l15) * ROWB) + ((unsigned)((4 + g) ^ (l15 & 7)) << 4);
    c.fw0 = (unsigned)(OPER_BYTES + (wn * 64 + l15) * ROWB) + ((unsigned)((0 + g) ^ (l15 & 7)) << 4);
    c.fw1 = (unsigned)(OPER_BYTES + (wn * 64 + l15) * ROWB) + ((unsigned)((4 + g) ^ (l15 & 7)) << 4);
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

    // prologue of the FIRST tile: its whole first K-tile, in need-order; PA0 + PW0 must have landed before phase 1
    LnLoads lnl;
    if constexpr (FOLD) ln_stats_issue(p, m0 + wm * 128, n0 + wn * 64, lane, lnl);      // before the DMA: these return first
    issue_piece(PA0, c.Ab, c.Wb, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, lds, wave);
    issue_piece(PW0, c.Ab, c.Wb, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, lds, wave);
    issue_piece(PW1, c.Ab, c.Wb, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, lds, wave);
    issue_piece(PA1, c.Ab, c.Wb, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, lds, wave);
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
                issue_piece(PA0, Ab1, Wb1, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, b1, wave);
                issue_piece(PW0, Ab1, Wb1, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, b1, wave);
                issue_piece(PW1, Ab1, Wb1, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, b1, wave);
                issue_piece(PA1, Ab1, Wb1, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, b1, wave);
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
