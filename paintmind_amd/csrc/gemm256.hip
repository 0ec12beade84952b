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
// retired it.  Overwrites: PA0 >= 3 phases, PW0 2 phases, PW1 / PA1 >= 4 phases after the rows' last read in a steady
// K-tile; the last K-tile of an output tile is the tight case and is handled explicitly (k_tile, phase 1): between the
// last read of a row and the DMA instruction that overwrites it there is always an s_waitcnt lgkmcnt(0) of the reading
// wave followed by a workgroup barrier.
// Each phase is two barrier-separated slots.  Waves with wm = 0 do {fragment reads | MFMAs}; waves with wm = 1 run
// one slot behind, {MFMAs of the previous phase | fragment reads}.  A SIMD hosts one wave of each kind, so its
// matrix pipe and the LDS pipe are busy in the same slot instead of alternating.  DMA issue (even slots) and
// vmcnt waits (odd slots) are slot-aligned for all waves, which keeps the vmcnt arithmetic identical.
#include <stdlib.h>

#include <mutex>

#include "gemm_common.h"

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

// A piece = 16 chunks of 8 rows; wave w DMAs chunks 2w, 2w+1 with `buffer_load_dwordx4 ... lds`: one buffer descriptor per
// operand for the whole kernel (SGPRs), the tile origin + chunk row + k offset form the scalar offset, and ONE per-lane byte
// offset per operand (row-in-chunk * ld + swizzled 16-B slot) is the vector offset -- a DMA instruction costs its wave
// two SALU operations and no VALU (the flat global_load_lds form needed a 64-bit VALU add per instruction, issued by the
// read-slot wave while its SIMD partner holds priority for MFMAs).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t tile_rsrc(const void* origin) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(origin), 0, 0x7fffffff, 0x00020000);   // raw buffer, no bounds in play
}
__device__ __forceinline__ void issue_piece(int piece, rsrc_t Ar, rsrc_t Wr, unsigned aorg, unsigned worg, unsigned lda_b, unsigned ldw_b,
                                            unsigned laneoffA, unsigned laneoffW, int k0, unsigned char* buf, int wave, int i0 = 0, int i1 = 2) {
    const bool isA = (piece == PA0 || piece == PA1);
    const unsigned org = isA ? aorg : worg;
    const unsigned ld_b = isA ? lda_b : ldw_b;
    const unsigned laneoff = isA ? laneoffA : laneoffW;
    unsigned char* tile = buf + (isA ? 0 : OPER_BYTES);
#pragma unroll
    for (int i = i0; i < i1; ++i) {
        const int row0 = piece_row(piece, wave * 2 + i);           // wave-uniform
        const unsigned soff = org + (unsigned)row0 * ld_b + (unsigned)k0 * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? Ar : Wr, (__attribute__((address_space(3))) void*)(tile + row0 * ROWB), 16, laneoff,
                                                 soff, 0, 0);
    }
}

struct LoopCtx {
    rsrc_t Ar, Wr;                                      // buffer descriptors of A / W
    unsigned aorg, worg;                                // byte offsets of the tile's origin rows in A / W
    unsigned aorg1, worg1;                              // the same for the workgroup's NEXT tile (streamed in by the last K-tile)
    unsigned lda_b, ldw_b;
    unsigned laneoffA, laneoffW;                        // per-lane DMA source offsets
    unsigned fa0, fa1, fw0, fw1;                        // per-lane fragment read offsets (kk = 0, 1)
    unsigned lds_base;                                  // LDS byte address of the staging buffers
    int nk, wave;
    int mw, nw, lane;                                   // first row / column of this wave in the current tile (LayerNorm fold)
};


// Fragment reads are inline-asm ds_read_b128: hipcc would otherwise put `s_waitcnt vmcnt(0)` in front of every
// LDS read while a DMA (an LDS write on the VM counter) is in flight and drain the pipeline each phase.  The
// counted waits + barriers below are what orders a read after the DMA that produced its data.  Offsets are
// literal immediates (the "n" constraint needs constants, hence the macro expansion).
#define DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
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
#define BAR __builtin_amdgcn_s_barrier()
#define VMW(n) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

// K-tile kinds.  They differ in what the tile's read slots DMA and in the phase-end vmcnt waits; the fragment reads, MFMAs
// and barriers are identical.  All of it is compile-time: with run-time flags hipcc merged the `if (has_next)` blocks of a
// phase and hoisted the phase-end vmcnt wait into the read slot, in front of the barrier and the MFMAs it was placed behind
// (and every extra variant on a run-time branch costs accumulator spills at the joins, hence exactly three kinds).
//   KT_STEADY  next K-tile of the same output tile, one piece per phase (PA0, PW0, PW1, PA1); every wait is vmcnt(4): the two
//              youngest pieces may still be flying
//   KT_FIRST   first K-tile of an output tile.  All four of its pieces have landed (prologue, or KT_LAST of the previous tile),
//              and the only younger VMEM operations are the previous epilogue's STORES: phases 1-3 do not wait at all, so the
//              stores drain under three phases of MFMAs; phase 4 needs the first two pieces of the next K-tile and therefore
//              (vmcnt retires in issue order: AMDGPU memory model, tools/hwtests/vmcnt_order.hip) the stores.  The lag wave
//              has no previous phase to multiply.
//   KT_LAST    the tile's last K-tile.  The first K-tile of the workgroup's NEXT output tile is DMA'd into the other buffer --
//              dead since the previous K-tile -- two pieces in each of phases 1 and 2, nothing afterwards, so the closing
//              vmcnt(0) waits for loads that are >= 2 phases old and the epilogue starts with no VMEM load of this wave in
//              flight.  Waits: phase 1 needs PW1 (outstanding: PA1 + 4 new = 6), phase 2 needs PA1 (8 new).  A workgroup
//              without a next tile re-reads its own tile's first K-tile (64 KiB from L2, once per workgroup) rather than
//              carrying a fourth variant.
enum { KT_STEADY = 0, KT_FIRST = 1, KT_LAST = 2 };

// A read slot is {fragment reads, DMA issue} with NO wait: the reads' LDS latency runs under the DMA issue and the
// barrier, and the lgkmcnt(0) heads the wave's NEXT slot, right before the MFMAs that consume them.  The DMA goes behind the
// reads of the wave's read slot, i.e. while the other wave of the SIMD runs MFMAs (behind the wave's own MFMAs it measured
// 2-5 % slower).  LEAD = true: waves with wm = 0 ({reads | MFMAs}); LEAD = false: waves with wm = 1, one slot behind
// ({MFMAs of the previous phase | reads}).  Both roles execute the SAME sequence of barriers, DMA issues and vmcnt waits.
template <bool LEAD, int KIND, bool FOLD>
__device__ __forceinline__ void k_tile(const GemmParams& p, const LoopCtx& c, unsigned char* lds, f32x4_t (&acc)[8][4], uint4 (&a)[4][2],
                                       uint4 (&w)[2][2], float* fscr, int kt, int pb) {
    // LayerNorm fold: the tile's per-row coefficients are DMA'd into the wave's scratch in the first read slot of its LAST
    // K-tile (LN_COEF_LOADS more operations in flight through phases 1 and 2, older than the phase's DMA) and are retired by
    // the closing vmcnt(0)
#ifdef PM_FOLD_DIRECT
    constexpr bool FOLD_DIRECT = true;
#else
    constexpr bool FOLD_DIRECT = false;
#endif
    constexpr int XL = (FOLD && !FOLD_DIRECT && KIND == KT_LAST) ? LN_COEF_LOADS : 0;
    constexpr bool FIRST = KIND == KT_FIRST;
    constexpr bool ONE_PER_PHASE = KIND != KT_LAST;
    const unsigned boff = (unsigned)((kt + pb) & 1) * BUF_BYTES;   // LDS byte addresses of this K-tile's fragments (pb: buffer of K-tile 0)
    const unsigned ba0 = c.lds_base + boff + c.fa0, ba1 = c.lds_base + boff + c.fa1;
    const unsigned bw0 = c.lds_base + boff + c.fw0, bw1 = c.lds_base + boff + c.fw1;
    unsigned char* nxt = lds + ((kt + 1 + pb) & 1) * BUF_BYTES;
#define ISSUE(piece) issue_piece(piece, c.Ar, c.Wr, c.aorg, c.worg, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, (kt + 1) * KSTEP, nxt, c.wave);
#define ISSUE_NEXT_TILE(piece) issue_piece(piece, c.Ar, c.Wr, c.aorg1, c.worg1, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, nxt, c.wave);
#define DMA(phase)                                                                                                   \
    if constexpr (ONE_PER_PHASE) {                                                                                   \
        if constexpr (phase == 1) { ISSUE(PA0) } else if constexpr (phase == 2) { ISSUE(PW0) }                       \
        else if constexpr (phase == 3) { ISSUE(PW1) } else { ISSUE(PA1) }                                            \
    } else {                                                                                                         \
        if constexpr (phase == 1 && FOLD) ln_coef_issue(p, c.mw, c.nw, c.lane, fscr);                                 \
        if constexpr (phase == 1) { ISSUE_NEXT_TILE(PA0) ISSUE_NEXT_TILE(PW0) }                                      \
        else if constexpr (phase == 2) { ISSUE_NEXT_TILE(PW1) ISSUE_NEXT_TILE(PA1) }                                 \
    }
#define WAIT(phase)                                                                                                  \
    if constexpr (KIND == KT_STEADY) VMW(4)                                                                          \
    else if constexpr (KIND == KT_FIRST) { if constexpr (phase == 4) VMW(4) }                                        \
    else if constexpr (XL == 0) {                                                                                    \
        if constexpr (phase == 1) VMW(6) else if constexpr (phase == 2) VMW(8) else if constexpr (phase == 4) VMW(0) \
    } else {                                                                                                         \
        if constexpr (phase == 1) VMW(9) else if constexpr (phase == 2) VMW(11) else if constexpr (phase == 4) VMW(0) \
    }
    // ---------------- phase 1: A rows [0,64) x W rows [0,32)
    // KT_LAST: its phase-1 DMA writes W rows [0,32) of the OTHER buffer, which the lag waves read (RD_W(0), phase 4 of the
    // previous K-tile) one slot earlier and only retire with the LGKM0 that heads this slot.  The lead waves therefore issue
    // it in their second slot, behind the barrier that follows that LGKM0, so a barrier -- not global-load latency -- orders
    // the LDS write after the reads.  (Steady K-tiles write PA0 here, rows last read three phases earlier.)
    constexpr bool LATE_DMA1 = KIND == KT_LAST;
    if constexpr (LEAD) { RD_A(0) RD_W(0) if constexpr (!LATE_DMA1) { DMA(1) } } else if constexpr (!FIRST) { LGKM0; MMA(1, 0) }
    BAR;
    if constexpr (LEAD) { LGKM0; MMA(0, 0) if constexpr (LATE_DMA1) { DMA(1) } } else { RD_A(0) RD_W(0) DMA(1) }
    WAIT(1)
    BAR;
    // ---------------- phase 2: A rows [0,64) x W rows [32,64)
    if constexpr (LEAD) { RD_W(1) DMA(2) } else { LGKM0; MMA(0, 0) }
    BAR;
    if constexpr (LEAD) { LGKM0; MMA(0, 1) } else { RD_W(1) DMA(2) }
    WAIT(2)
    BAR;
    // ---------------- phase 3: A rows [64,128) x W rows [32,64)
    if constexpr (LEAD) { RD_A(1) DMA(3) } else { LGKM0; MMA(0, 1) }
    BAR;
    if constexpr (LEAD) { LGKM0; MMA(1, 1) } else { RD_A(1) DMA(3) }
    WAIT(3)
    BAR;
    // ---------------- phase 4: A rows [64,128) x W rows [0,32) (re-read: cheaper than 16 more live registers)
    if constexpr (LEAD) { RD_W(0) DMA(4) } else { LGKM0; MMA(1, 1) }
    BAR;
    if constexpr (LEAD) { LGKM0; MMA(1, 0) } else { RD_W(0) DMA(4) }
    WAIT(4)
    BAR;
#undef ISSUE
#undef ISSUE_NEXT_TILE
#undef DMA
#undef WAIT
}

// The K loop of one output tile for one wave: straight-line K-tiles, the first and the last peeled (nk >= 2).
template <bool LEAD, bool FOLD>
__device__ __forceinline__ void k_loop(const GemmParams& p, const LoopCtx& c, unsigned char* lds, f32x4_t (&acc)[8][4], float* fscr, int pb) {
    uint4 a[4][2], w[2][2];                        // [fragment][kk]
    k_tile<LEAD, KT_FIRST, FOLD>(p, c, lds, acc, a, w, fscr, 0, pb);
    for (int kt = 1; kt + 1 < c.nk; ++kt) k_tile<LEAD, KT_STEADY, FOLD>(p, c, lds, acc, a, w, fscr, kt, pb);
    k_tile<LEAD, KT_LAST, FOLD>(p, c, lds, acc, a, w, fscr, c.nk - 1, pb);
    if constexpr (!LEAD) { LGKM0; MMA(1, 0) }
}
#undef DSR
#undef RD_A
#undef RD_W
#undef MMA
#undef LGKM0
#undef BAR
#undef VMW


// FOLD: LayerNorm folded into this GEMM (gemm_common.h): A is the raw bf16 residual row, the epilogue normalises.
template <int EPI, typename OutT, bool FOLD = false>
__global__ __launch_bounds__(THREADS) void gemm256_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * BUF_BYTES + (FOLD ? 8 * 2048 : 0)];   // + 2 KiB per wave for the fold

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, g = lane >> 4;

    // PERSISTENT and STREAMED: the grid holds at most one workgroup per CU; workgroup w walks tiles w, w + grid, ... of the
    // same XCD / L2-aware order, and the K-tiles of consecutive output tiles form ONE stream through the two LDS buffers: the
    // last K-tile of a tile DMAs the first K-tile of the next one (k_tile kinds above).  The epilogue stages through the
    // buffer of the tile's last K-tile and starts with no load of its wave in flight; its stores are waited for three phases
    // into the next tile's K loop (they used to be waited for at the tile boundary: removing the stores alone took the
    // head-split QKV GEMM from 128 to 104 us, i.e. the store drain was fully exposed).
    const int ntiles = (p.M / BM) * (p.N / BN);
    const int tiles_m = p.M / BM, tiles_n = p.N / BN;
    LoopCtx c;
    c.lda_b = (unsigned)p.lda * 2u; c.ldw_b = (unsigned)p.ldw * 2u;
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
    c.Ar = tile_rsrc(p.A);
    c.Wr = tile_rsrc(p.W);
    c.aorg = (unsigned)m0 * c.lda_b; c.worg = (unsigned)n0 * c.ldw_b;

    // prologue of the FIRST tile: its whole first K-tile
    issue_piece(PA0, c.Ar, c.Wr, c.aorg, c.worg, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, lds, wave);
    issue_piece(PW0, c.Ar, c.Wr, c.aorg, c.worg, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, lds, wave);
    issue_piece(PW1, c.Ar, c.Wr, c.aorg, c.worg, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, lds, wave);
    issue_piece(PA1, c.Ar, c.Wr, c.aorg, c.worg, c.lda_b, c.ldw_b, c.laneoffA, c.laneoffW, 0, lds, wave);
    float* fscr = reinterpret_cast<float*>(lds + 2 * BUF_BYTES + (FOLD ? wave * 2048 : 0));
    c.lane = lane;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the whole first K-tile (KT_FIRST takes no waits for it)
    __builtin_amdgcn_s_barrier();

    int pb = 0;                                    // LDS buffer that holds K-tile 0 of the current tile
    for (;;) {
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        const int next = tile + (int)gridDim.x;
        const bool has_next = next < ntiles;
        int m1 = m0, n1 = n0;                          // no next tile: KT_LAST re-reads this tile's first K-tile (harmless)
        if (has_next) origin(next, m1, n1);
        c.aorg1 = (unsigned)m1 * c.lda_b; c.worg1 = (unsigned)n1 * c.ldw_b;

        c.mw = m0 + wm * 128; c.nw = n0 + wn * 64;
        if (wm == 0) k_loop<true, FOLD>(p, c, lds, acc, fscr, pb); else k_loop<false, FOLD>(p, c, lds, acc, fscr, pb);

        // every fragment read finished before the last barrier.  The buffer of the LAST K-tile is free for the epilogue's
        // staging (8 KiB per wave); the other one already holds the next tile's first K-tile.
#if defined(PM_FOLD_NOAPPLY)                     // timing ablation only (results are not normalised)
#elif defined(PM_FOLD_DIRECT)
        if constexpr (FOLD) ln_apply_direct<8>(p, c.mw, c.nw, acc, lane);
#else
        if constexpr (FOLD) ln_apply<8>(fscr, acc, lane);          // the coefficient DMA was retired by the last K-tile's vmcnt(0)
#endif
        const float4 no_pre[1] = {};
        unsigned char* eraw = lds + ((pb + c.nk - 1) & 1) * BUF_BYTES + wave * EPI_WAVE_BYTES;
        const int mw = m0 + wm * 128, nw = n0 + wn * 64;
        if constexpr (EPI == EPI_STD && sizeof(OutT) == 4) {
            if (p.residual) wave_epilogue<EPI, OutT, 8, 1, true, 1>(p, acc, eraw, mw, nw, lane, no_pre);
            else wave_epilogue<EPI, OutT, 8, 1, true, 0>(p, acc, eraw, mw, nw, lane, no_pre);
        } else if constexpr (EPI == EPI_STD) {
            // bf16 hi/lo residual stream: a producer never has a LayerNorm folded in (launch256 checks), so the folded kernel does
            // not carry that epilogue
            if (!FOLD && p.out_lo) wave_epilogue<EPI, OutT, 8, 1, true, 2>(p, acc, eraw, mw, nw, lane, no_pre);
            else wave_epilogue<EPI, OutT, 8, 1, true, 0>(p, acc, eraw, mw, nw, lane, no_pre);
        } else {
            wave_epilogue<EPI, OutT, 8, 1, true, 0>(p, acc, eraw, mw, nw, lane, no_pre);
        }
        if (!has_next) break;
        __builtin_amdgcn_s_barrier();                  // every wave is done staging through the free buffer (the next K-tile 1 lands there)
        tile = next; m0 = m1; n0 = n1;
        c.aorg = c.aorg1; c.worg = c.worg1;
        pb = (pb + c.nk) & 1;
    }
}

// Process-wide tuning knobs of this kernel (development builds only, common.h pm_dev_knob: tools/chunk_sweep.sh, tools/env_sweep.sh),
// read exactly ONCE, under std::call_once: the first launches of a process come from several lane threads at the same time.
struct Knobs256 {
    int chunk = 6;          // PMHIP_CHUNK256: width of the L2-aware tile walk
    int persist = 256;      // PMHIP_PERSIST256: workgroups of the persistent grid (0 = one per tile)
    int res_kmin = 1024;    // PMHIP_G256_RES_KMIN: shortest K at which a residual GEMM takes this kernel
};
const Knobs256& knobs256() {
    static Knobs256 k;
    static std::once_flag once;
    std::call_once(once, [] {
        k.chunk = pm_dev_knob("PMHIP_CHUNK256", k.chunk);
        k.persist = pm_dev_knob("PMHIP_PERSIST256", k.persist);
        k.res_kmin = pm_dev_knob("PMHIP_G256_RES_KMIN", k.res_kmin);
    });
    return k;
}

template <int EPI, typename OutT>
int launch256(const GemmParams& p0, hipStream_t s) {
    GemmParams p = p0;
    p.chunk = knobs256().chunk;
    const int tiles = (p.M / BM) * (p.N / BN);
    const int persist = knobs256().persist;
    const int grid = (persist > 0 && tiles > persist) ? persist : tiles;
    if (p.ln_coef && p.out_lo) { pm_set_error("gemm256: a hi/lo residual producer cannot have a LayerNorm folded in"); return PMHIP_EINVAL; }
    PmTimer tm(gemm_family(p, EPI), s);
    if (p.ln_coef) hipLaunchKernelGGL((gemm256_kernel<EPI, OutT, true>), dim3(grid), dim3(THREADS), 0, s, p);
    else hipLaunchKernelGGL((gemm256_kernel<EPI, OutT, false>), dim3(grid), dim3(THREADS), 0, s, p);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

}  // namespace

int pm_gemm256_supported(const GemmParams& p, int dtype, int epi, int out_dtype) {
    if (dtype != PMHIP_BF16) return 0;
    if (p.M % BM || p.N % BN || p.K % KSTEP || p.K < 2 * KSTEP) return 0;   // the streamed K loop peels a first and a last K-tile
    if ((unsigned long long)p.M * p.lda * 2 >= (1ull << 31) || (unsigned long long)p.N * p.ldw * 2 >= (1ull << 31)) return 0;   // 32-bit buffer offsets   // the streamed K loop peels a first and a last K-tile
    if ((p.M / BM) * (p.N / BN) < 96) return 0;             // too few tiles: the small kernel has 4x the workgroups
    if (epi == EPI_STD && p.residual) {
        // residual GEMMs: the 128x128 kernel prefetches the residual tile and overlaps two workgroups per CU, which wins
        // while the GEMM is HBM-bound (small K); with a long K loop the faster main loop of this kernel wins
        if (p.K < knobs256().res_kmin) return 0;
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
