// Fused softmax(Q K^T) V in bf16 for dim_head = 64 (gfx950), fourth generation (the third-generation notes are kept below
// because the fourth builds on them).
// Replaces the reference's materialised-score attention (modules/attention.py:51-58: q@k^T -> softmax -> @v, a
// (B*H, N, N) fp32 tensor per layer) and its xformers alternative (:100).  Same data layout and work split as the
// f32 kernel in attention.hip (which stays the fp32-verify path): Q [B,H,Nq,64] pre-scaled, K [B,H,Np,64],
// V^T [B,H,64,Np]; one workgroup = 64*QF queries of one (batch, head), 4 waves x 16*QF queries (QF = 4, or 2 / 1 when the
// launch would otherwise leave CUs idle: bit-identical forms); K / V^T tiles of 64 keys by DMA into a 4-stage LDS ring with
// counted vmcnt waits, one bare s_barrier per tile; swapped QK^T (a lane owns 8 consecutive keys of ONE query per
// 32-key half-tile, so P feeds the P.V product straight from the S^T accumulators).
//
// What changed against the second generation (round 2: 0.40 MFMA busy, 4.5 VALU per MFMA):
//  * the row sums l = sum_k P[k, q] are computed by the MATRIX pipe: one extra MFMA per 16-query tile with an all-ones
//    row operand accumulates sum_k bf16(P) into an f32 accumulator whose 16 rows are all l (32 adds per half-tile ->
//    4 MFMAs; no cross-lane reduction at the end either).  l is therefore the sum of the ROUNDED probabilities, the
//    same values that multiply V.
//  * S^T accumulators start from -m by naming the running-max quad as the MFMA's C operand (D != C): no copies.
//  * growth of the running max is detected from ONE in-lane maximum over the lane's 32 scores (16 v_max3 instead of 20
//    + compares); the per-tile maxima are only computed inside the rare rescale branch.
//  * per half-tile the instruction stream is two blocks that each carry matrix work AND vector work:
//      A: 16 MFMAs of S^T(h+1)          with the 32 exponentials + 16 bf16 packs of S^T(h)
//      B: 20 MFMAs of P.V(h) + l(h)     with the 16 v_max3 of S^T(h+1)
//    V^T fragments of h are requested before block A, K fragments of h+2 before block B, so no LDS latency is exposed.
//
// Fourth generation (round 5): the steady loop has NO running-max bookkeeping at all.
//  * The reference max of a query is fixed after the first 32-key half-tile; every later probability is 2^(s - m_ref),
//    whatever its size.  bf16 P and the f32 accumulators have the exponent range of f32, so nothing is lost until a
//    probability overflows -- which the epilogue detects POST HOC (l not below 2^64, NaN included) and answers by running the
//    whole workgroup again through the exact path (running max raised at every half-tile: the rare-path code that ragged and
//    short contexts use anyway).  Round 4 counted the old growth branch on real data: 0 executions in 134 M steps, while
//    its detector was 16 of the 69 vector instructions of a half-tile.
//  * S^T is single-buffered and the half-tile is walked QUERY-TILE-major: group g issues the 4 QK^T MFMAs of S^T(h+1, g)
//    (overwriting S^T(h, g), whose exponentials were issued one group earlier), the 4 P.V MFMAs + the row-sum MFMA of
//    (h, g), and the 8 exponentials + 4 packs of (h, g+1).  Every group is 9 MFMAs beside 12 vector instructions: the
//    vector work is spread evenly under ALL matrix instructions (third generation: 3 per MFMA in block A, none in B).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#ifndef ABL
#define ABL 0                 // ablation bit mask (tools/hwtests/attn_abl.hip); 0 in the library
#endif

// workgroups whose fast path overflowed and that were run again through the exact path (pmhip_attention_fallbacks)
__device__ unsigned long long g_attn_fallbacks;
#ifndef PM_ATTN_NO_ABI           // tools/hwtests/attn_ab.hip compiles this file several times in one program
extern "C" int pmhip_attention_fallbacks(unsigned long long* count, int reset) {
    PM_REQUIRE(count != nullptr, "pmhip_attention_fallbacks: count is NULL");
    PM_HIP(hipDeviceSynchronize());
    PM_HIP(hipMemcpyFromSymbol(count, HIP_SYMBOL(g_attn_fallbacks), sizeof(unsigned long long)));
    if (reset) {
        const unsigned long long z = 0;
        PM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_fallbacks), &z, sizeof(z)));
    }
    return PMHIP_OK;
}
#endif

#ifdef PM_ATTN_COUNT
// DEBUG BUILD ONLY (tools/attn_rescale_count.sh): how often the steady loop leaves its fast path on real data.
// [1] fast half-tile steps (per wave), [2] exact steps
__device__ unsigned long long g_attn_counters[4];
extern "C" int pmhip_debug_attention_counters(unsigned long long* out4, int reset) {
    if (out4 && hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_attn_counters), 32) != hipSuccess) return 1;
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_counters), z, 32) != hipSuccess) return 1; }
    return 0;
}
#endif

namespace {

constexpr int KT = 64;        // keys per tile
constexpr int DH = 64;
constexpr int THREADS = 256;
constexpr int TILE_BYTES = KT * 128;
constexpr int STAGE_BYTES = 2 * TILE_BYTES;              // K tile + V^T tile
#ifndef PM_ATTN_KSWZ
#define PM_ATTN_KSWZ 1        // 0: the K tile swizzle of rounds 3-4 (2-way bank conflicts on the K fragment reads); A/B only
#endif
#ifndef PM_ATTN_RING
#define PM_ATTN_RING 4
#endif
constexpr int RING = PM_ATTN_RING;       // ring stages: a tile is requested RING - 2 tiles before it is entered (round 5: 4;
                                         // with 3 the DMA had ONE tile period, about 1 us, to come back from HBM)
constexpr int AHEAD = RING - 2;

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4_t mma(const v4u_t& rows, const v4u_t& cols, const f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, rows), __builtin_bit_cast(bf16x8_t, cols), c, 0, 0, 0);
}

#if ABL & 2
#define DSRX(dst, addr, off) asm volatile("; no read %0 %1 %2" : "=v"(dst) : "v"(addr), "n"(off))
#else
#define DSRX(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#endif
// one counted wait that names four fragments as in/out operands: every MFMA that consumes one is ordered behind it
#define LGKM4(n, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define LGKM2(n, a, b) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b))

// QF = 16-query tiles per wave: 4 (256 queries per workgroup) wherever that fills the chip; 2 and 1 (128 / 64 queries per
// workgroup) for small batches, where a grid of 256-query workgroups leaves most CUs idle (B = 1, H = 8, N = 1024: 32
// workgroups of 41 us each).  The arithmetic of a 16-query tile does not depend on QF or on its neighbours in the workgroup
// (same MFMA chains, same half-tile order, the fallback below decided per tile), so an image's result does not depend on the
// batch it runs in.
// (Measured and not adopted, round 5: ONE workgroup of 8 waves / 512 queries per CU sharing the ring -- half the DMA pieces per
// wave and per CU -- needs 13 % MORE cycles, 2.45e6 against 2.16e6 per launch, MFMA busy 0.49 against 0.55: the 8-wave barrier
// per tile costs more than the DMA saves; two independent 4-wave workgroups cover each other's barrier waits.  The patch is
// tools/ab_variants/attn_wv8.patch, the numbers profiles/r05_a_attention_gen4_ab.txt (5).)
template <bool EXP2, int QF>
__global__ __launch_bounds__(THREADS, 2) void attention_bf16_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kp,
                                                                    const bf16_t* __restrict__ Vt, bf16_t* __restrict__ out,
                                                                    int ldo, int heads, int Nq, int Nkv, int Nkv_pad, int nqb) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[RING * STAGE_BYTES];   // K / V^T ring

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // 1-D grid.  Workgroup L runs on XCD L % 8 (private 4 MiB L2): give all query blocks of one (batch, head) the
    // same L % 8 so its K / V^T (256 KiB) are fetched from HBM once and re-read from that XCD's L2.
    int bh, qblk;
    {
        const int L = blockIdx.x, total_bh = gridDim.x / nqb;
        if ((total_bh & 7) == 0) {
            const int slot = L >> 3;
            qblk = slot % nqb;
            bh = (slot / nqb) * 8 + (L & 7);
        } else {
            qblk = L % nqb;
            bh = L / nqb;
        }
    }
    const int b = bh / heads, h = bh % heads;
    const int q0 = qblk * (4 * QF * 16) + wave * (QF * 16);

    const bf16_t* Qbh = Q + (size_t)bh * Nq * DH;
    const unsigned char* Kbh = reinterpret_cast<const unsigned char*>(Kp + (size_t)bh * Nkv_pad * DH);
    const unsigned char* Vbh = reinterpret_cast<const unsigned char*>(Vt + (size_t)bh * DH * Nkv_pad);
    const unsigned v_row_bytes = (unsigned)Nkv_pad * 2u;
    // DMA descriptors / lane offsets, and the per-lane parts of the fragment addresses (ds_read_b128 with immediate offsets)
    //   K row of S^T tile kf = 2 pc + kk, row i = l15:  32 pc + 8 (l15 >> 2) + 4 kk + (l15 & 3);  slot (4 c + g) ^ (row & 7)
    //     = stage + [8 (l15 >> 2) + (l15 & 3)] * 128 + (g ^ (l15 & 3)) * 16  +  pc * 4096 + kk * 512 + (c ^ kk) * 64
    //     (round 5: slot additionally ^ 4 where bit 3 of the row is set, i.e. "+ (c ^ kk ^ ((l15 >> 2) & 1)) * 64")
    //   V^T row 16 df + l15, slot (4 pc + g) ^ (l15 & 7)
    //     = stage + 8192 + l15 * 128 + ((4 pc + g) ^ (l15 & 7)) * 16  +  df * 2048
    const rsrc_t Kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Kbh), 0, 0x7fffffff, 0x00020000);
    const rsrc_t Vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Vbh), 0, 0x7fffffff, 0x00020000);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned lslot = (unsigned)(((lane & 7) ^ ((lane >> 3) & 7)) << 4);
    const unsigned kvoff = (unsigned)(lane >> 3) * 128u + lslot;
    // K tile, bank conflicts (round 5): a ds_read_b128 is served 16 lanes at a time ({0-3, 12-15, 20-27}, ...), and the S^T row
    // order puts l15 = 0..3 and 12..15 on rows 0..3 and 24..27 -- same row & 7, same slot: a 2-way conflict on every K fragment
    // read (SQ_LDS_BANK_CONFLICT a third of SQ_LDS_IDX_ACTIVE).  Bit 3 of the row now flips bit 2 of the slot as well: the odd
    // 8-row chunks are DMA'd with the flipped source slot, and the read side flips it for the lanes with (l15 >> 2) odd.
    const unsigned kvoff1 = PM_ATTN_KSWZ ? kvoff ^ 64u : kvoff;
    const unsigned vvoff = (unsigned)(lane >> 3) * v_row_bytes + lslot;
    const unsigned kfrag_lane = lds_base + (unsigned)(8 * (l15 >> 2) + (l15 & 3)) * 128u + (unsigned)((g ^ (l15 & 3)) << 4) +
                                (PM_ATTN_KSWZ ? (unsigned)(((l15 >> 2) & 1) << 6) : 0u);          // slots with c ^ kk = 0
    const unsigned kfrag_laneB = kfrag_lane ^ 64u;                                                 // slots with c ^ kk = 1
    const unsigned vfrag_lane0 = lds_base + 8192u + (unsigned)l15 * 128u + (unsigned)(((0 + g) ^ (l15 & 7)) << 4);
    const unsigned vfrag_lane1 = lds_base + 8192u + (unsigned)l15 * 128u + (unsigned)(((4 + g) ^ (l15 & 7)) << 4);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int q0u = qblk * (4 * QF * 16) + wave_u * (QF * 16);       // q0, provably wave-uniform

    // one K tile + one V^T tile by DMA, 1 KiB per wave-instruction; the bank swizzle (slot ^ row) is applied to the SOURCE
    // address (kvoff / vvoff) and again on the read side
    auto stage_tiles = [&](int t) {
        unsigned char* stage = lds + (t % RING) * STAGE_BYTES;
        const unsigned kv0 = (unsigned)t * KT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned chunk = (unsigned)wave_u * 2 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Kr, (__attribute__((address_space(3))) void*)(stage + chunk * 1024), 16, i ? kvoff1 : kvoff,
                                                     (kv0 + chunk * 8) * 128u, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Vr, (__attribute__((address_space(3))) void*)(stage + KT * 128 + chunk * 1024), 16, vvoff,
                                                     chunk * 8 * v_row_bytes + kv0 * 2u, 0, 0);
        }
    };

    // Q fragments stay in registers for the whole kernel (column operand of S^T)
    v4u_t qreg[QF][2];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        int q = q0 + qf * 16 + l15;
        q = q < Nq ? q : Nq - 1;
        const unsigned char* qrow = reinterpret_cast<const unsigned char*>(Qbh + (size_t)q * DH);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if constexpr (ABL & 64) { (void)qrow; qreg[qf][c] = v4u_t{0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}; }
            else qreg[qf][c] = *reinterpret_cast<const v4u_t*>(qrow + (c * 4 + g) * 16);
        }
    }

    f32x4_t o[4][QF];
    f32x4_t lacc[QF];                    // every element = l of the query column (sum of bf16 P, by MFMA with a ones operand)
    f32x4_t negm[QF];                    // -m (reference max of the query column) x4: the C operand of the S^T MFMAs
    v4u_t ones = v4u_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    asm volatile("" : "+v"(ones));       // keep it in registers (not re-materialised in front of every use)

    const int ntiles = (Nkv + KT - 1) / KT;
    const int nhalves = (Nkv + 31) / 32;                     // 32-key half-tiles that contain at least one valid key

    auto k_issue = [&](v4u_t (&kf)[2][2], int hh) {
        const unsigned so = (unsigned)((hh >> 1) % RING) * STAGE_BYTES + (unsigned)(hh & 1) * 4096u;
        const unsigned ka = kfrag_lane + so, kb = kfrag_laneB + so;
        DSRX(kf[0][0], ka, 0 * 512); DSRX(kf[0][1], kb, 0 * 512);
        DSRX(kf[1][0], kb, 1 * 512); DSRX(kf[1][1], ka, 1 * 512);
    };
    auto v_issue = [&](v4u_t (&vf)[4], int hh) {
        const unsigned va = ((hh & 1) ? vfrag_lane1 : vfrag_lane0) + (unsigned)((hh >> 1) % RING) * STAGE_BYTES;
        DSRX(vf[0], va, 0 * 2048); DSRX(vf[1], va, 1 * 2048); DSRX(vf[2], va, 2 * 2048); DSRX(vf[3], va, 3 * 2048);
    };

    // S^T of one half-tile, starting from -m
    auto qk = [&](f32x4_t (&sd)[2][QF], v4u_t (&kf)[2][2]) {
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) sd[0][qf] = mma(kf[0][0], qreg[qf][0], negm[qf]);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) sd[0][qf] = mma(kf[0][1], qreg[qf][1], sd[0][qf]);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) sd[1][qf] = mma(kf[1][0], qreg[qf][0], negm[qf]);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) sd[1][qf] = mma(kf[1][1], qreg[qf][1], sd[1][qf]);
    };

    // rare, wave-uniform: mask a ragged last tile, raise the running max, rescale everything at the old max exactly once
    auto rescale = [&](auto ragged_c, auto first_c, f32x4_t (&sc)[2][QF], int hh) {
        constexpr bool first = decltype(first_c)::value;
        const int kv0 = (hh >> 1) * KT, pc = hh & 1;
        if (decltype(ragged_c)::value && kv0 + KT > Nkv) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kv0 + 32 * pc + 8 * g + 4 * kk + r;
                    if (key >= Nkv) {
#pragma unroll
                        for (int qf = 0; qf < QF; ++qf) sc[kk][qf][r] = -INFINITY;
                    }
                }
        }
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            float m = vmax3(sc[0][qf][0], sc[0][qf][1], sc[0][qf][2]);
            m = vmax3(m, sc[0][qf][3], sc[1][qf][0]);
            m = vmax3(m, sc[1][qf][1], sc[1][qf][2]);
            m = vmax2(m, sc[1][qf][3]);                      // this lane's 8 keys, relative to mb
            const float mold = first ? -INFINITY : -negm[qf][0];
            const float mb = first ? 0.f : mold;             // what the accumulators started from
            const float mnew = vmax3(mold, group4_max(m) + mb, -1e30f);   // column max over the 4 lane groups
            const float delta = mb - mnew;                   // scores hold s - mb: move them to s - mnew
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[kk][qf][r] += delta;
            // -m moves by the same delta, component by component and in place (a quad rebuilt from one scalar costs the
            // COMMON path a copy of all of negm at the join)
            if constexpr (first) {
                negm[qf][0] = delta; negm[qf][1] = delta; negm[qf][2] = delta; negm[qf][3] = delta;
            } else {
                negm[qf][0] += delta; negm[qf][1] += delta; negm[qf][2] += delta; negm[qf][3] += delta;
            }
            if constexpr (!first) {                          // (the first half-tile finds l = O = 0: nothing to move)
                const float alpha = EXP2 ? __builtin_amdgcn_exp2f(mold - mnew) : expf(mold - mnew);
                lacc[qf][0] *= alpha; lacc[qf][1] *= alpha; lacc[qf][2] *= alpha; lacc[qf][3] *= alpha;
#pragma unroll
                for (int df = 0; df < 4; ++df) {
                    o[df][qf][0] *= alpha; o[df][qf][1] *= alpha; o[df][qf][2] *= alpha; o[df][qf][3] *= alpha;
                }
            }
        }
    };

    // entering tile tn (called while the previous tile's second half is still to be consumed): its DMA has landed
    // and is published by the barrier; the barrier also proves every wave is done with tile tn-2, whose stage the
    // DMA of tile tn+AHEAD now reuses (RING stages: tn-2 and tn+AHEAD share one)
    auto enter_tile = [&](auto ragged_c, int tn) {
        if (!(ABL & 8) || tn == 0) {
            // this wave's pieces of tile tn have landed: everything but the pieces of the younger tiles in flight behind them
            // (4 instructions per tile; vmcnt retires in issue order)
            // The barrier is the bare instruction: __syncthreads() carries a fence, for which hipcc drains vmcnt to 0 -- that
            // would wait for the younger tile as well.  Nothing else needs the fence here: the fast path reads LDS with
            // inline-asm ds_read only, and the exact path's V^T patch below is followed by a full __syncthreads().
            if (AHEAD == 2 && tn + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (tn + AHEAD < ntiles && !(ABL & 4)) stage_tiles(tn + AHEAD);
        if (decltype(ragged_c)::value && tn * KT + KT > Nkv) {   // ragged last tile: zero the V^T columns of keys >= Nkv
            unsigned char* Vl = lds + (tn % RING) * STAGE_BYTES + TILE_BYTES;
            for (int idx = tid; idx < KT * 8; idx += THREADS) {
                const int row = idx / 8, ls = idx % 8;
                uint4* p = reinterpret_cast<uint4*>(Vl + row * 128 + ((ls ^ (row & 7)) << 4));
                uint4 v = *p;
                const int n = Nkv - (tn * KT + ls * 8);      // valid keys in this 8-key chunk (may be <= 0)
                v.x = n <= 0 ? 0u : (n == 1 ? (v.x & 0xffffu) : v.x);
                v.y = n <= 2 ? 0u : (n == 3 ? (v.y & 0xffffu) : v.y);
                v.z = n <= 4 ? 0u : (n == 5 ? (v.z & 0xffffu) : v.z);
                v.w = n <= 6 ? 0u : (n == 7 ? (v.w & 0xffffu) : v.w);
                *p = v;
            }
            __syncthreads();
        }
    };

    f32x4_t sA[2][QF];                   // S^T of ONE half-tile (single-buffered: group g of a step overwrites the tile it has consumed)
    v4u_t pf[QF];
    v4u_t kf[2][2], vf[4];

    // exponentials of the 16-query tile qf of S^T(h) and their packing into the P^T operand; S^T itself is left as it is
    auto exp_pack1 = [&](f32x4_t (&sc)[2][QF], int qf) {
        float e[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                e[kk][r] = (ABL & 1) ? sc[kk][qf][r] * 1.0001f : (EXP2 ? __builtin_amdgcn_exp2f(sc[kk][qf][r]) : expf(sc[kk][qf][r]));   // sc = s - m
        pf[qf] = v4u_t{pack_bf16x2(e[0][0], e[0][1]), pack_bf16x2(e[0][2], e[0][3]), pack_bf16x2(e[1][0], e[1][1]), pack_bf16x2(e[1][2], e[1][3])};
    };
    // P.V and the row sums of one half-tile (exact path)
    auto pv_all = [&]() {
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) o[df][qf] = mma(vf[df], pf[qf], o[df][qf]);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) lacc[qf] = mma(ones, pf[qf], lacc[qf]);
    };

    // One half-tile h of the FAST path, query-tile-major.  On entry sA = S^T(h) - m_ref (tiles 1..3 untouched, tile 0 already
    // turned into pf[0]), the K fragments of h+1 and then the V^T fragments of h are in flight (in that order).  Group g:
    //     matrix:  S^T(h+1, g) = K(h+1) Q_g - m_ref   (4 MFMAs, overwrites S^T(h, g))
    //              O^T(., g) += V^T(h) P(h, g),  l_g += 1 P(h, g)                    (5 MFMAs)
    //     vector:  P(h, g+1) = bf16(exp2(S^T(h, g+1)))  -- for g = 3: P(h+1, 0), from the S^T(h+1, 0) of this step's group 0
    // The K fragments of h+2 are requested behind the last QK^T MFMA, the V^T fragments of h+1 behind the last P.V MFMA.
    //   OPENS: h+2 is the first half of a new tile
    auto grp_mma = [&](f32x4_t (&sc)[2][QF], int g) {
        sc[0][g] = mma(kf[0][0], qreg[g][0], negm[g]);
        sc[1][g] = mma(kf[1][0], qreg[g][0], negm[g]);
        sc[0][g] = mma(kf[0][1], qreg[g][1], sc[0][g]);
        sc[1][g] = mma(kf[1][1], qreg[g][1], sc[1][g]);
    };
    auto grp_pv = [&](int g) {
#pragma unroll
        for (int df = 0; df < 4; ++df) o[df][g] = mma(vf[df], pf[g], o[df][g]);
        lacc[g] = mma(ones, pf[g], lacc[g]);
    };
    auto grp_sched = [&](bool last) {                        // 9 MFMAs, 8 transcendentals, 4 packs: M T T M P  x4, M  (last group: M T T P)
        if constexpr (EXP2 && !(ABL & 1)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                if (!last) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
    };
    auto step = [&](auto opens_c, f32x4_t (&sc)[2][QF], int hh) {
        constexpr bool OPENS = decltype(opens_c)::value;
        const unsigned va = (((hh + 1) & 1) ? vfrag_lane1 : vfrag_lane0) + (unsigned)(((hh + 1) >> 1) % RING) * STAGE_BYTES;
        const unsigned kso = (unsigned)(((hh + 2) >> 1) % RING) * STAGE_BYTES + (unsigned)(hh & 1) * 4096u;
        const unsigned ka = kfrag_lane + kso, kb = kfrag_laneB + kso;
#pragma unroll
        for (int g = 0; g < QF; ++g) {
            const bool last = g == QF - 1;
            if (OPENS && last) enter_tile(std::false_type{}, (hh + 2) >> 1);
            if (g == 0) LGKM4(4, kf[0][0], kf[0][1], kf[1][0], kf[1][1]);          // K(h+1) landed; the V^T(h) reads are younger
            grp_mma(sc, g);
            if (last) {                                       // the fragment registers are handed over to the next half-tile as they die
                __builtin_amdgcn_sched_barrier(0);
                DSRX(kf[0][0], ka, 0 * 512); DSRX(kf[0][1], kb, 0 * 512);
                DSRX(kf[1][0], kb, 1 * 512); DSRX(kf[1][1], ka, 1 * 512);
            }
            if (g == 0) {
                if (last) LGKM4(4, vf[0], vf[1], vf[2], vf[3]);                     // (QF = 1: the K(h+2) reads just issued stay in flight)
                else LGKM4(0, vf[0], vf[1], vf[2], vf[3]);
            }
            grp_pv(g);
            exp_pack1(sc, (g + 1) % QF);                      // last group: P(h+1, 0), from the S^T(h+1, 0) of this step's group 0
            grp_sched(last);
            asm volatile("" : "+v"(pf[(g + 1) % QF]));            // the packs are complete here (not sunk to their first use)
            __builtin_amdgcn_sched_barrier(0);
            if (last) { DSRX(vf[0], va, 0 * 2048); DSRX(vf[1], va, 1 * 2048); DSRX(vf[2], va, 2 * 2048); DSRX(vf[3], va, 3 * 2048); }
        }
    };

    // The exact path, one half-tile with every condition at run time, full waits and the running max raised at once: first and
    // last tiles, ragged tiles, short contexts, and a workgroup the fast path gave up on.  sA: S^T(h) -> P(h) -> S^T(h+1).
    auto slow_step = [&](int hh) {
        const bool next = hh + 1 < nhalves, next2 = hh + 2 < nhalves;
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) exp_pack1(sA, qf);
        v_issue(vf, hh);
        if (next) {
            LGKM4(4, kf[0][0], kf[0][1], kf[1][0], kf[1][1]);
            qk(sA, kf);
        }
        if (next2 && !(hh & 1)) enter_tile(std::true_type{}, (hh + 2) >> 1);
        if (next2) k_issue(kf, hh + 2);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]), "+v"(kf[0][0]), "+v"(kf[0][1]),
                     "+v"(kf[1][0]), "+v"(kf[1][1]));
        pv_all();
        if (next) rescale(std::true_type{}, std::false_type{}, sA, hh + 1);
    };

    constexpr std::true_type Y{};
    constexpr std::false_type N{};
    __shared__ int redo_vote[4];

    const int nh_full = 2 * (Nkv / KT);                      // half-tiles that lie in full tiles
    // A context without a ragged tile (self-attention: every stage-2 / ViT launch of the decode loop) runs ALL its half-tiles,
    // the last two included, through the fast step.  Past the end `step` still computes S^T(h+1) and prefetches K(h+2) / V^T(h+1):
    // they address ring stages that still hold already-consumed tiles (no DMA is issued past the last tile), and nothing they
    // produce is consumed.
    const bool all_steady = (Nkv % KT) == 0 && ntiles >= 3;
    const int steady_end = all_steady ? nhalves - 1 : min(nhalves - 3, nh_full - 2);   // one bound: the loop's shape is unchanged
    bool exact = steady_end <= 0;                            // workgroup-uniform: no fast step at all, or second attempt
    unsigned redo_mask = ~0u;                                // tiles the exact attempt stores (all, unless it is a second attempt)

    // O = O^T / l, head-major inside the output row, for the 16-query tiles in `mask`.  The wave's output rows go through the
    // (idle) K / V^T ring, so that every global store instruction writes 8 whole 128-byte rows (non-temporal)
    auto finalize = [&](unsigned mask) {
        constexpr int RS = 144;                              // staged row: 64 bf16 + pad, 16-B aligned, conflict-free
        unsigned char* obuf = lds + wave * (QF * 16 * RS);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            const float inv = 1.0f / lacc[qf][0];
#pragma unroll
            for (int df = 0; df < 4; ++df)
                *reinterpret_cast<uint2*>(obuf + (qf * 16 + l15) * RS + (df * 16 + g * 4) * 2) =
                    make_uint2(pack_bf16x2(o[df][qf][0] * inv, o[df][qf][1] * inv), pack_bf16x2(o[df][qf][2] * inv, o[df][qf][3] * inv));
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // wave-uniform 64-bit base + one 32-bit lane offset: nothing lane-dependent and 64 bits wide for the compiler to hoist to
        // the kernel entry and spill around the loop
        unsigned char* rowbase = reinterpret_cast<unsigned char*>(out + ((size_t)b * Nq + q0u) * ldo + h * DH);
        unsigned lane_off = ((unsigned)(lane >> 3) * (unsigned)ldo + (unsigned)(lane & 7) * 8u) * 2u;
        unsigned rd_off = (unsigned)(lane >> 3) * RS + (unsigned)(lane & 7) * 16u;
        // (this lambda sits inside the attempt loop: without the opaque moves the per-row addresses are loop-invariant, get hoisted
        // to the kernel entry -- 30 registers -- and are spilled around the K loop)
        asm volatile("" : "+v"(lane_off), "+v"(rd_off));
#pragma unroll
        for (int it = 0; it < QF * 2; ++it) {                // 8 rows x 128 B per store instruction
            const int q = q0 + it * 8 + (lane >> 3);
            if (q < Nq && ((mask >> (it >> 1)) & 1u) && (!(ABL & 32) || q < 0)) {
                const v4u_t v = *reinterpret_cast<const v4u_t*>(obuf + rd_off + it * 8 * RS);
                v4u_t* dst = reinterpret_cast<v4u_t*>(rowbase + (lane_off + (unsigned)(it * 8) * (unsigned)ldo * 2u));
                if constexpr (QF == 4) __builtin_nontemporal_store(v, dst);   // large launches stream their output past the caches;
                else *dst = v;                                                 // a small one is read at once by the next kernel of the chain
            }
        }
    };

    for (;;) {
#pragma unroll
        for (int j = 0; j < QF; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            lacc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            negm[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        stage_tiles(0);
        if (AHEAD == 2 && ntiles > 1) stage_tiles(1);
        enter_tile(Y, 0);
        k_issue(kf, 0);
        LGKM4(0, kf[0][0], kf[0][1], kf[1][0], kf[1][1]);
        qk(sA, kf);
        rescale(Y, Y, sA, 0);                                // m_ref = the maximum over the first 32 keys
        if (nhalves > 1) k_issue(kf, 1);

        int hs = 0;
        if (!exact) {
            v_issue(vf, 0);
            exp_pack1(sA, 0);
            for (; hs < steady_end; hs += 2) {               // fast path
                step(Y, sA, hs);
                step(N, sA, hs + 1);
            }
            LGKM4(0, kf[0][0], kf[0][1], kf[1][0], kf[1][1]);    // the reads of the last step are not left outstanding
            LGKM4(0, vf[0], vf[1], vf[2], vf[3]);
            // hand-over to the exact steps: sA = S^T(hs) with tiles 1..3 untouched (tile 0 is exponentiated again, same bits), K(hs+1) in kf
        }
#ifdef PM_ATTN_COUNT
        if (lane == 0) { atomicAdd(&g_attn_counters[1], (unsigned long long)hs); atomicAdd(&g_attn_counters[2], (unsigned long long)(nhalves - hs)); }
#endif
        for (; hs < nhalves; ++hs) slow_step(hs);

        // every wave is done reading the ring; and the vote: did a probability of the fast path leave the f32 range?
        unsigned badmask = 0;                                // wave-uniform: bit qf = tile qf of this wave overflowed
        if (!exact && !(ABL & 6)) {                          // (ablations that compute garbage do not vote)
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) badmask |= __any(!(lacc[qf][0] < 1.8446744e19f)) ? (1u << qf) : 0u;     // 2^64; NaN fails too
            if (lane == 0) redo_vote[wave] = (int)badmask;
        }
        __syncthreads();
        if (exact) break;
        const int4 votes = *reinterpret_cast<const int4*>(redo_vote);
        if (__builtin_expect(__builtin_amdgcn_readfirstlane(votes.x | votes.y | votes.z | votes.w) == 0, 1)) break;
        // Rare: some 16-query tile of this workgroup overflowed.  The good tiles are stored now, from the fast path (a tile's
        // result never depends on its neighbours); the workgroup then runs again through the exact path and stores the others.
        if (tid == 0) atomicAdd(&g_attn_fallbacks, 1ull);
        finalize(~badmask);
        redo_mask = badmask;
        exact = true;                                        // (the exact attempt does not vote: no write races the read above)
        __syncthreads();                                     // the staging area is the ring: every wave has read its rows back
    }
    finalize(redo_mask);                                     // the common case: every tile, straight from the fast path
}

#undef DSRX
#undef LGKM4
#undef LGKM2

template <int QF>
static void launch_qf(const void* Q, const void* K, const void* Vt, void* out, int ldo, int B, int heads, int Nq, int Nkv, int Nkv_pad,
                      int use_exp2, hipStream_t s) {
    const int nqb = ceil_div(Nq, 4 * QF * 16);
    dim3 grid(nqb * B * heads), block(THREADS);
    if (use_exp2)
        hipLaunchKernelGGL((attention_bf16_kernel<true, QF>), grid, block, 0, s, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)Vt,
                           (bf16_t*)out, ldo, heads, Nq, Nkv, Nkv_pad, nqb);
    else
        hipLaunchKernelGGL((attention_bf16_kernel<false, QF>), grid, block, 0, s, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)Vt,
                           (bf16_t*)out, ldo, heads, Nq, Nkv, Nkv_pad, nqb);
}

}  // namespace

// bf16 leg of pmhip_attention (attention.hip): arguments already validated there.  The largest workgroup that still gives the
// chip two workgroups per CU (256 CUs on this part: 512) is taken; the result of an image does not depend on the choice.
int pm_attention_bf16(const void* Q, const void* K, const void* Vt, void* out, int ldo, int B, int heads, int Nq, int Nkv,
                      int Nkv_pad, int use_exp2, hipStream_t s) {
#ifdef PM_ATTN_FORCE_QF
    constexpr int kFill = 0;
    const int force = PM_ATTN_FORCE_QF;
#else
    constexpr int kFill = 512;
    const int force = 0;
#endif
    const long long bh = (long long)B * heads;
    if (force == 4 || (!force && bh * ceil_div(Nq, 256) >= kFill)) launch_qf<4>(Q, K, Vt, out, ldo, B, heads, Nq, Nkv, Nkv_pad, use_exp2, s);
    else if (force == 2 || (!force && bh * ceil_div(Nq, 128) >= kFill)) launch_qf<2>(Q, K, Vt, out, ldo, B, heads, Nq, Nkv, Nkv_pad, use_exp2, s);
    else launch_qf<1>(Q, K, Vt, out, ldo, B, heads, Nq, Nkv, Nkv_pad, use_exp2, s);
    return PMHIP_OK;
}
