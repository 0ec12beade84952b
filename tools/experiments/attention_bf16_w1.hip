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
// The work of ONE workgroup (4 waves x 16 QF queries of one (batch, head), first query `qbase`) as a device function: it is the
// body of attention_bf16_kernel below, and -- ONLY_EXACT, every half-tile through the exact path -- the re-run of a workgroup of
// the one-wave-per-SIMD kernel (attention_bf16_w1_kernel) whose fast path overflowed.  start_mask: the 16-query tiles of this
// wave that the exact attempt stores (ONLY_EXACT), all of them otherwise.
template <bool EXP2, int QF, bool ONLY_EXACT>
__device__ __forceinline__ void attention_wg(unsigned char* lds, int* redo_vote, const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kp,
                                             const bf16_t* __restrict__ Vt, bf16_t* __restrict__ out, int ldo, int heads, int Nq, int Nkv,
                                             int Nkv_pad, int bh, int qbase, unsigned start_mask) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int b = bh / heads, h = bh % heads;
    const int q0 = qbase + wave * (QF * 16);

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
    const int q0u = qbase + wave_u * (QF * 16);                      // q0, provably wave-uniform (qbase derives from blockIdx)

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
    const int nh_full = 2 * (Nkv / KT);                      // half-tiles that lie in full tiles
    // A context without a ragged tile (self-attention: every stage-2 / ViT launch of the decode loop) runs ALL its half-tiles,
    // the last two included, through the fast step.  Past the end `step` still computes S^T(h+1) and prefetches K(h+2) / V^T(h+1):
    // they address ring stages that still hold already-consumed tiles (no DMA is issued past the last tile), and nothing they
    // produce is consumed.
    const bool all_steady = (Nkv % KT) == 0 && ntiles >= 3;
    const int steady_end = all_steady ? nhalves - 1 : min(nhalves - 3, nh_full - 2);   // one bound: the loop's shape is unchanged
    bool exact = ONLY_EXACT || steady_end <= 0;              // workgroup-uniform: no fast step at all, or second attempt
    unsigned redo_mask = start_mask;                              // tiles the exact attempt stores (all, unless it is a second attempt)

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

template <bool EXP2, int QF>
__global__ __launch_bounds__(THREADS, 2) void attention_bf16_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kp,
                                                                    const bf16_t* __restrict__ Vt, bf16_t* __restrict__ out,
                                                                    int ldo, int heads, int Nq, int Nkv, int Nkv_pad, int nqb) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[RING * STAGE_BYTES];   // K / V^T ring
    __shared__ int redo_vote[4];
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
    attention_wg<EXP2, QF, false>(lds, redo_vote, Q, Kp, Vt, out, ldo, heads, Nq, Nkv, Nkv_pad, bh, qblk * (4 * QF * 16), ~0u);
}


// ------------------------------------------------------------------------------------------------------------------------------
// Fifth generation (round 6): ONE wave per SIMD, 128 queries per wave, and a register file the kernel owns.
//
// Why.  Round 5's ablations put 22 % of the fourth-generation kernel in operand movement: every wave reads the whole K / V^T tile
// from LDS (one fragment feeds 4 MFMAs) and issues a quarter of the tile's DMA.  128 queries per wave halve both per flop; that is
// ONE wave per SIMD with the whole 512-register file.  hipcc does not place that layout by itself (given the MFMA builtins it used
// the accumulator half as spill space: 208 v_accvgpr_read + 88 v_accvgpr_write per tile), so here the placement is stated:
//   AGPRs (224): O^T 8 x 4 accumulator quads, l 8 quads, the Q fragments 8 x 2 x 4 -- only ever touched by MFMAs in the loop
//                (accumulate in place / B operand), through inline-asm MFMAs with "a" constraints;
//   VGPRs (~200): S^T of one half-tile (64), -m (32), P (32), the K and V^T fragments double-buffered (64), addresses.
// Every instruction of the steady loop is an asm volatile statement, so the stream is exactly the source order (hipcc only
// allocates registers); the hazards hipcc would pad are kept apart by construction: an S^T quad is read by v_exp a whole group (9
// MFMAs) after the MFMA that wrote it, P is packed a group before the MFMA that reads it, a pack follows its exponentials by >= 2
// instructions, and the accumulators are read by VALU only behind the s_nops after the loop.
//
// Arithmetic: per 16-query tile exactly the fourth generation's -- same MFMA chains in the same half-tile order, same fixed
// reference maximum, same exp2 / pack / row-sum-by-MFMA -- so the bits of a tile do not depend on which kernel computed it (batch
// invariance), and the post-hoc overflow vote is the same; a workgroup that fails it stores its good tiles and re-runs as two
// fourth-generation exact workgroups (attention_wg<.., ONLY_EXACT>).
// Taken by pm_attention_bf16 for exp2 launches with whole 64-key tiles, >= 3 of them, Nq % 512 == 0 and enough workgroups.
// ------------------------------------------------------------------------------------------------------------------------------
#define W1_MFMA_QK0(d, kfr, qa, c) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(kfr), "a"(qa), "v"(c))
#define W1_MFMA_QK1(d, kfr, qa) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(kfr), "a"(qa))
#define W1_MFMA_ACC(d, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b))
#if ABL & 1
#define W1_EXP(d, s) asm volatile("v_mul_f32 %0, 1.0, %1" : "=v"(d) : "v"(s))
#else
#define W1_EXP(d, s) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(s))
#endif
#define W1_PK(d, lo, hi) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi))

#define W1_DSR_A(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(addr), "n"(off))

#ifdef PM_ATTN_W1_TIMING
// DEBUG BUILD ONLY (tools/hwtests/attn_w1.hip): shader cycles per phase, summed over all (wave, item) pairs: [0] pairs, [1] cold
// start (entry -> Q in registers and tile 0 entered; first items only), [2] first half-tile, [3] steady loop, [4] seam (vote,
// O -> LDS, accumulators zeroed), [5] final flush of the stores
__device__ unsigned long long g_w1_times[8];
#define W1_STAMP(var) var = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
#define W1_STAMP(var)
#endif
constexpr int W1_QF = 8;                                     // 16-query tiles per wave
constexpr int W1_QUERIES = 4 * W1_QF * 16;                   // per work item
constexpr int W1_XBYTES = W1_QF * 16 * 128;                  // per wave: 128 rows of 64 bf16 (the next item's Q, then this item's O)
constexpr int W1_MIN_TILES = 16;                             // the store / Q-prefetch schedule below needs 16 tiles per item

// PERSISTENT and PIPELINED ACROSS WORK ITEMS (one item = 512 queries of one (batch, head)).  Measured on the first form of this
// kernel (one item per workgroup): of 104 k cycles per workgroup only 59 k were the key loop -- 21 k went to waiting for Q, 12 k to
// issuing the output stores and 10 k to their completion, because all 256 CUs run equal items in lockstep and so load 25 MB and
// store 16 MB in the same few microseconds while HBM idles during the loops.  Here a workgroup walks items w, w + G, ... and
//   * the K / V^T tiles form ONE stream through the 4-stage ring: the last tiles of an item request the first tiles of the next;
//   * the next item's Q is DMA'd into a wave-private 16 KiB LDS area X during the second half of the item (one 1 KiB piece per
//     half-tile step) and read into the Q registers during the item's last step, which has no S^T product;
//   * the item's O^T / l goes to X as bf16 rows at the seam (2 k cycles) and is stored during the first half of the NEXT item,
//     one 1 KiB store per half-tile step; the last item flushes its stores before the kernel ends.
// X is wave-private (a wave reads and writes only the rows of its own 128 queries), so only the ring needs workgroup barriers.
// Counted waits: every vmcnt wait names the number of operations that are CERTAINLY younger than the one it needs (the stores and Q
// pieces come and go); vmcnt retires in issue order, stores included (tools/hwtests/vmcnt_order.hip).
__global__ __launch_bounds__(THREADS, 1) void attention_bf16_w1_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kp,
                                                                       const bf16_t* __restrict__ Vt, bf16_t* __restrict__ out,
                                                                       int ldo, int heads, int Nq, int Nkv, int Nkv_pad, int nqb, int nitems) {
    constexpr int QF = W1_QF;
#ifdef PM_ATTN_W1_TIMING
    unsigned long long t_a = 0, t_b = 0, t_c = 0, t_d = 0, t_e = 0, acc_t[6] = {0, 0, 0, 0, 0, 0};
#endif
    W1_STAMP(t_a)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[RING * STAGE_BYTES + 4 * W1_XBYTES];   // K / V^T ring, then X
    __shared__ __attribute__((aligned(16))) int redo_vote[4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned v_row_bytes = (unsigned)Nkv_pad * 2u;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned lslot = (unsigned)(((lane & 7) ^ ((lane >> 3) & 7)) << 4);
    const unsigned kvoff = (unsigned)(lane >> 3) * 128u + lslot;           // also the lane offset of a Q piece (128-byte rows)
    const unsigned kvoff1 = PM_ATTN_KSWZ ? kvoff ^ 64u : kvoff;
    const unsigned vvoff = (unsigned)(lane >> 3) * v_row_bytes + lslot;
    const unsigned kfrag_lane = lds_base + (unsigned)(8 * (l15 >> 2) + (l15 & 3)) * 128u + (unsigned)((g4 ^ (l15 & 3)) << 4) +
                                (PM_ATTN_KSWZ ? (unsigned)(((l15 >> 2) & 1) << 6) : 0u);
    const unsigned kfrag_laneB = kfrag_lane ^ 64u;
    const unsigned vfrag_lane0 = lds_base + 8192u + (unsigned)l15 * 128u + (unsigned)(((0 + g4) ^ (l15 & 7)) << 4);
    const unsigned vfrag_lane1 = lds_base + 8192u + (unsigned)l15 * 128u + (unsigned)(((4 + g4) ^ (l15 & 7)) << 4);
    // X of this wave: row r of the wave's 128 queries at r * 128, 16-byte slot s of the row at slot s ^ (r & 7)
    unsigned char* xw = lds + RING * STAGE_BYTES + wave_u * W1_XBYTES;
    const unsigned xw_base = lds_base + (unsigned)(RING * STAGE_BYTES) + (unsigned)wave_u * (unsigned)W1_XBYTES;
    const unsigned xq_lane0 = xw_base + (unsigned)l15 * 128u + (unsigned)(((0 + g4) ^ (l15 & 7)) << 4);      // Q fragment, d chunk 0 (+ qf * 2048)
    const unsigned xq_lane1 = xw_base + (unsigned)l15 * 128u + (unsigned)(((4 + g4) ^ (l15 & 7)) << 4);      //             d chunk 1
    const unsigned xo_lane = xw_base + (unsigned)l15 * 128u + (unsigned)(g4 & 1) * 8u;                        // O staging write: + ((df*2 + (g4>>1)) ^ (l15&7)) * 16 + qf * 2048
    const unsigned xs_lane = xw_base + kvoff;                                                                  // O store read: row lane >> 3 of the piece, swizzled slot

    const int ntiles = Nkv / KT;
    const int nhalves = 2 * ntiles;

    // work item -> (batch * heads + head, query block): items w, w + G, ... of one workgroup share w % 8, i.e. the XCD whose L2 holds
    // the (batch, head)'s K / V^T for both query blocks (G % 8 == 0)
    auto item_bh_q = [&](int item, int& bh, int& qbase) {
        const int total_bh = nitems / nqb;
        int qblk;
        if ((total_bh & 7) == 0) { const int slot = item >> 3; qblk = slot % nqb; bh = (slot / nqb) * 8 + (item & 7); }
        else { qblk = item % nqb; bh = item / nqb; }
        qbase = qblk * W1_QUERIES;
    };
    auto kv_rsrc = [&](int bh, rsrc_t& kr, rsrc_t& vr) {
        kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(Kp + (size_t)bh * Nkv_pad * DH)), 0, 0x7fffffff, 0x00020000);
        vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(Vt + (size_t)bh * DH * Nkv_pad)), 0, 0x7fffffff, 0x00020000);
    };

    // ---- registers the kernel owns
    v4u_t qa[QF][2];                     // AGPR: Q fragments (column operand of S^T)
    f32x4_t o[4][QF];                    // AGPR: O^T accumulators
    f32x4_t lacc[QF];                    // AGPR: row sums (every element = l of the query column)
    f32x4_t negm[QF];                    // -m_ref x4: the C operand of the S^T MFMAs
    f32x4_t s0[QF], s1[QF];              // S^T of ONE half-tile: key sub-tiles kk = 0 / 1
    unsigned pf[QF][4];                  // P^T operand of the 16-query tiles (bf16 pairs)
    v4u_t kfE[2][2], kfO[2][2], vfE[4], vfO[4];       // K / V^T fragments of the even / odd half-tiles
    v4u_t ostg;                          // one 1 KiB piece of the previous item's O on its way from X to global memory
    v4u_t ones = v4u_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    asm volatile("" : "+v"(ones));
    asm volatile("" : "=v"(ostg));

    // ---- item state (workgroup-uniform)
    int item = blockIdx.x;
    int bh, qbase;
    item_bh_q(item, bh, qbase);
    rsrc_t Kr, Vr, Krn, Vrn, Qrn;        // K / V^T of this item; K / V^T / Q of the next one
    kv_rsrc(bh, Kr, Vr);
    Krn = Kr; Vrn = Vr; Qrn = Kr;
    bool has_next = false;
    unsigned q0n_bytes = 0;              // byte offset of the next item's first query row of this wave in its (batch, head)'s Q
    // deferred stores of the previous item: destination of the wave's row 0, per-tile good mask (no previous item: 0)
    unsigned char* prev_rowbase = nullptr;
    unsigned prev_good = 0;
    bool have_q = false;                 // the Q registers already hold this item's Q (read from X during the previous item)

#define W1_KADDR(hh) ((unsigned)(((hh) >> 1) % RING) * STAGE_BYTES + (unsigned)((hh) & 1) * 4096u)
#define W1_VADDR(hh) ((((hh) & 1) ? vfrag_lane1 : vfrag_lane0) + (unsigned)(((hh) >> 1) % RING) * STAGE_BYTES)

    // One of this wave's four DMA pieces (K chunk 0 / V^T chunk 0 / K chunk 1 / V^T chunk 1) of stream tile t of the CURRENT item;
    // t >= ntiles is tile t - ntiles of the NEXT item (ntiles % RING == 0: its ring stage is t % RING all the same).
    auto dma_piece = [&](int t, int piece) {
        if (ABL & 4) return;
        __builtin_amdgcn_sched_barrier(0);
        const bool nxt = t >= ntiles;
        if (!nxt || has_next) {
            unsigned char* stage = lds + (t % RING) * STAGE_BYTES;
            const unsigned kv0 = (unsigned)(nxt ? t - ntiles : t) * KT;
            const unsigned chunk = (unsigned)wave_u * 2 + (piece >> 1);
            if (ABL & 16) {      // experiment: what a plain register load of the same bytes costs the wave (the data goes nowhere)
                v4u_t dummy;
                const unsigned vo = (piece & 1) ? vvoff : ((piece >> 1) ? kvoff1 : kvoff);
                const unsigned so = (piece & 1) ? chunk * 8 * v_row_bytes + kv0 * 2u : (kv0 + chunk * 8) * 128u;
                if (piece & 1) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dummy) : "v"(vo), "s"(nxt ? Vrn : Vr), "s"(so));
                else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dummy) : "v"(vo), "s"(nxt ? Krn : Kr), "s"(so));
            } else if (!(piece & 1))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(nxt ? Krn : Kr, (__attribute__((address_space(3))) void*)(stage + chunk * 1024), 16,
                                                         (piece >> 1) ? kvoff1 : kvoff, (kv0 + chunk * 8) * 128u, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(nxt ? Vrn : Vr, (__attribute__((address_space(3))) void*)(stage + KT * 128 + chunk * 1024), 16, vvoff,
                                                         chunk * 8 * v_row_bytes + kv0 * 2u, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // the extras of half-tile step hh (tile T = hh >> 1):
    //   T < 8:                       piece hh of the previous item's O: X -> register in this step (o_fetch), register -> global in
    //                                the NEXT step (o_store: the step's opening lgkmcnt(0) has retired the read)
    //   ntiles - 9 <= T < ntiles - 1: piece hh - 2 (ntiles - 9) of the next item's Q: global -> X
    auto o_fetch_at = [&](int hh) {
        if (hh < 16 && prev_good) {
            const unsigned a = xs_lane + (unsigned)hh * 1024u;
            DSRX(ostg, a, 0);
        }
    };
    auto o_store = [&](int hh) {         // piece hh - 1, fetched in the previous step
        const int p = hh - 1;
        if (p >= 0 && p < 16 && ((prev_good >> (p >> 1)) & 1u)) {
            __builtin_amdgcn_sched_barrier(0);
            v4u_t* dst = reinterpret_cast<v4u_t*>(prev_rowbase + ((unsigned)(lane >> 3) * (unsigned)ldo + (unsigned)(lane & 7) * 8u) * 2u +
                                                  (size_t)(p * 8) * (size_t)ldo * 2u);
            __builtin_nontemporal_store(ostg, dst);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto q_piece = [&](int hh) {
        const int p = hh - 2 * (ntiles - 9);
        if (p >= 0 && p < 16 && has_next && !(ABL & 4)) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Qrn, (__attribute__((address_space(3))) void*)(xw + p * 1024), 16, kvoff, q0n_bytes + (unsigned)p * 1024u, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // entering stream tile tn of the current item (tn == ntiles: tile 0 of the next one): its DMA has landed -- at least the four
    // pieces of the tile after it are younger -- the barrier publishes it and proves every wave done with tile tn - 2, whose ring
    // stage the pieces requested from now on reuse.  The last item has nothing younger in its last two tiles.
    auto enter_tile = [&](int tn) {
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 8)) {
            if (has_next || tn + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- one half-tile step: see the first form of this kernel (tools/ab_variants) -- group g issues S^T(hh+1, g) (4 MFMAs),
    //      O^T(., g) += V^T(hh) P(hh, g) and l_g += 1 P(hh, g) (5), and P(hh, g+1) = bf16(exp2(S^T(hh, g+1))) interleaved
    //      M T M T M TP M T M TP M T M TP M T M P; fragment reads for the next steps and this step's memory operations sit between groups
#define W1_GROUP(g, kN, vC, DOQK)                                                                                        \
    {                                                                                                                    \
        constexpr int nx = ((g) + 1) % QF;                                                                               \
        constexpr bool DOEXP = (DOQK) || (g) + 1 < QF;      /* last step: P(hh, g+1) still, but no P(hh+1, 0) */          \
        float e[8];                                                                                                      \
        const v4u_t pv = v4u_t{pf[g][0], pf[g][1], pf[g][2], pf[g][3]};                                                  \
        if (DOQK) W1_MFMA_QK0(s0[g], kN[0][0], qa[g][0], negm[g]);                                                       \
        if (DOEXP) { W1_EXP(e[0], s0[nx][0]); }                                                                          \
        if (DOQK) W1_MFMA_QK0(s1[g], kN[1][0], qa[g][0], negm[g]);                                                       \
        if (DOEXP) { W1_EXP(e[1], s0[nx][1]); }                                                                          \
        if (DOQK) W1_MFMA_QK1(s0[g], kN[0][1], qa[g][1]);                                                                \
        if (DOEXP) { W1_EXP(e[2], s0[nx][2]); W1_PK(pf[nx][0], e[0], e[1]); }                                            \
        if (DOQK) W1_MFMA_QK1(s1[g], kN[1][1], qa[g][1]);                                                                \
        if (DOEXP) { W1_EXP(e[3], s0[nx][3]); }                                                                          \
        W1_MFMA_ACC(o[0][g], vC[0], pv);                                                                                 \
        if (DOEXP) { W1_EXP(e[4], s1[nx][0]); W1_PK(pf[nx][1], e[2], e[3]); }                                            \
        W1_MFMA_ACC(o[1][g], vC[1], pv);                                                                                 \
        if (DOEXP) { W1_EXP(e[5], s1[nx][1]); }                                                                          \
        W1_MFMA_ACC(o[2][g], vC[2], pv);                                                                                 \
        if (DOEXP) { W1_EXP(e[6], s1[nx][2]); W1_PK(pf[nx][2], e[4], e[5]); }                                            \
        W1_MFMA_ACC(o[3][g], vC[3], pv);                                                                                 \
        if (DOEXP) { W1_EXP(e[7], s1[nx][3]); }                                                                          \
        W1_MFMA_ACC(lacc[g], ones, pv);                                                                                  \
        if (DOEXP) { W1_PK(pf[nx][3], e[6], e[7]); }                                                                     \
    }
#define W1_RDK_A(kD, hh2) { const unsigned so_ = W1_KADDR(hh2), ka_ = kfrag_lane + so_, kb_ = kfrag_laneB + so_; DSRX(kD[0][0], ka_, 0 * 512); DSRX(kD[0][1], kb_, 0 * 512); }
#define W1_RDK_B(kD, hh2) { const unsigned so_ = W1_KADDR(hh2), ka_ = kfrag_lane + so_, kb_ = kfrag_laneB + so_; DSRX(kD[1][0], kb_, 1 * 512); DSRX(kD[1][1], ka_, 1 * 512); }
#define W1_RDV_A(vD, hh1) { const unsigned va_ = W1_VADDR(hh1); DSRX(vD[0], va_, 0 * 2048); DSRX(vD[1], va_, 1 * 2048); }
#define W1_RDV_B(vD, hh1) { const unsigned va_ = W1_VADDR(hh1); DSRX(vD[2], va_, 2 * 2048); DSRX(vD[3], va_, 3 * 2048); }
    // all fragment reads of the previous step (and its O piece) have landed: they were issued >= 4 groups ago, the wait is free
#define W1_LANDED(kN, vC) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kN[0][0]), "+v"(kN[0][1]), "+v"(kN[1][0]), "+v"(kN[1][1]), \
                                       "+v"(vC[0]), "+v"(vC[1]), "+v"(vC[2]), "+v"(vC[3]), "+v"(ostg))
    // the fragment registers of a step stay allocated to its end (hipcc otherwise hands a dying fragment's registers to the
    // exponentials that follow the asm MFMA reading it)
#define W1_KEEP(kN, vC) asm volatile("" :: "v"(kN[0][0]), "v"(kN[0][1]), "v"(kN[1][0]), "v"(kN[1][1]), "v"(vC[0]), "v"(vC[1]), "v"(vC[2]), "v"(vC[3]))
    // this wave's 16 Q fragment quads of the next item, X -> the Q registers, four per call (the last step has no S^T product, the
    // old fragments are dead); the Q pieces were requested up to two steps ago: a counted wait precedes the first call
#define W1_RDQ(i) { W1_DSR_A(qa[2 * (i)][0], xq_lane0, (2 * (i)) * 2048); W1_DSR_A(qa[2 * (i)][1], xq_lane1, (2 * (i)) * 2048);     \
                    W1_DSR_A(qa[2 * (i) + 1][0], xq_lane0, (2 * (i) + 1) * 2048); W1_DSR_A(qa[2 * (i) + 1][1], xq_lane1, (2 * (i) + 1) * 2048); }

    // The memory operations of a step -- store of an O piece, Q piece, K piece, V^T piece -- go behind groups 3..6, one each (every one
    // costs its wave about 100 cycles of issue, plain register loads included: tools/hwtests/attn_w1.hip, ABL 16).
    // (Measured and dropped: wave-dependent positions (2 k + wave) % 8, so that the four waves do not issue at the same point behind
    // a barrier -- the position tests are scalar branches at every group boundary, and with one wave per SIMD nothing hides a taken
    // branch: 3131 instead of 1927 cycles per half-tile step.)
    const int pos_s = 3, pos_q = 4, pos_k = 5, pos_v = 6;
#define W1_MEMOPS(g, kpiece)                                                                     \
    {                                                                                            \
        if (pos_s == (g)) o_store(hh);                                                           \
        if (pos_q == (g)) q_piece(hh);                                                           \
        if (pos_k == (g)) dma_piece((hh >> 1) + 3, kpiece);                                      \
        if (pos_v == (g)) dma_piece((hh >> 1) + 3, (kpiece) + 1);                                \
    }
    auto step_even = [&](auto qk_c, int hh) {                        // uses K(hh+1) = kfO, V^T(hh) = vfE; fills kfE, vfO
        constexpr bool DOQK = decltype(qk_c)::value;
        const bool nextk = has_next || hh + 2 < nhalves;              // a further tile (of this item or the next) to enter
        W1_LANDED(kfO, vfE);
        W1_GROUP(0, kfO, vfE, DOQK)
        W1_RDV_A(vfO, hh + 1)
        W1_MEMOPS(0, 0)
        W1_GROUP(1, kfO, vfE, DOQK)
        W1_RDV_B(vfO, hh + 1)
        W1_MEMOPS(1, 0)
        W1_GROUP(2, kfO, vfE, DOQK)
        if (nextk) { enter_tile((hh + 2) >> 1); W1_RDK_A(kfE, hh + 2) }
        W1_MEMOPS(2, 0)
        W1_GROUP(3, kfO, vfE, DOQK)
        if (nextk) { W1_RDK_B(kfE, hh + 2) }
        W1_MEMOPS(3, 0)
        W1_GROUP(4, kfO, vfE, DOQK)
        W1_MEMOPS(4, 0)
        W1_GROUP(5, kfO, vfE, DOQK)
        W1_MEMOPS(5, 0)
        W1_GROUP(6, kfO, vfE, DOQK)
        W1_MEMOPS(6, 0)
        o_fetch_at(hh);
        W1_GROUP(7, kfO, vfE, DOQK)
        W1_MEMOPS(7, 0)
        W1_KEEP(kfO, vfE);
    };
    auto step_odd = [&](auto qk_c, int hh) {                          // uses K(hh+1) = kfE, V^T(hh) = vfO; fills kfO, vfE
        constexpr bool DOQK = decltype(qk_c)::value;
        W1_LANDED(kfE, vfO);
        if (!DOQK && has_next) {                                      // last step: the next item's Q, X -> registers
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");          // the last Q piece went out two steps ago; the K / V^T pieces of the step in between are younger
            __builtin_amdgcn_sched_barrier(0);
        }
        W1_GROUP(0, kfE, vfO, DOQK)
        if (DOQK) { W1_RDK_A(kfO, hh + 2) } else if (has_next) { W1_RDQ(0) }
        W1_MEMOPS(0, 2)
        W1_GROUP(1, kfE, vfO, DOQK)
        if (DOQK) { W1_RDK_B(kfO, hh + 2) } else if (has_next) { W1_RDQ(1) }
        W1_MEMOPS(1, 2)
        W1_GROUP(2, kfE, vfO, DOQK)
        if (DOQK) { W1_RDV_A(vfE, hh + 1) } else if (has_next) { W1_RDQ(2) }
        W1_MEMOPS(2, 2)
        W1_GROUP(3, kfE, vfO, DOQK)
        if (DOQK) { W1_RDV_B(vfE, hh + 1) } else if (has_next) { W1_RDQ(3) }
        W1_MEMOPS(3, 2)
        W1_GROUP(4, kfE, vfO, DOQK)
        W1_MEMOPS(4, 2)
        W1_GROUP(5, kfE, vfO, DOQK)
        W1_MEMOPS(5, 2)
        W1_GROUP(6, kfE, vfO, DOQK)
        W1_MEMOPS(6, 2)
        o_fetch_at(hh);
        W1_GROUP(7, kfE, vfO, DOQK)
        W1_MEMOPS(7, 2)
        W1_KEEP(kfE, vfO);
    };
    constexpr std::true_type Y{};
    constexpr std::false_type N{};

    for (;;) {
        // ---- the next item of this workgroup
        const int item_n = item + (int)gridDim.x;
        has_next = item_n < nitems;
        int bhn = bh, qbasen = qbase;
        if (has_next) {
            item_bh_q(item_n, bhn, qbasen);
            kv_rsrc(bhn, Krn, Vrn);
            Qrn = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(Q + (size_t)bhn * Nq * DH)), 0, 0x7fffffff, 0x00020000);
            q0n_bytes = (unsigned)(qbasen + wave_u * (QF * 16)) * 128u;
        }
        const int b = bh / heads, h = bh % heads;
        const int q0u = qbase + wave_u * (QF * 16);
        unsigned char* rowbase = reinterpret_cast<unsigned char*>(out + ((size_t)b * Nq + q0u) * ldo + h * DH);

        if (!have_q) {
            // ---- cold start (the workgroup's first item, or behind a fallback): Q straight from global memory, tiles 0..2 requested here
            const bf16_t* Qbh = Q + (size_t)bh * Nq * DH;
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) {
                const unsigned char* qrow = reinterpret_cast<const unsigned char*>(Qbh + (size_t)(q0u + qf * 16 + l15) * DH);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    qa[qf][c] = *reinterpret_cast<const v4u_t*>(qrow + (c * 4 + g4) * 16);
                    asm volatile("" : "+a"(qa[qf][c]));
                }
            }
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int pc = 0; pc < 4; ++pc) dma_piece(t, pc);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // tile 0 (tiles 1 and 2 are younger)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the Q registers read from X during the last step
        }
        W1_STAMP(t_b)
#pragma unroll
        for (int j = 0; j < QF; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { o[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+a"(o[i][j])); }
            lacc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+a"(lacc[j]));
            negm[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+v"(negm[j]));    // materialised HERE: hipcc otherwise writes the zero quad in the instruction before the
        }                                        // first asm MFMA that reads it as C (VALU write -> MFMA operand needs wait states it
                                                 // cannot know about: the first MFMA's tile came out wrong in ~90 % of the workgroups)

        // ---- first half-tile: S^T(0) from C = 0, the reference maximum of every query, S^T(0) - m_ref (exactly attention_wg's
        //      qk + rescale(first) for a context without ragged tiles).  Tile 0 has been entered (cold start above / the previous
        //      item's last tile).
        {
            const unsigned so = W1_KADDR(0), ka = kfrag_lane + so, kb = kfrag_laneB + so;
            DSRX(kfE[0][0], ka, 0 * 512); DSRX(kfE[0][1], kb, 0 * 512); DSRX(kfE[1][0], kb, 1 * 512); DSRX(kfE[1][1], ka, 1 * 512);
            LGKM4(0, kfE[0][0], kfE[0][1], kfE[1][0], kfE[1][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 3" ::: "memory");    // whatever hipcc wrote last (an operand of the first MFMA?) has been written
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) W1_MFMA_QK0(s0[qf], kfE[0][0], qa[qf][0], negm[qf]);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) W1_MFMA_QK1(s0[qf], kfE[0][1], qa[qf][1]);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) W1_MFMA_QK0(s1[qf], kfE[1][0], qa[qf][0], negm[qf]);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) W1_MFMA_QK1(s1[qf], kfE[1][1], qa[qf][1]);
        // MFMA result -> VALU: hipcc does not see the asm MFMAs and pads nothing; nothing may be scheduled across the wait states
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            float m = vmax3(s0[qf][0], s0[qf][1], s0[qf][2]);
            m = vmax3(m, s0[qf][3], s1[qf][0]);
            m = vmax3(m, s1[qf][1], s1[qf][2]);
            m = vmax2(m, s1[qf][3]);
            const float mnew = vmax3(-INFINITY, group4_max(m) + 0.f, -1e30f);
            const float delta = 0.f - mnew;
#pragma unroll
            for (int r = 0; r < 4; ++r) { s0[qf][r] += delta; s1[qf][r] += delta; }
            negm[qf][0] = delta; negm[qf][1] = delta; negm[qf][2] = delta; negm[qf][3] = delta;
        }
        {   // K(1), V^T(0); P(0, 0)
            const unsigned so = W1_KADDR(1), ka = kfrag_lane + so, kb = kfrag_laneB + so, va = W1_VADDR(0);
            DSRX(kfO[0][0], ka, 0 * 512); DSRX(kfO[0][1], kb, 0 * 512); DSRX(kfO[1][0], kb, 1 * 512); DSRX(kfO[1][1], ka, 1 * 512);
            DSRX(vfE[0], va, 0 * 2048); DSRX(vfE[1], va, 1 * 2048); DSRX(vfE[2], va, 2 * 2048); DSRX(vfE[3], va, 3 * 2048);
            float e[8];
            W1_EXP(e[0], s0[0][0]); W1_EXP(e[1], s0[0][1]); W1_EXP(e[2], s0[0][2]); W1_EXP(e[3], s0[0][3]);
            W1_EXP(e[4], s1[0][0]); W1_EXP(e[5], s1[0][1]); W1_EXP(e[6], s1[0][2]); W1_EXP(e[7], s1[0][3]);
            W1_PK(pf[0][0], e[0], e[1]); W1_PK(pf[0][1], e[2], e[3]); W1_PK(pf[0][2], e[4], e[5]); W1_PK(pf[0][3], e[6], e[7]);
        }
        __builtin_amdgcn_sched_barrier(0);
        W1_STAMP(t_c)
        __builtin_amdgcn_sched_barrier(0);

        // ---- the key loop
        int hs = 0;
        for (; hs < nhalves - 2; hs += 2) {
            step_even(Y, hs);
            step_odd(Y, hs + 1);
        }
        step_even(Y, hs);                // the last tile
        step_odd(N, hs + 1);             // ... whose second half has no S^T product: the next item's Q comes in instead
        __builtin_amdgcn_sched_barrier(0);                   // accumulator MFMAs -> v_accvgpr_read: as above
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        W1_STAMP(t_d)
        __builtin_amdgcn_sched_barrier(0);

        // ---- the seam.  Vote: did a probability of the fixed-reference path leave the f32 range?  (attention_wg's test, per tile.)
        //      LDS traffic by asm only: hipcc would drain vmcnt to 0 -- the stream's tiles in flight -- in front of an LDS access it sees.
        unsigned badmask = 0;
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) badmask |= __any(!(lacc[qf][0] < 1.8446744e19f)) ? (1u << qf) : 0u;
        if (ABL & 23) badmask = 0;                            // (ablations that compute garbage do not vote)
        {
            const unsigned va = (unsigned)(size_t)(__attribute__((address_space(3))) int*)redo_vote;
            unsigned bm = badmask;
            unsigned vaddr = va + (unsigned)wave_u * 4u;
            asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" :: "v"(vaddr), "v"(bm) : "memory");
            v4u_t votes;
            unsigned va0 = va;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(votes) : "v"(va0) : "memory");
            const unsigned any = __builtin_amdgcn_readfirstlane(votes[0] | votes[1] | votes[2] | votes[3]);
            // the previous item's stores all went out in the first half of this item; its staging area X held the next item's Q,
            // now in the Q registers: X is free for this item's O
            const unsigned good = any ? ~badmask & 0xffu : 0xffu;
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) {
                const float inv = 1.0f / lacc[qf][0];
#pragma unroll
                for (int df = 0; df < 4; ++df) {
                    const unsigned addr = xo_lane + (unsigned)(((df * 2 + (g4 >> 1)) ^ (l15 & 7)) << 4);
                    const uint2 val = make_uint2(pack_bf16x2(o[df][qf][0] * inv, o[df][qf][1] * inv), pack_bf16x2(o[df][qf][2] * inv, o[df][qf][3] * inv));
                    asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(val), "n"(qf * 2048) : "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            prev_rowbase = rowbase;
            prev_good = good;
            if (__builtin_expect(any != 0, 0)) {
                // Rare: some 16-query tile overflowed.  Its good tiles are in X like every item's; the two 256-query halves that hold a bad
                // tile run again as fourth-generation workgroups through the exact path and store exactly those tiles (tile t of
                // fourth-generation wave w in half j is tile (w & 1) * 4 + t of this kernel's wave 2 j + (w >> 1)).  The exact path takes
                // the ring: the stream is drained first and restarts cold with the next item.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                const unsigned vv[4] = {votes[0], votes[1], votes[2], votes[3]};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (__builtin_amdgcn_readfirstlane(vv[2 * j] | vv[2 * j + 1]) == 0) continue;
                    if (tid == 0) atomicAdd(&g_attn_fallbacks, 1ull);
                    const unsigned mine = ((wave_u < 2 ? vv[2 * j] : vv[2 * j + 1]) >> ((wave_u & 1) * 4)) & 0xfu;
                    attention_wg<true, 4, true>(lds, redo_vote, Q, Kp, Vt, out, ldo, heads, Nq, Nkv, Nkv_pad, bh, qbase + j * 256, mine);
                    __syncthreads();
                }
                have_q = false;                              // (the next item's Q registers are not trusted across the call)
            } else {
                have_q = has_next;
            }
        }
        W1_STAMP(t_e)
#ifdef PM_ATTN_W1_TIMING
        acc_t[0] += 1; acc_t[1] += t_b - t_a; acc_t[2] += t_c - t_b; acc_t[3] += t_d - t_c; acc_t[4] += t_e - t_d;
        t_a = t_e;
#endif
        if (!has_next) break;
        item = item_n; bh = bhn; qbase = qbasen;
        Kr = Krn; Vr = Vrn;
    }

    // ---- the last item's stores
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        if ((prev_good >> (p >> 1)) & 1u) {
            v4u_t v;
            const unsigned a = xs_lane + (unsigned)p * 1024u;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
            v4u_t* dst = reinterpret_cast<v4u_t*>(prev_rowbase + ((unsigned)(lane >> 3) * (unsigned)ldo + (unsigned)(lane & 7) * 8u) * 2u +
                                                  (size_t)(p * 8) * (size_t)ldo * 2u);
            __builtin_nontemporal_store(v, dst);
        }
    }
#ifdef PM_ATTN_W1_TIMING
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W1_STAMP(t_e)
        if (lane == 0) {
            for (int i = 0; i < 5; ++i) atomicAdd(&g_w1_times[i], acc_t[i]);
            atomicAdd(&g_w1_times[5], t_e - t_a);
        }
    }
#endif
}
#undef W1_DSR_A
#undef W1_RDQ
#undef W1_MEMOPS
#undef W1_MFMA_QK0
#undef W1_MFMA_QK1
#undef W1_MFMA_ACC
#undef W1_EXP
#undef W1_PK
#undef W1_GROUP
#undef W1_RDK
#undef W1_RDK_A
#undef W1_RDK_B
#undef W1_RDV_A
#undef W1_RDV_B
#undef W1_RDV
#undef W1_LANDED
#undef W1_KEEP
#undef W1_KADDR
#undef W1_VADDR

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
#ifndef PM_ATTN_NO_W1
    // the one-wave-per-SIMD kernel: whole tiles, at least three of them, whole 512-query workgroups and one or more per CU
    static const bool w1_on = [] { const char* e = getenv("PMHIP_ATTN_W1"); return e && atoi(e) != 0; }();     // (development: off by default)
    if ((force == 8 || (!force && w1_on && bh * (Nq / W1_QUERIES) >= 256)) && use_exp2 && Nkv % (KT * RING) == 0 && Nkv / KT >= W1_MIN_TILES &&
        Nq % W1_QUERIES == 0) {
        const int nqb = Nq / W1_QUERIES;
        const int nitems = nqb * B * heads;
        static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
        int grid = nitems < cus ? nitems : cus;
        if (grid >= 8) grid &= ~7;                           // items w, w + grid, ... of a workgroup stay on one XCD
        hipLaunchKernelGGL(attention_bf16_w1_kernel, dim3(grid), dim3(THREADS), 0, s, (const bf16_t*)Q, (const bf16_t*)K,
                           (const bf16_t*)Vt, (bf16_t*)out, ldo, heads, Nq, Nkv, Nkv_pad, nqb, nitems);
        return PMHIP_OK;
    }
#endif
    if (force == 4 || (!force && bh * ceil_div(Nq, 256) >= kFill)) launch_qf<4>(Q, K, Vt, out, ldo, B, heads, Nq, Nkv, Nkv_pad, use_exp2, s);
    else if (force == 2 || (!force && bh * ceil_div(Nq, 128) >= kFill)) launch_qf<2>(Q, K, Vt, out, ldo, B, heads, Nq, Nkv, Nkv_pad, use_exp2, s);
    else launch_qf<1>(Q, K, Vt, out, ldo, B, heads, Nq, Nkv, Nkv_pad, use_exp2, s);
    return PMHIP_OK;
}
