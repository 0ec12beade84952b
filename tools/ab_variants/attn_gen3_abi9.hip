// Fused softmax(Q K^T) V in bf16 for dim_head = 64 (gfx950), third generation.
// Replaces the reference's materialised-score attention (modules/attention.py:51-58: q@k^T -> softmax -> @v, a
// (B*H, N, N) fp32 tensor per layer) and its xformers alternative (:100).  Same data layout and work split as the
// f32 kernel in attention.hip (which stays the fp32-verify path): Q [B,H,Nq,64] pre-scaled, K [B,H,Np,64],
// V^T [B,H,64,Np]; one workgroup = 256 queries of one (batch, head), 4 waves x 64 queries; K / V^T tiles of 64 keys
// by DMA into a 3-stage LDS ring, one barrier per tile; swapped QK^T (a lane owns 8 consecutive keys of ONE query per
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
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#ifndef ABL
#define ABL 0                 // ablation bit mask (tools/hwtests/attn_abl.hip); 0 in the library
#endif

#ifdef PM_ATTN_COUNT
// DEBUG BUILD ONLY (tools/attn_rescale_count.sh): how often the steady loop leaves its fast path on real data.
// [0] wave-level executions of the rescale branch inside the steady loop, [1] steady half-tile steps (per wave), [2] slow steps
__device__ unsigned long long g_attn_counters[4];
extern "C" int pmhip_debug_attention_counters(unsigned long long* out4, int reset) {
    if (out4 && hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_attn_counters), 32) != hipSuccess) return 1;
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_counters), z, 32) != hipSuccess) return 1; }
    return 0;
}
#endif

extern "C" int pmhip_attention_fallbacks(unsigned long long* count, int) { *count = 0; return 0; }
namespace {

constexpr int KT = 64;        // keys per tile
constexpr int DH = 64;
constexpr int THREADS = 256;
constexpr int QF = 4;         // 16-query tiles per wave
constexpr int TILE_BYTES = KT * 128;
constexpr int STAGE_BYTES = 2 * TILE_BYTES;              // K tile + V^T tile

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

template <bool EXP2>
__global__ __launch_bounds__(THREADS, 2) void attention_bf16_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kp,
                                                                    const bf16_t* __restrict__ Vt, bf16_t* __restrict__ out,
                                                                    int ldo, int heads, int Nq, int Nkv, int Nkv_pad, int nqb) {
    constexpr float kDefer = EXP2 ? 8.0f : 0.0f;             // skip the O rescale while the row max grows < 2^8
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * STAGE_BYTES];   // 3-stage K / V^T ring

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
    //   V^T row 16 df + l15, slot (4 pc + g) ^ (l15 & 7)
    //     = stage + 8192 + l15 * 128 + ((4 pc + g) ^ (l15 & 7)) * 16  +  df * 2048
    const rsrc_t Kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Kbh), 0, 0x7fffffff, 0x00020000);
    const rsrc_t Vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Vbh), 0, 0x7fffffff, 0x00020000);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned lslot = (unsigned)(((lane & 7) ^ ((lane >> 3) & 7)) << 4);
    const unsigned kvoff = (unsigned)(lane >> 3) * 128u + lslot;
    const unsigned vvoff = (unsigned)(lane >> 3) * v_row_bytes + lslot;
    const unsigned kfrag_lane = lds_base + (unsigned)(8 * (l15 >> 2) + (l15 & 3)) * 128u + (unsigned)((g ^ (l15 & 3)) << 4);
    const unsigned vfrag_lane0 = lds_base + 8192u + (unsigned)l15 * 128u + (unsigned)(((0 + g) ^ (l15 & 7)) << 4);
    const unsigned vfrag_lane1 = lds_base + 8192u + (unsigned)l15 * 128u + (unsigned)(((4 + g) ^ (l15 & 7)) << 4);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    // one K tile + one V^T tile by DMA, 1 KiB per wave-instruction; the bank swizzle (slot ^ row) is applied to the SOURCE
    // address (kvoff / vvoff) and again on the read side
    auto stage_tiles = [&](int t) {
        unsigned char* stage = lds + (t % 3) * STAGE_BYTES;
        const unsigned kv0 = (unsigned)t * KT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned chunk = (unsigned)wave_u * 2 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Kr, (__attribute__((address_space(3))) void*)(stage + chunk * 1024), 16, kvoff,
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
    f32x4_t negm[QF];                    // -m (running max of the query column) x4: the C operand of the S^T MFMAs
#pragma unroll
    for (int j = 0; j < QF; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        lacc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        negm[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    v4u_t ones = v4u_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    asm volatile("" : "+v"(ones));       // keep it in registers (not re-materialised in front of every use)

    const int ntiles = (Nkv + KT - 1) / KT;
    const int nhalves = (Nkv + 31) / 32;                     // 32-key half-tiles that contain at least one valid key

    auto k_issue = [&](v4u_t (&kf)[2][2], int hh) {
        const unsigned ka = kfrag_lane + (unsigned)((hh >> 1) % 3) * STAGE_BYTES + (unsigned)(hh & 1) * 4096u;
        DSRX(kf[0][0], ka, 0 * 512 + 0 * 64); DSRX(kf[0][1], ka, 0 * 512 + 1 * 64);
        DSRX(kf[1][0], ka, 1 * 512 + 1 * 64); DSRX(kf[1][1], ka, 1 * 512 + 0 * 64);
    };
    auto v_issue = [&](v4u_t (&vf)[4], int hh) {
        const unsigned va = ((hh & 1) ? vfrag_lane1 : vfrag_lane0) + (unsigned)((hh >> 1) % 3) * STAGE_BYTES;
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
            const float alpha = EXP2 ? __builtin_amdgcn_exp2f(mold - mnew) : expf(mold - mnew);
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
            lacc[qf][0] *= alpha; lacc[qf][1] *= alpha; lacc[qf][2] *= alpha; lacc[qf][3] *= alpha;
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                o[df][qf][0] *= alpha; o[df][qf][1] *= alpha; o[df][qf][2] *= alpha; o[df][qf][3] *= alpha;
            }
        }
    };

    // entering tile tn (called while the previous tile's second half is still to be consumed): its DMA has landed
    // and is published by the barrier; the barrier also proves every wave is done with tile tn-2, whose stage the
    // DMA of tile tn+1 now reuses (3-stage ring)
    auto enter_tile = [&](auto ragged_c, int tn) {
        if (!(ABL & 8) || tn == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (tn + 1 < ntiles && !(ABL & 4)) stage_tiles(tn + 1);
        if (decltype(ragged_c)::value && tn * KT + KT > Nkv) {   // ragged last tile: zero the V^T columns of keys >= Nkv
            unsigned char* Vl = lds + (tn % 3) * STAGE_BYTES + TILE_BYTES;
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

    f32x4_t sA[2][QF], sB[2][QF];
    v4u_t pf[QF];
    v4u_t kf[2][2], vf[4];

    // exponentials of the 16-query tiles qa, qa+1 of S^T(h) (in place) and their packing into the P^T operand
    auto exp_pack = [&](f32x4_t (&sc)[2][QF], int qa) {
#pragma unroll
        for (int qf = qa; qf < qa + 2; ++qf) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sc[kk][qf][r] = (ABL & 1) ? sc[kk][qf][r] * 1.0001f : (EXP2 ? __builtin_amdgcn_exp2f(sc[kk][qf][r]) : expf(sc[kk][qf][r]));   // sc = s - m
            pf[qf] = v4u_t{pack_bf16x2(sc[0][qf][0], sc[0][qf][1]), pack_bf16x2(sc[0][qf][2], sc[0][qf][3]),
                           pack_bf16x2(sc[1][qf][0], sc[1][qf][1]), pack_bf16x2(sc[1][qf][2], sc[1][qf][3])};
        }
    };
    // P.V for two 16-row blocks of O^T and two of the four row-sum tiles
    auto pv2 = [&](int d0) {
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) o[d0][qf] = mma(vf[d0], pf[qf], o[d0][qf]);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) o[d0 + 1][qf] = mma(vf[d0 + 1], pf[qf], o[d0 + 1][qf]);
        lacc[d0] = mma(ones, pf[d0], lacc[d0]);
        lacc[d0 + 1] = mma(ones, pf[d0 + 1], lacc[d0 + 1]);
    };
    // growth check: max over the lane's 16 scores of S^T tile kk as SIGNED INTEGERS (exact whenever the maximum is >= 0,
    // negative otherwise: all the comparison against the threshold needs; no NaN canonicalisation, plain VALU)
    auto imax16 = [&](f32x4_t (&sn)[2][QF], int kk, int m) {
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            m = max(max(m, __float_as_int(sn[kk][qf][0])), __float_as_int(sn[kk][qf][1]));
            m = max(max(m, __float_as_int(sn[kk][qf][2])), __float_as_int(sn[kk][qf][3]));
        }
        return m;
    };

    // One half-tile h in the steady state (h+1 and h+2 exist, h+1 lies in a full tile).  On entry sc = S^T(h) - m, already
    // checked against growth, and the K fragments of h+1 are in flight (kf[0][*] older than kf[1][*]).  The K and V^T
    // fragments time-share registers: V^T of h is requested as the K fragments of h+1 are consumed, K of h+2 as V^T is.
    // Every wait is lgkmcnt(2): two reads older and two reads younger than the ones needed are outstanding.
    //   OPENS: h+2 is the first half of a new tile
    auto step = [&](auto opens_c, f32x4_t (&sc)[2][QF], f32x4_t (&sn)[2][QF], int hh) {
        constexpr bool OPENS = decltype(opens_c)::value;
        const unsigned va = ((hh & 1) ? vfrag_lane1 : vfrag_lane0) + (unsigned)((hh >> 1) % 3) * STAGE_BYTES;
        const unsigned ka = kfrag_lane + (unsigned)(((hh + 2) >> 1) % 3) * STAGE_BYTES + (unsigned)(hh & 1) * 4096u;
        // ---- A1: first key tile of S^T(h+1) under the exponentials of query tiles 0, 1
        LGKM2(2, kf[0][0], kf[0][1]);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) sn[0][qf] = mma(kf[0][0], qreg[qf][0], negm[qf]);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) sn[0][qf] = mma(kf[0][1], qreg[qf][1], sn[0][qf]);
        exp_pack(sc, 0);
        if constexpr (EXP2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);   // 2 transcendental
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // 1 VALU (a pack of two earlier exponentials)
            }
        }
        asm volatile("" : "+v"(pf[0]), "+v"(pf[1]));             // the packs are complete here (not sunk to their first use)
        __builtin_amdgcn_sched_barrier(0);
        DSRX(vf[0], va, 0 * 2048); DSRX(vf[1], va, 1 * 2048);
        // ---- A2: second key tile under query tiles 2, 3
        LGKM2(2, kf[1][0], kf[1][1]);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) sn[1][qf] = mma(kf[1][0], qreg[qf][0], negm[qf]);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) sn[1][qf] = mma(kf[1][1], qreg[qf][1], sn[1][qf]);
        exp_pack(sc, 2);
        if constexpr (EXP2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
        }
        asm volatile("" : "+v"(pf[2]), "+v"(pf[3]));
        __builtin_amdgcn_sched_barrier(0);
        DSRX(vf[2], va, 2 * 2048); DSRX(vf[3], va, 3 * 2048);
        if constexpr (OPENS) enter_tile(std::false_type{}, (hh + 2) >> 1);
        // ---- B1: rows 0..31 of O^T and two row-sum tiles under the growth check of the first key tile of S^T(h+1)
        LGKM2(2, vf[0], vf[1]);
        pv2(0);
        int m = (ABL & 16) ? 0 : imax16(sn, 0, (int)0x80000000);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        DSRX(kf[0][0], ka, 0 * 512 + 0 * 64); DSRX(kf[0][1], ka, 0 * 512 + 1 * 64);
        // ---- B2: rows 32..63 under the check of the second key tile
        LGKM2(2, vf[2], vf[3]);
        pv2(2);
        if (!(ABL & 16)) m = imax16(sn, 1, m);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        DSRX(kf[1][0], ka, 1 * 512 + 1 * 64); DSRX(kf[1][1], ka, 1 * 512 + 0 * 64);
        if constexpr (ABL & 16) asm volatile("" :: "v"(m));
        else if (__builtin_expect(__any(__int_as_float(m) > kDefer), 0)) {
#ifdef PM_ATTN_COUNT
            if (lane == 0) atomicAdd(&g_attn_counters[0], 1ull);
#endif
            rescale(std::false_type{}, std::false_type{}, sn, hh + 1);
        }
    };

    // The same half-tile with every condition at run time and full waits: first and last tiles, ragged tiles, short contexts.
    // Always sA -> sB, then sB is copied back.
    auto slow_step = [&](int hh) {
        const bool next = hh + 1 < nhalves, next2 = hh + 2 < nhalves;
        if (next) {
            LGKM4(0, kf[0][0], kf[0][1], kf[1][0], kf[1][1]);
            qk(sB, kf);
        }
        exp_pack(sA, 0);
        exp_pack(sA, 2);
        v_issue(vf, hh);
        if (next2 && !(hh & 1)) enter_tile(std::true_type{}, (hh + 2) >> 1);
        if (next2) k_issue(kf, hh + 2);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]), "+v"(kf[0][0]), "+v"(kf[0][1]),
                     "+v"(kf[1][0]), "+v"(kf[1][1]));
        pv2(0);
        pv2(2);
        if (next) {
            rescale(std::true_type{}, std::false_type{}, sB, hh + 1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int qf = 0; qf < QF; ++qf) sA[kk][qf] = sB[kk][qf];
        }
    };

    constexpr std::true_type Y{};
    constexpr std::false_type N{};

    stage_tiles(0);
    enter_tile(Y, 0);
    k_issue(kf, 0);
    LGKM4(0, kf[0][0], kf[0][1], kf[1][0], kf[1][1]);
    qk(sA, kf);
    rescale(Y, Y, sA, 0);
    if (nhalves > 1) k_issue(kf, 1);

    int hs = 0;
    const int nh_full = 2 * (Nkv / KT);                      // half-tiles that lie in full tiles
    // A context without a ragged tile (self-attention: every stage-2 / ViT launch of the decode loop) runs ALL its half-tiles,
    // the last two included, through the steady-state step.  Past the end `step` still computes S^T(h+1) and prefetches K(h+2):
    // both address ring stage ntiles % 3, which still holds tile ntiles - 3 (no DMA is issued past the last tile), i.e. finite
    // scores of keys that were already folded into the running max, so the growth check cannot fire on them and nothing they
    // produce is consumed: 16 wasted MFMAs per workgroup-wave instead of two full-wait slow steps with their unconditional
    // rescale (round 4; the last tile was 6 % of the half-tiles at N = 1024 and ran at less than half the steady-state speed).
    const bool all_steady = (Nkv % KT) == 0 && ntiles >= 3;
    const int steady_end = all_steady ? nhalves - 1 : min(nhalves - 3, nh_full - 2);   // one bound: the loop's shape is unchanged
    for (; hs < steady_end; hs += 2) {                       // steady state
        step(Y, sA, sB, hs);
        step(N, sB, sA, hs + 1);
    }
#ifdef PM_ATTN_COUNT
    if (lane == 0) { atomicAdd(&g_attn_counters[1], (unsigned long long)hs); atomicAdd(&g_attn_counters[2], (unsigned long long)(nhalves - hs)); }
#endif
    for (; hs < nhalves; ++hs) slow_step(hs);

    // ---- finalize: O = O^T / l, head-major inside the output row.  The wave's 64 output rows go through the (now idle)
    // K / V^T ring, so that every global store instruction writes 8 whole 128-byte rows (non-temporal)
    constexpr int RS = 144;                                // staged row: 64 bf16 + pad, 16-B aligned, conflict-free
    __syncthreads();                                       // every wave is done reading the ring
    unsigned char* obuf = lds + wave * (64 * RS);
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
#pragma unroll
    for (int it = 0; it < QF * 2; ++it) {                  // 8 rows x 128 B per store instruction
        const int r = it * 8 + (lane >> 3), c16 = lane & 7, q = q0 + r;
        if (q < Nq && (!(ABL & 32) || q < 0)) {
            const v4u_t v = *reinterpret_cast<const v4u_t*>(obuf + r * RS + c16 * 16);
            __builtin_nontemporal_store(v, reinterpret_cast<v4u_t*>(out + ((size_t)b * Nq + q) * ldo + h * DH + c16 * 8));
        }
    }
}

#undef DSRX
#undef LGKM4
#undef LGKM2

}  // namespace

// bf16 leg of pmhip_attention (attention.hip): arguments already validated there
int pm_attention_bf16(const void* Q, const void* K, const void* Vt, void* out, int ldo, int B, int heads, int Nq, int Nkv,
                      int Nkv_pad, int use_exp2, hipStream_t s) {
    const int nqb = ceil_div(Nq, 4 * QF * 16);
    dim3 grid(nqb * B * heads), block(THREADS);
    if (use_exp2)
        hipLaunchKernelGGL((attention_bf16_kernel<true>), grid, block, 0, s, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)Vt,
                           (bf16_t*)out, ldo, heads, Nq, Nkv, Nkv_pad, nqb);
    else
        hipLaunchKernelGGL((attention_bf16_kernel<false>), grid, block, 0, s, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)Vt,
                           (bf16_t*)out, ldo, heads, Nq, Nkv, Nkv_pad, nqb);
    return PMHIP_OK;
}
