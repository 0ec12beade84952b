// EXPERIMENT (round 2), not part of libpaintmind_hip.so: an attention kernel on v_mfma_f32_32x32x16_bf16 with 32 queries
// per wave, hidden LDS-DMA two tiles ahead, lane-constant fragment addresses and sum-based max-growth detection.
// Correct (float64 check 3.5e-4, the attention parity tests pass when it is linked in and dispatched) but NOT faster
// than csrc/attention.hip on the bench shape (B=64, H=8, N=1024): 713-743 vs 718-725 TFLOP/s.  Ablations of this
// kernel on one box (tools/attn_only.py, PMHIP_ATTN_ABL): full 185 us; without barrier / vmcnt waits 226 us; exp2 replaced
// by a multiply 178 us; LDS fragment reads replaced by registers 156 us; both 140 us (981 TFLOP/s) -- i.e. the bare
// MFMA + pack + row-sum skeleton already runs at 39 % of the matrix peak, so neither the exponentials, nor the LDS
// reads, nor the synchronisation is what holds attention at ~30 %; see DESIGN.md section 4.
// To try it: copy next to csrc/attention.hip, add `attention32` to build.sh's unit list and call
// pm_attention32_supported / pm_attention32_launch at the top of pmhip_attention.
// Fused softmax(Q K^T) V, bf16, dim_head 64, exp2 domain, for the self-attention shapes of the decode loop
// (Nq a multiple of 256, Nkv a multiple of 64): the 8-wave, 32x32x16-MFMA member of the attention family.
// Same maths and the same deferred-rescale rule as attention.hip (reference modules/attention.py:51-58); ragged /
// f32 / short-context problems stay on attention.hip.
//
// Work split: one workgroup = 512 threads = 8 waves x 32 queries = 256 queries of one (batch, head); two waves per
// SIMD.  K / V^T tiles of 64 keys (8 KiB each) stream through a 4-stage LDS ring by LDS-DMA that the compiler does
// not see (inline asm, counted s_waitcnt vmcnt): one tile in flight across the one barrier per tile.
//
// MFMA formulation (v_mfma_f32_32x32x16_bf16; operand lane l supplies row / column l&31 and k-slots 8*(l>>5)..+7;
// result lane l holds column l&31 and rows (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15):
//   S^T[key, q] = K . Q^T     A = K rows, B = Q^T: a lane holds ONE query column and 16 keys per 32-key block, so the
//                             row max and row sum are in-lane plus one half-wave swap
//   O^T[d, q]  += V^T . P^T   A = V^T rows, B = P^T taken straight from the S^T accumulators
// K rows are fed to the matrix core in a permuted order (bits 2 and 3 of the row index swapped), which makes the 8
// keys a lane owns in registers 8j..8j+7 of a block CONTIGUOUS (32b + 16j + 8h .. +7): the matching V^T operand is one
// 16-byte LDS read and P needs no cross-lane shuffle, only v_cvt_pk_bf16_f32.
// The S^T accumulators start from -m (running max of the column) so the MFMA delivers s - m; the max is only
// raised (and O, l rescaled) when it grows by more than 2^8 -- detected from in-lane maxima, resolved in a rare
// wave-uniform branch.
#include <stdlib.h>

#include <type_traits>

#include "../../paintmind_amd/csrc/common.h"

#ifndef PM_ATTN32_WAVES
#define PM_ATTN32_WAVES 4
#endif

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int KT = 64, DH = 64, NWAVES = PM_ATTN32_WAVES, THREADS = 64 * NWAVES, QW = 32, QWG = QW * NWAVES;
constexpr int CPW = 8 / NWAVES;                 // 1-KiB DMA chunks per wave per operand tile
constexpr int TILE_BYTES = KT * 128;            // one K tile == one V^T tile: 64 rows x 128 B
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int NSTAGE = 4;
constexpr float kDefer = 8.0f;

__device__ __forceinline__ f32x16_t mma32(const uint4& a, const uint4& b, const f32x16_t& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// LDS-DMA the compiler does not count: 16 B per lane, LDS destination = wave-uniform byte address + lane * 16
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

template <int ABL>
__global__ __launch_bounds__(THREADS, 2) void attention32_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kp,
                                                              const bf16_t* __restrict__ Vt, bf16_t* __restrict__ out,
                                                              int ldo, int heads, int Nq, int Nkv, int Nkv_pad, int nqb) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NSTAGE * STAGE_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    // same XCD-aware block order as attention.hip: all query blocks of one (batch, head) on one XCD's L2
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
    const int q0 = qblk * QWG + wave * QW;
    const unsigned char* Kbh = reinterpret_cast<const unsigned char*>(Kp + (size_t)bh * Nkv_pad * DH);
    const unsigned char* Vbh = reinterpret_cast<const unsigned char*>(Vt + (size_t)bh * DH * Nkv_pad);
    const size_t v_row_bytes = (size_t)Nkv_pad * 2;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;

    // Q^T column operand: query q0 + l31, d = 16 s + 8 hh .. + 7
    uint4 qreg[4];
    {
        const unsigned char* qrow = reinterpret_cast<const unsigned char*>(Q + ((size_t)bh * Nq + q0 + l31) * DH);
#pragma unroll
        for (int s = 0; s < 4; ++s) qreg[s] = *reinterpret_cast<const uint4*>(qrow + (2 * s + hh) * 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // before any hidden DMA is issued

    // DMA geometry: a tile is 8 chunks of 8 rows x 128 B; wave w copies chunks w*CPW .. of K and of V^T
    const int drow0 = wave * CPW * 8 + (lane >> 3);
    auto stage = [&](int t) {
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int drow = drow0 + c * 8;
            const int dslot = (lane & 7) ^ ((drow >> 1) & 7);  // bank swizzle on the SOURCE (LDS image is lane-linear)
            const unsigned dst = lds_base + (unsigned)((t & (NSTAGE - 1)) * STAGE_BYTES + (wave * CPW + c) * 1024);
            dma16(Kbh + (size_t)(t * KT + drow) * 128 + dslot * 16, dst);
            dma16(Vbh + (size_t)drow * v_row_bytes + (size_t)t * KT * 2 + dslot * 16, dst + TILE_BYTES);
        }
    };
    // Fragment addresses.  Swizzle: the 16-B chunk c of tile row r sits at chunk c ^ ((r >> 1) & 7): a 256-B bank row
    // holds two tile rows, so the 16 distinct rows one ds_read_b128 lane group touches land on 16 distinct slots.
    // Everything lane-dependent is computed ONCE (4 K offsets, 4 V^T offsets); stage / block offsets are compile-time
    // constants that fold into the ds_read immediates (the tile loop is unrolled over the 4 ring stages).
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) u32x4_t* lds_u4;
    auto ld16 = [](lds_u4 p) -> uint4 { const u32x4_t v = *p; return make_uint4(v[0], v[1], v[2], v[3]); };
    const auto lds3 = (__attribute__((address_space(3))) unsigned char*)lds;
    const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    unsigned ka[4], va[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ka[i] = (unsigned)(krow * 128) + (((unsigned)(2 * i + hh) ^ (unsigned)((krow >> 1) & 7)) << 4);       // d = 16 i + 8 hh
        va[i] = (unsigned)(TILE_BYTES + l31 * 128) + (((unsigned)(2 * i + hh) ^ (unsigned)((l31 >> 1) & 7)) << 4);   // keys 16 i + 8 hh
    }
#define KFRAG(STG, BLK, S) (ABL & 4) ? qreg[S] : ld16((lds_u4)(lds3 + ((STG) * STAGE_BYTES + (BLK) * 4096) + ka[S]))
#define VFRAG(STG, DB, C) (ABL & 4) ? qreg[C] : ld16((lds_u4)(lds3 + ((STG) * STAGE_BYTES + (DB) * 4096) + va[C]))

    f32x16_t o[2];
    f32x16_t negm;                        // all 16 registers = -m of this lane's query column: the C operand of the first QK^T step
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; }
    float mrun = -INFINITY, lrun = 0.f;

    const int ntiles = Nkv / KT;

    // S^T of the tile in ring stage STG: 2 key blocks x 4 k-steps
    auto qk = [&](auto stg_c, f32x16_t (&s)[2]) {
        constexpr int STG = decltype(stg_c)::value;
        // consecutive MFMAs alternate between the two accumulators: a dependent MFMA that is not issued back to back
        // loses the accumulator forwarding path (MI355X_MICROARCH: +43 cycles), an independent one in between hides it
        s[0] = mma32(KFRAG(STG, 0, 0), qreg[0], negm);
        s[1] = mma32(KFRAG(STG, 1, 0), qreg[0], negm);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            s[0] = mma32(KFRAG(STG, 0, k), qreg[k], s[0]);
            s[1] = mma32(KFRAG(STG, 1, k), qreg[k], s[1]);
        }
    };

    // move the column to a new running max: `s` (and `s2`, the already issued S^T of the next tile) hold scores relative
    // to the old max `mb`; everything accumulated at the old max is scaled exactly once
    auto raise_max = [&](f32x16_t (&s)[2], f32x16_t* s2, bool first) {
        float m0 = vmax3(s[0][0], s[0][1], s[0][2]), m1 = vmax3(s[1][0], s[1][1], s[1][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) { m0 = vmax3(m0, s[0][r], s[0][r + 1]); m1 = vmax3(m1, s[1][r], s[1][r + 1]); }
        const float m = vmax3(m0, m1, vmax2(s[0][15], s[1][15]));
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        const float mcol = vmax2(__uint_as_float(sw[0]), __uint_as_float(sw[1]));       // both halves of the column
        const float mb = first ? 0.f : mrun;
        const float mnew = vmax3(mrun, mcol + mb, -1e30f);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        const float delta = mb - mnew;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[0][r] += delta; s[1][r] += delta; o[0][r] *= alpha; o[1][r] *= alpha; negm[r] = -mnew; }
        if (s2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s2[0][r] += delta; s2[1][r] += delta; }
        }
        mrun = mnew;
        lrun *= alpha;
    };

    // tile tn's DMA has landed for this wave; after the barrier for every wave, and every wave is done with tile tn - 2,
    // whose stage the DMA of tile tn + 2 reuses
    auto enter = [&](int tn) {
        if constexpr (!(ABL & 1)) {
        if (tn + 1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::: "memory");                         // no LDS read of the new tile may move above the barrier
        if (tn + 2 < ntiles) stage(tn + 2);
    };

    uint4 pf[2][2];
    // exponentials, lane-partial row sum and bf16 packing of one tile.  Returns the lane's partial sum (NOT yet added to lrun).
    auto exp_pack = [&](f32x16_t (&s)[2]) -> float {
        float psum = 0.f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = (ABL & 2) ? s[blk][r] * 0.001f : __builtin_amdgcn_exp2f(s[blk][r]);
                s[blk][r] = p;
                psum += p;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
                pf[blk][j] = make_uint4(pack_bf16x2(s[blk][8 * j + 0], s[blk][8 * j + 1]), pack_bf16x2(s[blk][8 * j + 2], s[blk][8 * j + 3]),
                                        pack_bf16x2(s[blk][8 * j + 4], s[blk][8 * j + 5]), pack_bf16x2(s[blk][8 * j + 6], s[blk][8 * j + 7]));
        }
        return psum;
    };

    // O^T += V^T . P^T; V^T operand chunk: keys 32 blk + 16 j + 8 hh .. + 7  ->  chunk 4 blk + 2 j + hh  ->  va[2 blk + j]
    auto pv = [&](auto stg_c) {
        constexpr int STG = decltype(stg_c)::value;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int db = 0; db < 2; ++db) o[db] = mma32(VFRAG(STG, db, 2 * blk + j), pf[blk][j], o[db]);
    };

    // One tile.  After the first tile the running max is NOT tracked score by score: the exponentials are taken
    // against the current max and growth is detected from the lane's partial row sum (a score more than 2^10 above the
    // max makes the sum exceed kGrow; an overflow makes it inf).  Only then -- rare, wave-uniform -- the tile's S^T is
    // recomputed from the K tile still in LDS, the max is raised and the exponentials are redone.
    constexpr float kGrow = 1024.0f;
    auto step = [&](auto stg_c, auto has_next_c, f32x16_t (&sc)[2], f32x16_t (&sn)[2], int t) {
        constexpr int STG = decltype(stg_c)::value;
        constexpr bool has_next = decltype(has_next_c)::value;
        if (t == 0) raise_max(sc, nullptr, true);              // first tile: the max is unknown, establish it
        if constexpr (has_next) {
            enter(t + 1);
            qk(std::integral_constant<int, (STG + 1) & 3>{}, sn);       // 8 MFMAs independent of the softmax below
        }
        float psum = exp_pack(sc);
        if (__any(!(psum <= kGrow))) {                         // also catches inf / nan
            qk(stg_c, sc);                                     // scores relative to the max they were first computed against
            raise_max(sc, has_next ? sn : nullptr, false);
            psum = exp_pack(sc);
        }
        lrun += psum;
        pv(stg_c);
    };

    stage(0);
    if (ntiles > 1) stage(1);
    if (ntiles > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ntiles > 2) stage(2);                                  // enter(tn) keeps issuing tile tn + 2 from tn = 1 on

    f32x16_t sA[2], sB[2];
    constexpr std::true_type kNext{};
    constexpr std::false_type kLast{};
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    constexpr std::integral_constant<int, 2> S2{};
    constexpr std::integral_constant<int, 3> S3{};
    qk(S0, sA);
    int t = 0;
    for (; t + 4 < ntiles; t += 4) {                           // ntiles is a multiple of 4: the ring stage of a tile is static
        step(S0, kNext, sA, sB, t);
        step(S1, kNext, sB, sA, t + 1);
        step(S2, kNext, sA, sB, t + 2);
        step(S3, kNext, sB, sA, t + 3);
    }
    step(S0, kNext, sA, sB, t);
    step(S1, kNext, sB, sA, t + 1);
    step(S2, kNext, sA, sB, t + 2);
    step(S3, kLast, sB, sA, t + 3);
#undef KFRAG
#undef VFRAG

    // ---- finalize: O = O^T / l, head-major inside the output row
    {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lrun), __float_as_uint(lrun), false, false);
        const float inv = 1.0f / (__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
        bf16_t* orow = out + ((size_t)b * Nq + q0 + l31) * ldo + h * DH;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                store4(orow + db * 32 + rr * 8 + hh * 4, o[db][4 * rr] * inv, o[db][4 * rr + 1] * inv, o[db][4 * rr + 2] * inv,
                       o[db][4 * rr + 3] * inv);
    }
}

}  // namespace

int pm_attention32_supported(int dtype, int Nq, int Nkv, int Nkv_pad, int use_exp2) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("PMHIP_ATTN32"); on = e ? atoi(e) : 1; }
    return on && dtype == PMHIP_BF16 && use_exp2 && Nq % QWG == 0 && Nkv % (4 * KT) == 0 && Nkv_pad >= Nkv && Nkv >= 4 * KT;
}

int pm_attention32_launch(const void* Q, const void* K, const void* Vt, void* out, int ldo, int B, int heads, int Nq, int Nkv,
                          int Nkv_pad, hipStream_t s) {
    const int nqb = Nq / QWG;
    PmTimer tm(FAM_ATTENTION, s);
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("PMHIP_ATTN_ABL"); abl = e ? atoi(e) : 0; }
#define LAUNCH(A) hipLaunchKernelGGL(attention32_kernel<A>, dim3(nqb * B * heads), dim3(THREADS), 0, s, (const bf16_t*)Q, (const bf16_t*)K, \
                       (const bf16_t*)Vt, (bf16_t*)out, ldo, heads, Nq, Nkv, Nkv_pad, nqb)
    switch (abl) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; case 4: LAUNCH(4); break; case 5: LAUNCH(5); break;
                   case 6: LAUNCH(6); break; case 7: LAUNCH(7); break; default: LAUNCH(0); }
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}
