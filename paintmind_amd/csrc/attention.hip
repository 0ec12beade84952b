// Fused softmax(Q K^T) V for dim_head = 64, no mask, no dropout (gfx950): the fp32-verify kernel (bf16: attention_bf16.hip).
// Replaces the reference's materialised-score attention (modules/attention.py:51-58: q@k^T ->
// softmax -> @v, a (B*H, N, N) fp32 tensor per layer) and its xformers alternative (:100).
//
// Work split: one workgroup = 256 queries (bf16; 128 in f32) of one (batch, head); 4 waves x 64 queries, so
// every K / V^T fragment read from LDS feeds 4 MFMAs.  K/V stream through a 2-stage LDS ring in tiles of
// 64 keys: tile t+1 is written (from registers loaded one iteration earlier) while tile t is being
// multiplied, one barrier per tile.  In exp2 mode the running max is only raised (and O rescaled) when it
// grows by more than 2^8 (deferred rescale; P stays <= 256, l and O accumulate in fp32).
//
// MFMA formulation (16x16 tiles; operand chunk = 16 B per lane, see common.h Mma<T>):
//   S^T[key, query]  = K . Q^T     -> each lane holds ONE query column (lane&15) and 4 keys per tile,
//                                     so row max / row sum are in-lane plus two cross-lane steps
//   O^T[d, query]   += V^T . P^T   -> P^T is consumed straight from the S^T accumulators: the lane's
//                                     keys are exactly the k-slice the column operand needs
// For bf16 the K rows of a tile are read in a permuted order so that the 8 keys a lane owns across
// two S^T tiles are CONTIGUOUS, which makes the matching V^T operand one ds_read_b128.
// V arrives pre-transposed ([B,H,64,Nkv_pad], written by the projection GEMM's epilogue).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr int KT = 64;        // keys per tile
constexpr int DH = 64;
constexpr int THREADS = 256;

template <typename T> struct AttnCfg;
// (the bf16 instantiation of this kernel -- the round-2 bf16 path behind PMHIP_ATTN_OLD -- left the library in round 6: bf16 is
// served by attention_bf16.hip; this file is the fp32-verify kernel)
template <> struct AttnCfg<float> {
    static constexpr int ROWB = 256;
    static constexpr int SLOTS = 16;
    static constexpr int NCH = 4;
};

// tile row (key within the 64-key tile) that S^T tile kf presents as its row i (i = 0..15)
template <typename T> __device__ __forceinline__ int key_of_row(int kf, int i);
template <> __device__ __forceinline__ int key_of_row<bf16_t>(int kf, int i) {
    return 32 * (kf >> 1) + 8 * (i >> 2) + 4 * (kf & 1) + (i & 3);
}
template <> __device__ __forceinline__ int key_of_row<float>(int kf, int i) { return 16 * kf + i; }

template <typename T>
__device__ __forceinline__ uint4 lds_chunk(const unsigned char* tile, int row, int slot) {
    using C = AttnCfg<T>;
    return *reinterpret_cast<const uint4*>(tile + row * C::ROWB + ((slot ^ (row & (C::SLOTS - 1))) << 4));
}

// zero the elements of a 16-B V^T chunk whose key index is >= nkv (first key of the chunk = k0)
template <typename T> __device__ __forceinline__ uint4 mask_keys(uint4 v, int k0, int nkv);
template <> __device__ __forceinline__ uint4 mask_keys<bf16_t>(uint4 v, int k0, int nkv) {
    const int n = nkv - k0;                      // number of valid keys in this 8-key chunk (may be <= 0)
    v.x = n <= 0 ? 0u : (n == 1 ? (v.x & 0xffffu) : v.x);
    v.y = n <= 2 ? 0u : (n == 3 ? (v.y & 0xffffu) : v.y);
    v.z = n <= 4 ? 0u : (n == 5 ? (v.z & 0xffffu) : v.z);
    v.w = n <= 6 ? 0u : (n == 7 ? (v.w & 0xffffu) : v.w);
    return v;
}
template <> __device__ __forceinline__ uint4 mask_keys<float>(uint4 v, int k0, int nkv) {
    if (k0 + 0 >= nkv) v.x = 0u;
    if (k0 + 1 >= nkv) v.y = 0u;
    if (k0 + 2 >= nkv) v.z = 0u;
    if (k0 + 3 >= nkv) v.w = 0u;
    return v;
}

template <typename T> __device__ __forceinline__ uint4 pack_p(const f32x4_t& lo, const f32x4_t& hi);
template <> __device__ __forceinline__ uint4 pack_p<bf16_t>(const f32x4_t& lo, const f32x4_t& hi) {
    return make_uint4(pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]),
                      pack_bf16x2(hi[2], hi[3]));
}

// One K tile + one V^T tile (64 rows x ROWB bytes each) go global -> LDS by DMA, 1 KiB per wave-instruction.
// The LDS image is lane-linear, so the bank swizzle (slot ^ row) is applied to the SOURCE address here and
// again in lds_chunk() on the read side.
template <typename T>
__device__ __forceinline__ void stage_tiles(unsigned char* stage, const unsigned char* __restrict__ Kbh,
                                            const unsigned char* __restrict__ Vbh, size_t v_row_bytes, int t, int wave,
                                            int lane) {
    using C = AttnCfg<T>;
    constexpr int RPC = 1024 / C::ROWB;                  // rows per 1 KiB chunk
    constexpr int CHUNKS = KT / RPC;                     // chunks per tile
    constexpr int PER_WAVE = CHUNKS / 4;
    const int kv0 = t * KT;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int chunk = wave * PER_WAVE + i;
        const int row = chunk * RPC + lane / C::SLOTS;
        const int lslot = (lane % C::SLOTS) ^ (row & (C::SLOTS - 1));
        glds16(Kbh + (size_t)(kv0 + row) * C::ROWB + lslot * 16, stage + chunk * 1024);
        glds16(Vbh + (size_t)row * v_row_bytes + (size_t)kv0 * sizeof(T) + lslot * 16, stage + KT * C::ROWB + chunk * 1024);
    }
}

// bf16 form of stage_tiles on `buffer_load_dwordx4 ... lds`: descriptors based at the (batch, head)'s K / V^T, ONE per-lane
// vector offset each (row-in-chunk and swizzled slot do not depend on the chunk: chunks start at multiples of 8 rows), the
// chunk / tile position in the scalar offset -- no VALU address arithmetic per instruction.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ void stage_tiles_bf16(unsigned char* stage, rsrc_t Kr, rsrc_t Vr, unsigned kvoff, unsigned vvoff,
                                                 unsigned v_row_bytes, int t, int wave) {
    constexpr int PER_WAVE = 2;                          // 8 chunks of 8 rows per tile, 4 waves
    const unsigned kv0 = (unsigned)t * KT;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const unsigned chunk = (unsigned)wave * PER_WAVE + i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(Kr, (__attribute__((address_space(3))) void*)(stage + chunk * 1024), 16, kvoff,
                                                 (kv0 + chunk * 8) * 128u, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(Vr, (__attribute__((address_space(3))) void*)(stage + KT * 128 + chunk * 1024), 16, vvoff,
                                                 chunk * 8 * v_row_bytes + kv0 * 2u, 0, 0);
    }
}

// ragged last tile: zero the V^T columns of keys >= Nkv (K needs nothing: those scores are masked to -inf)
template <typename T>
__device__ __forceinline__ void zero_ragged_v(unsigned char* Vl, int kv0, int Nkv, int tid) {
    using C = AttnCfg<T>;
    for (int idx = tid; idx < KT * C::SLOTS; idx += THREADS) {
        const int row = idx / C::SLOTS, lslot = idx % C::SLOTS;
        uint4* p = reinterpret_cast<uint4*>(Vl + row * C::ROWB + ((lslot ^ (row & (C::SLOTS - 1))) << 4));
        *p = mask_keys<T>(*p, kv0 + lslot * (16 / (int)sizeof(T)), Nkv);
    }
}

// QF = 16-query tiles per wave (bf16: 4 -> 64 queries per wave, 256 per workgroup; f32: 2).
template <typename T, bool EXP2, int QF>
__global__ __launch_bounds__(THREADS, 2) void attention_kernel(const T* __restrict__ Q, const T* __restrict__ Kp,
                                                            const T* __restrict__ Vt, T* __restrict__ out,
                                                            int ldo, int heads, int Nq, int Nkv, int Nkv_pad, int nqb) {
    using C = AttnCfg<T>;
    constexpr int TILE_BYTES = KT * C::ROWB;
    constexpr int STAGE_BYTES = 2 * TILE_BYTES;              // K tile + V^T tile
    constexpr float kDefer = EXP2 ? 8.0f : 0.0f;             // skip the O rescale while the row max grows < 2^8
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * STAGE_BYTES];   // 3-stage K / V^T ring

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // 1-D grid.  Workgroup L runs on XCD L % 8 (private 4 MiB L2): give all query blocks of one (batch, head) the
    // same L % 8 so its K / V^T (256 KiB in bf16) are fetched from HBM once and re-read from that XCD's L2.
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

    const T* Qbh = Q + (size_t)bh * Nq * DH;
    const unsigned char* Kbh = reinterpret_cast<const unsigned char*>(Kp + (size_t)bh * Nkv_pad * DH);
    const unsigned char* Vbh = reinterpret_cast<const unsigned char*>(Vt + (size_t)bh * DH * Nkv_pad);
    const size_t v_row_bytes = (size_t)Nkv_pad * sizeof(T);
    // bf16: DMA descriptors / lane offsets, and the per-lane parts of the fragment addresses.  Fragment reads are inline-asm
    // ds_read_b128 with immediate offsets: with C++ LDS reads hipcc puts `s_waitcnt vmcnt(0)` in front of the first read
    // after the DMA issue, i.e. it waits for the NEXT tile's DMA right after issuing it (measured: 200 -> 171 us with the
    // DMA removed).  The explicit vmcnt(0) + barrier in enter_tile() is what orders reads after the DMA that fed them.
    //   K row of S^T tile kf = 2 pc + kk, row i = l15:  32 pc + 8 (l15 >> 2) + 4 kk + (l15 & 3);  slot (4 c + g) ^ (row & 7)
    //     = stage + [8 (l15 >> 2) + (l15 & 3)] * 128 + (g ^ (l15 & 3)) * 16  +  pc * 4096 + kk * 512 + (c ^ kk) * 64
    //   V^T row 16 df + l15, slot (4 pc + g) ^ (l15 & 7)
    //     = stage + 8192 + l15 * 128 + ((4 pc + g) ^ (l15 & 7)) * 16  +  df * 2048
    [[maybe_unused]] rsrc_t Kr, Vr;
    [[maybe_unused]] unsigned kvoff = 0, vvoff = 0, kfrag_lane = 0, vfrag_lane0 = 0, vfrag_lane1 = 0;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    if constexpr (sizeof(T) == 2) {
        Kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Kbh), 0, 0x7fffffff, 0x00020000);
        Vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Vbh), 0, 0x7fffffff, 0x00020000);
        const unsigned lslot = (unsigned)(((lane & 7) ^ ((lane >> 3) & 7)) << 4);
        kvoff = (unsigned)(lane >> 3) * 128u + lslot;
        vvoff = (unsigned)(lane >> 3) * (unsigned)v_row_bytes + lslot;
        kfrag_lane = lds_base + (unsigned)(8 * (l15 >> 2) + (l15 & 3)) * 128u + (unsigned)((g ^ (l15 & 3)) << 4);
        vfrag_lane0 = lds_base + 8192u + (unsigned)l15 * 128u + (unsigned)(((0 + g) ^ (l15 & 7)) << 4);
        vfrag_lane1 = lds_base + 8192u + (unsigned)l15 * 128u + (unsigned)(((4 + g) ^ (l15 & 7)) << 4);
    }
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
#define DSRX(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    // the wait names the fragments as in/out operands, so every MFMA that consumes one is ordered behind it
#define LGKM_N(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f))

    // Q fragments stay in registers for the whole kernel (column operand of S^T)
    uint4 qreg[QF][C::NCH];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        int q = q0 + qf * 16 + l15;
        q = q < Nq ? q : Nq - 1;
        const unsigned char* qrow = reinterpret_cast<const unsigned char*>(Qbh + (size_t)q * DH);
#pragma unroll
        for (int c = 0; c < C::NCH; ++c) qreg[qf][c] = *reinterpret_cast<const uint4*>(qrow + (c * 4 + g) * 16);
    }

    f32x4_t o[4][QF];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < QF; ++j) o[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float mrun[QF], lrun[QF];            // running max; per-lane partial row sums (reduced at the end)
    // The S^T accumulators START from -m (the running max of their query column) instead of 0, so the MFMA delivers
    // s - m and the softmax needs no subtraction per score.  m is the value at the time the QK^T of a half-tile is
    // issued; the (rare) rescale branch below moves already-computed scores to a new max.  Before the first
    // half-tile m is undefined and the accumulators start from 0.
    f32x4_t negm[QF];
#pragma unroll
    for (int j = 0; j < QF; ++j) { mrun[j] = -INFINITY; lrun[j] = 0.f; negm[j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

    const int ntiles = (Nkv + KT - 1) / KT;
    const int nhalves = (Nkv + 31) / 32;                     // 32-key half-tiles that contain at least one valid key
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    constexpr int LOADS_PER_TILE = 2 * ((KT / (1024 / C::ROWB)) / 4);   // DMA instructions per wave per tile (K + V^T)

    // bf16: issue the 4 K-fragment reads of half-tile hh ([kk][c]); they are waited for, one by one, in front of the MFMAs that
    // consume them (qk_half), so whatever the caller puts in between runs under their LDS latency.  Counted lgkmcnt is exact
    // here because LDS operations return in order and the loop has no scalar loads (checked in the ISA: all s_load are in
    // the kernel prologue).
    auto k_issue = [&](v4u_t (&kf)[2][2], int hh) {
        if constexpr (sizeof(T) == 2) {
            const unsigned ka = kfrag_lane + (unsigned)((hh >> 1) % 3) * STAGE_BYTES + (unsigned)(hh & 1) * 4096u;
            DSRX(kf[0][0], ka, 0 * 512 + 0 * 64); DSRX(kf[0][1], ka, 0 * 512 + 1 * 64);
            DSRX(kf[1][0], ka, 1 * 512 + 1 * 64); DSRX(kf[1][1], ka, 1 * 512 + 0 * 64);
        }
    };

    // S^T of one half-tile: 2 key tiles x QF query tiles
    auto qk_half = [&](f32x4_t (&sd)[2][QF], int hh, v4u_t (&kf)[2][2]) {
        const unsigned char* Kl = lds + ((hh >> 1) % 3) * STAGE_BYTES;
        const int pc = hh & 1;
        if constexpr (sizeof(T) == 2) {
            (void)Kl; (void)pc;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int qf = 0; qf < QF; ++qf) sd[kk][qf] = negm[qf];
            LGKM_N(3, kf[0][0]);
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) Mma<T>::run(sd[0][qf], __builtin_bit_cast(uint4, kf[0][0]), qreg[qf][0]);
            LGKM_N(2, kf[0][1]);
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) Mma<T>::run(sd[0][qf], __builtin_bit_cast(uint4, kf[0][1]), qreg[qf][1]);
            LGKM_N(1, kf[1][0]);
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) Mma<T>::run(sd[1][qf], __builtin_bit_cast(uint4, kf[1][0]), qreg[qf][0]);
            LGKM_N(0, kf[1][1]);
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) Mma<T>::run(sd[1][qf], __builtin_bit_cast(uint4, kf[1][1]), qreg[qf][1]);
            return;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) sd[kk][qf] = negm[qf];
            const int krow = key_of_row<T>(2 * pc + kk, l15);
#pragma unroll
            for (int c = 0; c < C::NCH; ++c) {
                const uint4 kfrag = lds_chunk<T>(Kl, krow, c * 4 + g);
#pragma unroll
                for (int qf = 0; qf < QF; ++qf) Mma<T>::run(sd[kk][qf], kfrag, qreg[qf][c]);
            }
        }
    };

    // (1) row max of half-tile hh and the rare rescale branch
    auto rowmax_rescale = [&](f32x4_t (&sc)[2][QF], int hh) {
        const int t = hh >> 1, pc = hh & 1, kv0 = t * KT;
        if (kv0 + KT > Nkv) {                                // ragged last tile: mask keys beyond Nkv
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kv0 + key_of_row<T>(2 * pc + kk, 4 * g + r);
                    if (key >= Nkv) {
#pragma unroll
                        for (int qf = 0; qf < QF; ++qf) sc[kk][qf][r] = -INFINITY;
                    }
                }
        }
        float tmax[QF];                                      // max of (s - mb), mb = what the accumulators started from
        const bool first = (hh == 0);
        bool grow = first;
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            float m = vmax3(sc[0][qf][0], sc[0][qf][1], sc[0][qf][2]);
            m = vmax3(m, sc[0][qf][3], sc[1][qf][0]);
            m = vmax3(m, sc[1][qf][1], sc[1][qf][2]);
            m = vmax2(m, sc[1][qf][3]);                      // this lane's 8 keys only: enough to DETECT growth
            tmax[qf] = m;
            grow |= (m > kDefer);
        }
        if (__any(grow)) {                                   // wave-uniform: rescale everything at the old max exactly once
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) {
                const float mb = first ? 0.f : mrun[qf];
                const float mnew = vmax3(mrun[qf], group4_max(tmax[qf]) + mb, -1e30f);   // column max over the 4 lane groups
                const float alpha = EXP2 ? __builtin_amdgcn_exp2f(mrun[qf] - mnew) : expf(mrun[qf] - mnew);
                const float delta = mb - mnew;               // scores already hold s - mb: move them to s - mnew
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[kk][qf][r] += delta;
                mrun[qf] = mnew;
                negm[qf] = f32x4_t{-mnew, -mnew, -mnew, -mnew};
                lrun[qf] *= alpha;
#pragma unroll
                for (int df = 0; df < 4; ++df) {
                    o[df][qf][0] *= alpha; o[df][qf][1] *= alpha; o[df][qf][2] *= alpha; o[df][qf][3] *= alpha;
                }
            }
        }
    };

    // (2) ONE basic block: the exponentials / row sums / bf16 packing of half-tile hh (VALU + transcendental) and,
    // when there is a next half-tile, the 16 MFMAs of its S^T -- independent work, interleaved by the scheduler
    // directives below so the matrix pipe runs under the softmax instead of after it.
    auto exp_and_next_qk = [&](auto has_next_c, f32x4_t (&sc)[2][QF], uint4 (&pfrag)[2][QF], f32x4_t (&sn)[2][QF], int hn, v4u_t (&kf)[2][2]) {
        constexpr bool has_next = decltype(has_next_c)::value;      // compile-time: the steady-state region has no branch
        if constexpr (has_next) qk_half(sn, hn, kf);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            float psum = 0.f;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = EXP2 ? __builtin_amdgcn_exp2f(sc[kk][qf][r]) : expf(sc[kk][qf][r]);   // sc = s - m
                    sc[kk][qf][r] = pv;
                    psum += pv;
                }
            lrun[qf] += psum;
            if constexpr (sizeof(T) == 2) {
                pfrag[0][qf] = pack_p<bf16_t>(sc[0][qf], sc[1][qf]);
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    pfrag[kk][qf] = make_uint4(__float_as_uint(sc[kk][qf][0]), __float_as_uint(sc[kk][qf][1]),
                                               __float_as_uint(sc[kk][qf][2]), __float_as_uint(sc[kk][qf][3]));
            }
        }
        if constexpr (sizeof(T) == 2 && EXP2 && has_next) {
            {
                // 4 K-fragment reads, then 16 x {1 MFMA, 2 transcendental}; the pack / row-sum VALU depend on exps of later groups
                // and are left to the scheduler (a VALU group inside the pattern makes it infeasible and it is dropped whole)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                }
            }
        }
    };

    // (3) O^T += V^T . P^T for half-tile hh
    auto pv_half = [&](uint4 (&pfrag)[2][QF], int hh) {
        const int t = hh >> 1, pc = hh & 1;
        const unsigned char* Vl = lds + (t % 3) * STAGE_BYTES + TILE_BYTES;
        if constexpr (sizeof(T) == 2) {
            (void)Vl;
            const unsigned va = (pc ? vfrag_lane1 : vfrag_lane0) + (unsigned)(t % 3) * STAGE_BYTES;
            v4u_t vf[4];
            DSRX(vf[0], va, 0 * 2048); DSRX(vf[1], va, 1 * 2048); DSRX(vf[2], va, 2 * 2048); DSRX(vf[3], va, 3 * 2048);
            LGKM_N(3, vf[0]);
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) Mma<T>::run(o[0][qf], __builtin_bit_cast(uint4, vf[0]), pfrag[0][qf]);
            LGKM_N(2, vf[1]);
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) Mma<T>::run(o[1][qf], __builtin_bit_cast(uint4, vf[1]), pfrag[0][qf]);
            LGKM_N(1, vf[2]);
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) Mma<T>::run(o[2][qf], __builtin_bit_cast(uint4, vf[2]), pfrag[0][qf]);
            LGKM_N(0, vf[3]);
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) Mma<T>::run(o[3][qf], __builtin_bit_cast(uint4, vf[3]), pfrag[0][qf]);
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int df = 0; df < 4; ++df) {
                    const uint4 vfrag = lds_chunk<T>(Vl, df * 16 + l15, (2 * pc + kk) * 4 + g);
#pragma unroll
                    for (int qf = 0; qf < QF; ++qf) Mma<T>::run(o[df][qf], vfrag, pfrag[kk][qf]);
                }
        }
    };

    // entering tile tn (called while the previous tile's second half is still to be consumed): its DMA has landed
    // and is published by the barrier; the barrier also proves every wave is done with tile tn-2, whose stage the
    // DMA of tile tn+1 now reuses (3-stage ring)
    auto enter_tile = [&](int tn) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tn + 1 < ntiles) {
            if constexpr (sizeof(T) == 2) stage_tiles_bf16(lds + ((tn + 1) % 3) * STAGE_BYTES, Kr, Vr, kvoff, vvoff, (unsigned)v_row_bytes, tn + 1, wave_u);
            else stage_tiles<T>(lds + ((tn + 1) % 3) * STAGE_BYTES, Kbh, Vbh, v_row_bytes, tn + 1, wave_u, lane);
        }
        if (tn * KT + KT > Nkv) {
            zero_ragged_v<T>(lds + (tn % 3) * STAGE_BYTES + TILE_BYTES, tn * KT, Nkv, tid);
            __syncthreads();
        }
    };

    if constexpr (sizeof(T) == 2) stage_tiles_bf16(lds, Kr, Vr, kvoff, vvoff, (unsigned)v_row_bytes, 0, wave_u);
    else stage_tiles<T>(lds, Kbh, Vbh, v_row_bytes, 0, wave_u, lane);
    (void)LOADS_PER_TILE;
    enter_tile(0);

    // Software pipeline over half-tiles: the MFMAs of S^T(h+1) are independent of the softmax VALU work on S^T(h),
    // so the two interleave inside one wave; two named S buffers alternate (static register indexing).
    f32x4_t sA[2][QF], sB[2][QF];
    uint4 pfrag[2][QF];
    constexpr std::true_type kNext{};
    constexpr std::false_type kLast{};
    v4u_t kf[2][2];
    k_issue(kf, 0);
    qk_half(sA, 0, kf);
    int hs = 0;
    for (; hs + 2 < nhalves; hs += 2) {                                  // steady state: both following halves exist
        k_issue(kf, hs + 1);                                            // same tile: the row max runs under the reads
        rowmax_rescale(sA, hs);
        exp_and_next_qk(kNext, sA, pfrag, sB, hs + 1, kf);              // S^T(hs+1): same tile, second half
        pv_half(pfrag, hs);
        enter_tile((hs + 2) >> 1);                                      // S^T(hs+2) opens the next tile
        k_issue(kf, hs + 2);
        rowmax_rescale(sB, hs + 1);
        exp_and_next_qk(kNext, sB, pfrag, sA, hs + 2, kf);
        pv_half(pfrag, hs + 1);
    }
    if (hs + 1 < nhalves) {                                             // tail: one or two halves left
        k_issue(kf, hs + 1);
        rowmax_rescale(sA, hs);
        exp_and_next_qk(kNext, sA, pfrag, sB, hs + 1, kf);
        pv_half(pfrag, hs);
        rowmax_rescale(sB, hs + 1);
        exp_and_next_qk(kLast, sB, pfrag, sA, 0, kf);
        pv_half(pfrag, hs + 1);
    } else {
        rowmax_rescale(sA, hs);
        exp_and_next_qk(kLast, sA, pfrag, sB, 0, kf);
        pv_half(pfrag, hs);
    }

    // ---- finalize: O = O^T / l, head-major inside the output row
    float inv[QF];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {      // cross-lane steps first, outside any divergent region
        inv[qf] = 1.0f / group4_sum(lrun[qf]);
    }
    if constexpr (sizeof(T) == 2) {
        // bf16: the wave's 64 output rows go through the (now idle) K / V^T ring, so that every global store instruction
        // writes 8 whole 128-byte rows (non-temporal) instead of 16 x 4 pieces of 8 bytes at a row stride
        constexpr int RS = 144;                                // staged row: 64 bf16 + pad, 16-B aligned, conflict-free
        __syncthreads();                                       // every wave is done reading the ring
        unsigned char* obuf = lds + wave * (64 * RS);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf)
#pragma unroll
            for (int df = 0; df < 4; ++df)
                *reinterpret_cast<uint2*>(obuf + (qf * 16 + l15) * RS + (df * 16 + g * 4) * 2) =
                    make_uint2(pack_bf16x2(o[df][qf][0] * inv[qf], o[df][qf][1] * inv[qf]), pack_bf16x2(o[df][qf][2] * inv[qf], o[df][qf][3] * inv[qf]));
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < QF * 2; ++it) {                  // 8 rows x 128 B per store instruction
            const int r = it * 8 + (lane >> 3), c16 = lane & 7, q = q0 + r;
            if (q < Nq) {
                typedef unsigned nt_v4u __attribute__((ext_vector_type(4)));
                const uint4 v = *reinterpret_cast<const uint4*>(obuf + r * RS + c16 * 16);
                __builtin_nontemporal_store(nt_v4u{v.x, v.y, v.z, v.w},
                                            reinterpret_cast<nt_v4u*>(out + ((size_t)b * Nq + q) * ldo + h * DH + c16 * 8));
            }
        }
    } else {
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            const int q = q0 + qf * 16 + l15;
            if (q < Nq) {
                T* orow = out + ((size_t)b * Nq + q) * ldo + h * DH;
#pragma unroll
                for (int df = 0; df < 4; ++df)
                    store4(orow + df * 16 + g * 4, o[df][qf][0] * inv[qf], o[df][qf][1] * inv[qf],
                           o[df][qf][2] * inv[qf], o[df][qf][3] * inv[qf]);
            }
        }
    }
}

#undef DSRX
#undef LGKM_N

}  // namespace

int pm_attention_bf16(const void* Q, const void* K, const void* Vt, void* out, int ldo, int B, int heads, int Nq, int Nkv,
                      int Nkv_pad, int use_exp2, hipStream_t s);

extern "C" int pmhip_attention(int dtype, const void* Q, const void* K, const void* Vt, void* out, int ldo,
                               int B, int heads, int Nq, int Nkv, int Nkv_pad, int use_exp2,
                               pmhip_stream stream) {
    PM_REQUIRE(dtype == PMHIP_F32 || dtype == PMHIP_BF16, "attention: bad dtype %d", dtype);
    PM_REQUIRE(Q && K && Vt && out, "attention: null pointer");
    PM_REQUIRE(B > 0 && heads > 0 && Nq > 0 && Nkv > 0, "attention: empty problem");
    PM_REQUIRE(Nkv_pad % KT == 0 && Nkv_pad >= Nkv, "attention: Nkv_pad=%d must be a multiple of 64 >= Nkv=%d", Nkv_pad, Nkv);
    PM_REQUIRE(ldo % 4 == 0 && (dtype == PMHIP_F32 || ldo % 8 == 0), "attention: ldo must be a multiple of 4 (f32) / 8 (bf16: 16-byte row stores)");
    hipStream_t s = (hipStream_t)stream;
    dim3 block(THREADS);
    PmTimer tm(FAM_ATTENTION, s);
    if (dtype == PMHIP_F32) {
        const int nqb = ceil_div(Nq, 4 * 2 * 16);
        dim3 grid(nqb * B * heads);
        if (use_exp2)
            hipLaunchKernelGGL((attention_kernel<float, true, 2>), grid, block, 0, s, (const float*)Q, (const float*)K,
                               (const float*)Vt, (float*)out, ldo, heads, Nq, Nkv, Nkv_pad, nqb);
        else
            hipLaunchKernelGGL((attention_kernel<float, false, 2>), grid, block, 0, s, (const float*)Q, (const float*)K,
                               (const float*)Vt, (float*)out, ldo, heads, Nq, Nkv, Nkv_pad, nqb);
    } else {
        PM_TRY(pm_attention_bf16(Q, K, Vt, out, ldo, B, heads, Nq, Nkv, Nkv_pad, use_exp2, s));
    }
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}
