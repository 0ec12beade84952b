// Fused softmax(Q K^T) V for dim_head = 64, no mask, no dropout (gfx950).
// Replaces the reference's materialised-score attention (modules/attention.py:51-58: q@k^T ->
// softmax -> @v, a (B*H, N, N) fp32 tensor per layer) and its xformers alternative (:100).
//
// Work split: one workgroup = 128 queries of one (batch, head); 4 waves x 32 queries.  K/V stream
// through LDS in tiles of 64 keys, prefetched into registers while the previous tile is being
// multiplied (issue-early / write-late staging).
//
// MFMA formulation (16x16 tiles; operand chunk = 16 B per lane, see common.h Mma<T>):
//   S^T[key, query]  = K . Q^T     -> each lane holds ONE query column (lane&15) and 4 keys per tile,
//                                     so row max / row sum are in-lane plus two cross-lane steps
//   O^T[d, query]   += V^T . P^T   -> P^T is consumed straight from the S^T accumulators: the lane's
//                                     keys are exactly the k-slice the column operand needs
// For bf16 the K rows of a tile are read in a permuted order so that the 8 keys a lane owns across
// two S^T tiles are CONTIGUOUS, which makes the matching V^T operand one ds_read_b128.
// V arrives pre-transposed ([B,H,64,Nkv_pad], written by the projection GEMM's epilogue).
#include "common.h"

namespace {

constexpr int QB = 128;       // queries per workgroup
constexpr int KT = 64;        // keys per tile
constexpr int DH = 64;
constexpr int THREADS = 256;

template <typename T> struct AttnCfg;
template <> struct AttnCfg<bf16_t> {
    static constexpr int ROWB = 128;     // bytes per K row (64 d) == bytes per V^T row segment (64 keys)
    static constexpr int SLOTS = 8;
    static constexpr int NCH = 2;        // 16-B chunks x4 lane groups per 64-wide contraction
    static constexpr int PASSES = 2;     // 256 threads x 16 B per pass to stage one 8 KiB tile
};
template <> struct AttnCfg<float> {
    static constexpr int ROWB = 256;
    static constexpr int SLOTS = 16;
    static constexpr int NCH = 4;
    static constexpr int PASSES = 4;
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
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (k0 + 2 * i >= nkv) w[i] = 0u;
        else if (k0 + 2 * i + 1 >= nkv) w[i] &= 0xffffu;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
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

template <typename T, bool EXP2>
__global__ __launch_bounds__(THREADS) void attention_kernel(const T* __restrict__ Q, const T* __restrict__ Kp,
                                                            const T* __restrict__ Vt, T* __restrict__ out,
                                                            int ldo, int heads, int Nq, int Nkv, int Nkv_pad) {
    using C = AttnCfg<T>;
    constexpr int TILE_BYTES = KT * C::ROWB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TILE_BYTES];
    unsigned char* Kl = lds;
    unsigned char* Vl = lds + TILE_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y;
    const int b = bh / heads, h = bh % heads;
    const int q0 = blockIdx.x * QB + wave * 32;

    const T* Qbh = Q + (size_t)bh * Nq * DH;
    const unsigned char* Kbh = reinterpret_cast<const unsigned char*>(Kp + (size_t)bh * Nkv_pad * DH);
    const unsigned char* Vbh = reinterpret_cast<const unsigned char*>(Vt + (size_t)bh * DH * Nkv_pad);
    const size_t v_row_bytes = (size_t)Nkv_pad * sizeof(T);

    // Q fragments stay in registers for the whole kernel (column operand of S^T)
    uint4 qreg[2][C::NCH];
#pragma unroll
    for (int qf = 0; qf < 2; ++qf) {
        int q = q0 + qf * 16 + l15;
        q = q < Nq ? q : Nq - 1;
        const unsigned char* qrow = reinterpret_cast<const unsigned char*>(Qbh + (size_t)q * DH);
#pragma unroll
        for (int c = 0; c < C::NCH; ++c) qreg[qf][c] = *reinterpret_cast<const uint4*>(qrow + (c * 4 + g) * 16);
    }

    f32x4_t o[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) o[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float mrun[2] = {-INFINITY, -INFINITY};
    float lrun[2] = {0.f, 0.f};           // per-lane partial row sums (reduced over the 4 lane groups at the end)

    const int ntiles = (Nkv + KT - 1) / KT;
    uint4 kst[C::PASSES], vst[C::PASSES];

    auto issue_loads = [&](int t) {
        const int kv0 = t * KT;
#pragma unroll
        for (int ps = 0; ps < C::PASSES; ++ps) {
            const int idx = ps * THREADS + tid;
            const int row = idx / C::SLOTS, slot = idx % C::SLOTS;
            kst[ps] = *reinterpret_cast<const uint4*>(Kbh + (size_t)(kv0 + row) * C::ROWB + slot * 16);
            vst[ps] = *reinterpret_cast<const uint4*>(Vbh + (size_t)row * v_row_bytes + (size_t)kv0 * sizeof(T) + slot * 16);
        }
    };
    auto write_lds = [&](int t) {
        const int kv0 = t * KT;
        const bool ragged = kv0 + KT > Nkv;
#pragma unroll
        for (int ps = 0; ps < C::PASSES; ++ps) {
            const int idx = ps * THREADS + tid;
            const int row = idx / C::SLOTS, slot = idx % C::SLOTS;
            const int off = row * C::ROWB + ((slot ^ (row & (C::SLOTS - 1))) << 4);
            *reinterpret_cast<uint4*>(Kl + off) = kst[ps];
            uint4 v = vst[ps];
            if (ragged) v = mask_keys<T>(v, kv0 + slot * (16 / (int)sizeof(T)), Nkv);
            *reinterpret_cast<uint4*>(Vl + off) = v;
        }
    };

    issue_loads(0);
    write_lds(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) issue_loads(t + 1);
        const int kv0 = t * KT;

        // ---- S^T = K . Q^T : 4 key tiles x 2 query tiles
        f32x4_t s[4][2];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            s[kf][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            s[kf][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const int krow = key_of_row<T>(kf, l15);
#pragma unroll
            for (int c = 0; c < C::NCH; ++c) {
                const uint4 kfrag = lds_chunk<T>(Kl, krow, c * 4 + g);
                Mma<T>::run(s[kf][0], kfrag, qreg[0][c]);
                Mma<T>::run(s[kf][1], kfrag, qreg[1][c]);
            }
        }
        // mask keys beyond Nkv (only the last tile can be ragged)
        if (kv0 + KT > Nkv) {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kv0 + key_of_row<T>(kf, 4 * g + r);
                    if (key >= Nkv) { s[kf][0][r] = -INFINITY; s[kf][1][r] = -INFINITY; }
                }
        }
        // ---- online softmax per query column
#pragma unroll
        for (int qf = 0; qf < 2; ++qf) {
            float tmax = s[0][qf][0];
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, s[kf][qf][r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float mnew = fmaxf(mrun[qf], tmax);      // finite: every tile has >= 1 valid key
            const float alpha = EXP2 ? __builtin_amdgcn_exp2f(mrun[qf] - mnew) : expf(mrun[qf] - mnew);
            mrun[qf] = mnew;
            float psum = 0.f;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = EXP2 ? __builtin_amdgcn_exp2f(s[kf][qf][r] - mnew) : expf(s[kf][qf][r] - mnew);
                    s[kf][qf][r] = pv;
                    psum += pv;
                }
            lrun[qf] = lrun[qf] * alpha + psum;
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                o[df][qf][0] *= alpha; o[df][qf][1] *= alpha; o[df][qf][2] *= alpha; o[df][qf][3] *= alpha;
            }
        }
        // ---- O^T += V^T . P^T
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const uint4 p0 = pack_p<bf16_t>(s[2 * pc][0], s[2 * pc + 1][0]);
                const uint4 p1 = pack_p<bf16_t>(s[2 * pc][1], s[2 * pc + 1][1]);
#pragma unroll
                for (int df = 0; df < 4; ++df) {
                    const uint4 vfrag = lds_chunk<T>(Vl, df * 16 + l15, pc * 4 + g);
                    Mma<T>::run(o[df][0], vfrag, p0);
                    Mma<T>::run(o[df][1], vfrag, p1);
                }
            }
        } else {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
                const uint4 p0 = make_uint4(__float_as_uint(s[kf][0][0]), __float_as_uint(s[kf][0][1]),
                                            __float_as_uint(s[kf][0][2]), __float_as_uint(s[kf][0][3]));
                const uint4 p1 = make_uint4(__float_as_uint(s[kf][1][0]), __float_as_uint(s[kf][1][1]),
                                            __float_as_uint(s[kf][1][2]), __float_as_uint(s[kf][1][3]));
#pragma unroll
                for (int df = 0; df < 4; ++df) {
                    const uint4 vfrag = lds_chunk<T>(Vl, df * 16 + l15, kf * 4 + g);
                    Mma<T>::run(o[df][0], vfrag, p0);
                    Mma<T>::run(o[df][1], vfrag, p1);
                }
            }
        }
        __syncthreads();                      // everyone is done reading this tile
        if (t + 1 < ntiles) {
            write_lds(t + 1);
            __syncthreads();
        }
    }

    // ---- finalize: O = O^T / l, head-major inside the output row
    float inv[2];
#pragma unroll
    for (int qf = 0; qf < 2; ++qf) {      // cross-lane steps first, outside any divergent region
        float l = lrun[qf];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        inv[qf] = 1.0f / l;
    }
#pragma unroll
    for (int qf = 0; qf < 2; ++qf) {
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

}  // namespace

extern "C" int pmhip_attention(int dtype, const void* Q, const void* K, const void* Vt, void* out, int ldo,
                               int B, int heads, int Nq, int Nkv, int Nkv_pad, int use_exp2,
                               pmhip_stream stream) {
    PM_REQUIRE(dtype == PMHIP_F32 || dtype == PMHIP_BF16, "attention: bad dtype %d", dtype);
    PM_REQUIRE(Q && K && Vt && out, "attention: null pointer");
    PM_REQUIRE(B > 0 && heads > 0 && Nq > 0 && Nkv > 0, "attention: empty problem");
    PM_REQUIRE(Nkv_pad % KT == 0 && Nkv_pad >= Nkv, "attention: Nkv_pad=%d must be a multiple of 64 >= Nkv=%d", Nkv_pad, Nkv);
    PM_REQUIRE(ldo % 4 == 0, "attention: ldo must be a multiple of 4");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(ceil_div(Nq, QB), B * heads), block(THREADS);
    PmTimer tm(FAM_ATTENTION, s);
    if (dtype == PMHIP_F32) {
        if (use_exp2)
            hipLaunchKernelGGL((attention_kernel<float, true>), grid, block, 0, s, (const float*)Q, (const float*)K,
                               (const float*)Vt, (float*)out, ldo, heads, Nq, Nkv, Nkv_pad);
        else
            hipLaunchKernelGGL((attention_kernel<float, false>), grid, block, 0, s, (const float*)Q, (const float*)K,
                               (const float*)Vt, (float*)out, ldo, heads, Nq, Nkv, Nkv_pad);
    } else {
        if (use_exp2)
            hipLaunchKernelGGL((attention_kernel<bf16_t, true>), grid, block, 0, s, (const bf16_t*)Q, (const bf16_t*)K,
                               (const bf16_t*)Vt, (bf16_t*)out, ldo, heads, Nq, Nkv, Nkv_pad);
        else
            hipLaunchKernelGGL((attention_kernel<bf16_t, false>), grid, block, 0, s, (const bf16_t*)Q, (const bf16_t*)K,
                               (const bf16_t*)Vt, (bf16_t*)out, ldo, heads, Nq, Nkv, Nkv_pad);
    }
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}
