// Shared pieces of the NT GEMM family (gemm.hip: 128x128 tile; gemm256.hip: 256x256 phase-staggered tile):
// parameter block, tile-walk helpers and the wave-level epilogue.
#pragma once
#include "common.h"

namespace pmgemm {

constexpr int ROWB = 128;                          // bytes of k per tile row per step (64 bf16 / 32 f32)
constexpr int ESTRIDE = 68;                        // floats per row of the per-wave epilogue buffer (64 + pad)
constexpr int EPI_WAVE_BYTES = 8192;               // idle LDS each wave needs for its epilogue

enum { EPI_STD = 0, EPI_SWIGLU = 1, EPI_HEADS = 2 };

struct GemmParams {
    const void* A; const void* W;
    const float* bias; const float* residual;
    void* out;
    int lda, ldw, ldr, res_rows, ldo;
    int M, N, K;
    // EPI_HEADS
    int heads, tokens, tokens_pad, inner;
    int kinds[3];
    void* outs[3];
    float q_scale;
    int fast_math;                                 // SwiGLU: 1 = fast exp (bf16 mode)
    int chunk;                                     // n-tiles per L2 chunk of the tile walk (256x256 kernel)
    // LayerNorm fold, producer side (EPI_STD, f32 out): also write the row as bf16 and, per 64-column chunk,
    // (sum x, sum x^2) of the row -> stats_out[N/64][m][2] (chunk-major)
    bf16_t* xb_out; int ldxb;
    float* stats_out;
    // LayerNorm fold, consumer side (256x256 kernel): A is the RAW bf16 row, W carries gamma, and the epilogue applies
    // out = rstd * acc - rstd * mean * c[n] + d[n] with (mean, rstd) from the producer's partial sums
    const float* ln_stats; int ln_nc;              // [ln_nc][M][2], ln_nc = K / 64
    const float* ln_coef;                          // [M][2]: (rstd, -rstd * mean), reduced from ln_stats by pm_ln_finalize
    const float* ln_c; const float* ln_d;          // [N]: c = sum_k bf16(gamma_k W_nk), d = sum_k beta_k W_nk
    float ln_eps;
};

// XCD-aware, bijective block remap: consecutive virtual ids stay on one XCD's L2 (block b runs on XCD b % 8).
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int xcd = bid & 7, local = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// Tile walk: n-tiles are visited in chunks of at most `max_chunk` tiles with m-tiles varying inside a chunk, so
// that an XCD's resident blocks share one W chunk (<= 1 MiB in bf16) + a few A row panels inside its 4 MiB L2.
__device__ __forceinline__ void tile_of_block(int vb, int tiles_m, int tiles_n, int max_chunk, int& tm, int& tn) {
    const int nchunks = (tiles_n + max_chunk - 1) / max_chunk;
    const int cw = (tiles_n + nchunks - 1) / nchunks;              // n-tiles per chunk (the last may be narrower)
    const int chunk = vb / (tiles_m * cw);
    const int cw_here = min(cw, tiles_n - chunk * cw);
    const int rem = vb - chunk * tiles_m * cw;
    tm = rem / cw_here;
    tn = chunk * cw + rem % cw_here;
}

__device__ __forceinline__ uint4 read_frag(const unsigned char* lds_tile, int row, int slot) {
    return *reinterpret_cast<const uint4*>(lds_tile + row * ROWB + ((slot ^ (row & 7)) << 4));
}

// silu(x1) * x2.  fast (bf16 mode): exp2-based exponential and a hardware reciprocal (the result is rounded to
// bf16 anyway); exact (f32 verify mode): accurate expf and an IEEE divide, like torch's CPU silu.
__device__ __forceinline__ float silu_mul_fast(float x1, float x2) {
    return x1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x1 * -1.4426950408889634f)) * x2;
}
__device__ __forceinline__ float silu_mul(float x1, float x2, int fast) {
    if (fast) return silu_mul_fast(x1, x2);
    return (x1 / (1.0f + expf(-x1))) * x2;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm fold, consumer side.  ln_row_coeffs: (rstd, -rstd * mean) of the MI rows this lane owns in the
// accumulator layout (row mwave + mi*16 + l15), from the producer's per-64-column partial sums.  Deterministic: the
// partials are summed in chunk order.  ln_apply: acc = rstd * acc - rstd * mean * c[n] + d[n].
// ------------------------------------------------------------------------------------------------
// Statistics layout: stats[chunk][row][2] (chunk-major), so that for one chunk the 128 rows of a wave are 1 KiB of
// contiguous memory: lane j fetches rows 2j and 2j+1 of every chunk with ONE coalesced 16-byte load per chunk, all
// issued before the first is consumed (one memory round trip).  The (rstd, -rstd*mean) pairs then go through `scratch`
// (LDS private to the wave) so that each lane picks up the 8 rows it owns in the accumulator layout; the wave's 64
// entries of c and d ride along into scratch[256 .. 383] so the epilogue reads them from LDS.
// The loads are inline asm: hipcc would otherwise wait vmcnt(0) -- draining the LDS-DMA of the first K-tile that is in
// flight at the same time -- before their first use.  ln_stats_issue goes BEFORE the DMA pieces are issued,
// ln_row_coeffs after them with `dma_in_flight` = the number of DMA instructions issued in between (counted wait).
// Round 2, second version: the consumer reads the per-row pair (rstd, -rstd * mean) that pm_ln_finalize reduced from the
// partial sums -- 8 eight-byte loads per lane plus one for c | d, issued (inline asm, invisible to hipcc's waits) in the
// first read slot of the tile's LAST K-tile and retired by that K-tile's closing vmcnt(0), so the fold costs the persistent,
// streamed K loop nothing but 20 registers in its last K-tile.
struct LnCoef { float2 ab[8]; f32x4_t cd; };
__device__ __forceinline__ void ln_coef_issue(const GemmParams& p, int mwave, int nw, int lane, LnCoef& L) {
    const float* a = p.ln_coef + ((size_t)mwave + (lane & 15)) * 2;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(L.ab[mi]) : "v"(a + mi * 32) : "memory");
    const float* cda = (lane < 16 ? p.ln_c : p.ln_d) + nw + (lane & 15) * 4;      // lanes >= 32 re-read lanes 0-31's addresses
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(L.cd) : "v"(cda) : "memory");
}
constexpr int LN_COEF_LOADS = 9;
// after the loads have been waited for: c | d go through the wave's LDS scratch (ln_apply reads them from there)
__device__ __forceinline__ void ln_coef_finish(int lane, float* scratch, LnCoef& L, float (&fa)[8], float (&fb)[8]) {
    if (lane < 32) *reinterpret_cast<f32x4_t*>(scratch + 256 + lane * 4) = L.cd;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) { fa[mi] = L.ab[mi].x; fb[mi] = L.ab[mi].y; }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

constexpr int LN_MAXC = 16;
struct LnLoads { f32x4_t t[LN_MAXC]; f32x4_t cd; };

__device__ __forceinline__ void ln_stats_issue(const GemmParams& p, int mwave, int nw, int lane, LnLoads& L) {
    const float* st = p.ln_stats + ((size_t)mwave + 2 * lane) * 2;
    const size_t cstride = (size_t)p.M * 2;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        if (c < p.ln_nc) {                                          // wave-uniform (scalar branch): no per-lane predication
            const float* a = st + c * cstride;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(L.t[c]) : "v"(a) : "memory");
        }
    }
    const float* cda = (lane < 16 ? p.ln_c : p.ln_d) + nw + (lane & 15) * 4;      // lanes >= 32 re-read lanes 0-31's addresses
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(L.cd) : "v"(cda) : "memory");
}

template <int DMA_IN_FLIGHT>
__device__ __forceinline__ void ln_row_coeffs(const GemmParams& p, int lane, float* scratch, LnLoads& L, float (&fa)[8], float (&fb)[8]) {
    const int l15 = lane & 15;
    // every statistics load is older than the DMA instructions: leave exactly those in flight
    asm volatile("s_waitcnt vmcnt(%17)"
                 : "+v"(L.t[0]), "+v"(L.t[1]), "+v"(L.t[2]), "+v"(L.t[3]), "+v"(L.t[4]), "+v"(L.t[5]), "+v"(L.t[6]), "+v"(L.t[7]),
                   "+v"(L.t[8]), "+v"(L.t[9]), "+v"(L.t[10]), "+v"(L.t[11]), "+v"(L.t[12]), "+v"(L.t[13]), "+v"(L.t[14]), "+v"(L.t[15]),
                   "+v"(L.cd)
                 : "n"(DMA_IN_FLIGHT) : "memory");
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c)
        if (c < p.ln_nc) { s[0] += L.t[c][0]; s[1] += L.t[c][1]; s[2] += L.t[c][2]; s[3] += L.t[c][3]; }      // chunk order: deterministic
    const float invD = 1.0f / (float)(p.ln_nc * 64);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float mean = s[2 * r] * invD;
        const float var = fmaxf(s[2 * r + 1] * invD - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + p.ln_eps);
        *reinterpret_cast<float2*>(scratch + (2 * lane + r) * 2) = make_float2(rstd, -rstd * mean);
    }
    if (lane < 32) *reinterpret_cast<f32x4_t*>(scratch + 256 + lane * 4) = L.cd;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const float2 ab = *reinterpret_cast<const float2*>(scratch + (mi * 16 + l15) * 2);
        fa[mi] = ab.x;
        fb[mi] = ab.y;
    }
}

template <int MI>
__device__ __forceinline__ void ln_apply(const float* scratch, f32x4_t (&acc)[MI][4], int lane, const float (&fa)[MI],
                                         const float (&fb)[MI]) {
    const int g = lane >> 4;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {                               // c | d of one 16-column group at a time: 8 live registers, not 32
        const float4 cc = *reinterpret_cast<const float4*>(scratch + 256 + ni * 16 + g * 4);
        const float4 dd = *reinterpret_cast<const float4*>(scratch + 256 + 64 + ni * 16 + g * 4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            acc[mi][ni][0] = fa[mi] * acc[mi][ni][0] + (fb[mi] * cc.x + dd.x);
            acc[mi][ni][1] = fa[mi] * acc[mi][ni][1] + (fb[mi] * cc.y + dd.y);
            acc[mi][ni][2] = fa[mi] * acc[mi][ni][2] + (fb[mi] * cc.z + dd.z);
            acc[mi][ni][3] = fa[mi] * acc[mi][ni][3] + (fb[mi] * cc.w + dd.w);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Wave-level epilogue for a (MI*16) x 64 accumulator tile held as acc[MI][4] (MFMA issued with W as the row
// operand: a lane owns row l15 of a 16-row group and 4 consecutive columns per 16x16 tile).  The tile is
// transposed 16 rows at a time through `eraw` (>= EPI_WAVE_BYTES of LDS nobody else touches) so that every
// global access covers 128-256 contiguous bytes per row.  `rpre` (used when NPRE > 1) holds the residual tile
// prefetched in the store layout: element [mi*ITERS + it].  FULL: the tile is known to lie inside the matrix (no
// predicates, so the slices are straight-line code and the compiler can count its vmcnt waits instead of draining
// the store queue at every branch join).  RES: 1 / 0 = residual known present / absent at compile time (-1: runtime);
// with FULL && RES == 1 the residual rows of slice mi+1 are requested before slice mi's stores are issued.
// ------------------------------------------------------------------------------------------------
// `hook` runs once, after the epilogue's own up-front loads (bias) have been waited for and before its first store: the
// persistent 256x256 kernel issues the NEXT tile's first K-tile DMA there, so that no compiler-inserted wait for a load of
// this epilogue can drain that DMA.

// The epilogue's global stores are NON-TEMPORAL (`global_store_dwordx4 ... nt`): outputs are 64-200 MB per launch, far more
// than the L2 holds, and are next read by a different kernel; measured on the whole bench (same box, tools/ab_same_box.sh):
// GEMM family 98.7 -> 92.1 ms per step, head-split QKV 103 -> 98.5 us, SwiGLU 203 -> 190 us.
typedef unsigned nt_v4u __attribute__((ext_vector_type(4)));
typedef unsigned nt_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nt_store16(void* p, uint4 v) {
    __builtin_nontemporal_store(nt_v4u{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_v4u*>(p));
}
__device__ __forceinline__ void nt_store_row(float* p, const float (&v)[4]) { nt_store16(p, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]))); }
__device__ __forceinline__ void nt_store_row(bf16_t* p, const float (&v)[8]) {
    nt_store16(p, make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])));
}
struct NoHook { __device__ __forceinline__ void operator()() const {} };

template <int EPI, typename OutT, int MI, int NPRE, bool FULL = false, int RES = -1, bool EMIT = false, typename Hook = NoHook>
__device__ __forceinline__ void wave_epilogue(const GemmParams& p, f32x4_t (&acc)[MI][4], unsigned char* eraw, int mwave,
                                              int nw, int lane, const float4 (&rpre)[NPRE], Hook hook = Hook()) {
    constexpr int CPL = 16 / (int)sizeof(OutT);                    // columns per lane per store (16 B)
    constexpr int LPR = 64 / CPL, RPI = 64 / LPR, ITERS = 16 / RPI;
    const int l15 = lane & 15, g = lane >> 4;
    const int ccol = (lane % LPR) * CPL, ncol = nw + ccol;
    float* ebuf = reinterpret_cast<float*>(eraw);
    if (!FULL && nw >= p.N) return;

    if constexpr (EPI == EPI_HEADS) {
        // V part: 64-token x 64-d blocks are transposed through LDS as OutT so that V^T[b,h,d,:] rows are written
        // 128 B (bf16) at a time; Q and K take the generic 16-row path below.
        if (p.kinds[nw / p.inner] == PMHIP_PART_V) {
            hook();
            const int h = (nw % p.inner) >> 6;
            OutT* dst = reinterpret_cast<OutT*>(p.outs[nw / p.inner]);
#pragma unroll
            for (int blk = 0; blk < MI / 4; ++blk) {
                const int mblk = mwave + blk * 64;
                const int b0 = mblk / p.tokens, t0 = mblk % p.tokens;
                const bool whole = (FULL || mblk + 63 < p.M) && (t0 + 63 < p.tokens) && (t0 % 8 == 0);
                if (whole && sizeof(OutT) == 2) {
                    OutT* vbuf = reinterpret_cast<OutT*>(eraw);                  // [64 d][64 tokens]
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                vbuf[(ni * 16 + g * 4 + r) * 64 + mi * 16 + l15] = from_f32<OutT>(acc[blk * 4 + mi][ni][r]);
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    OutT* vrow = dst + (((size_t)b0 * p.heads + h) * 64) * p.tokens_pad + t0;
#pragma unroll
                    for (int it = 0; it < 8; ++it) {                             // 8 d-rows x 128 B per store instruction
                        const int d = it * 8 + (lane >> 3), c = (lane & 7) * 8;
                        nt_store16(vrow + (size_t)d * p.tokens_pad + c, *reinterpret_cast<const uint4*>(vbuf + d * 64 + c));
                    }
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                } else {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        const int mm = mblk + mi * 16 + l15;
                        if (mm < p.M) {
                            const int b = mm / p.tokens, t = mm % p.tokens;
#pragma unroll
                            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    dst[(((size_t)b * p.heads + h) * 64 + ni * 16 + g * 4 + r) * p.tokens_pad + t] =
                                        from_f32<OutT>(acc[blk * 4 + mi][ni][r]);
                        }
                    }
                }
            }
            return;
        }
    }

    float bias_v[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) bias_v[j] = 0.f;
    if constexpr (EPI == EPI_STD) {
        if (p.bias && (FULL || ncol < p.N)) {
#pragma unroll
            for (int j = 0; j < CPL; j += 4) {
                const float4 bb = *reinterpret_cast<const float4*>(p.bias + ncol + j);
                bias_v[j] = bb.x; bias_v[j + 1] = bb.y; bias_v[j + 2] = bb.z; bias_v[j + 3] = bb.w;
            }
        }
    }
    // SwiGLU.  The wave's 64 columns are [x1 0-15 | x2 0-15 | x1 16-31 | x2 16-31] of 32 hidden columns, so in the
    // ACCUMULATOR layout lane (l15, g) already holds x1 and x2 of the same hidden columns (tiles 0|1 and 2|3).
    //   bf16 out: gate in registers, then transpose 32 bf16 columns per row (a quarter of the fp32 traffic);
    //   f32 out (verify mode): generic fp32 transposition first, gate on the transposed rows.
    constexpr bool GATE_IN_REGS = (EPI == EPI_SWIGLU) && sizeof(OutT) == 2;
    float4 sb[4] = {};                                             // this lane's bias slices
    if constexpr (GATE_IN_REGS) {
        const float* b1 = p.bias + nw + g * 4;                     // x1 lo | x2 lo | x1 hi | x2 hi, columns g*4 .. g*4+3
        sb[0] = *reinterpret_cast<const float4*>(b1);      sb[1] = *reinterpret_cast<const float4*>(b1 + 16);
        sb[2] = *reinterpret_cast<const float4*>(b1 + 32); sb[3] = *reinterpret_cast<const float4*>(b1 + 48);
    } else if constexpr (EPI == EPI_SWIGLU) {
        const int q = lane & 3;
        const float* b1 = p.bias + nw + (q >> 1) * 32 + (q & 1) * 8;
        sb[0] = *reinterpret_cast<const float4*>(b1);      sb[1] = *reinterpret_cast<const float4*>(b1 + 4);
        sb[2] = *reinterpret_cast<const float4*>(b1 + 16); sb[3] = *reinterpret_cast<const float4*>(b1 + 20);
    }
    // Retire the bias loads HERE, on every path, with a wait the compiler can see.  Otherwise each predicated
    // store block below gets its own `s_waitcnt vmcnt(0)` for the bias registers (the skipped-path state never
    // clears), and because vmcnt counts stores too, every 16-row slice then waits for the previous slice's stores
    // to be acknowledged by L2 instead of streaming them.
    __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0); expcnt / lgkmcnt untouched
    hook();
    // The two bf16 epilogues that carry most of the GEMM time (SwiGLU gate, Q / K head split) run their 16-row slices as a
    // two-stage pipeline over two staging buffers: slice mi+1 is staged while slice mi's rows are read back and stored, one
    // LDS round trip and one wait per slice instead of two (ablation: of 98 us for the QKV GEMM 17 us were this chain, of
    // 201 us for SwiGLU 31 us chain + gate arithmetic).
    if constexpr (GATE_IN_REGS && FULL) {
        constexpr int RS = 80, BUFB = 2048;                        // bytes per staged row: 32 bf16 + pad; buffer stride
        auto stage = [&](int mi) {
            unsigned char* row = eraw + (mi & 1) * BUFB + l15 * RS;
            *reinterpret_cast<uint2*>(row + g * 8) = make_uint2(
                pack_bf16x2(silu_mul_fast(acc[mi][0][0] + sb[0].x, acc[mi][1][0] + sb[1].x),
                            silu_mul_fast(acc[mi][0][1] + sb[0].y, acc[mi][1][1] + sb[1].y)),
                pack_bf16x2(silu_mul_fast(acc[mi][0][2] + sb[0].z, acc[mi][1][2] + sb[1].z),
                            silu_mul_fast(acc[mi][0][3] + sb[0].w, acc[mi][1][3] + sb[1].w)));
            *reinterpret_cast<uint2*>(row + 32 + g * 8) = make_uint2(
                pack_bf16x2(silu_mul_fast(acc[mi][2][0] + sb[2].x, acc[mi][3][0] + sb[3].x),
                            silu_mul_fast(acc[mi][2][1] + sb[2].y, acc[mi][3][1] + sb[3].y)),
                pack_bf16x2(silu_mul_fast(acc[mi][2][2] + sb[2].z, acc[mi][3][2] + sb[3].z),
                            silu_mul_fast(acc[mi][2][3] + sb[2].w, acc[mi][3][3] + sb[3].w)));
        };
        stage(0);
        const int erow = lane >> 2, q = lane & 3;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // slice mi staged; the reads of slice mi-1 (other buffer) returned
            const uint4 v = *reinterpret_cast<const uint4*>(eraw + (mi & 1) * BUFB + erow * RS + q * 16);
            if (mi + 1 < MI) stage(mi + 1);
            nt_store16(reinterpret_cast<OutT*>(p.out) + (size_t)(mwave + mi * 16 + erow) * p.ldo + (nw >> 1) + q * 8, v);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return;
    }
    if constexpr (EPI == EPI_HEADS && sizeof(OutT) == 2 && FULL) {     // Q / K part (the V part returned above)
        constexpr int RS = 144, BUFB = 4096;                       // bytes per staged row: 64 bf16 + pad; buffer stride
        const int part = nw / p.inner;
        const int h = (nw % p.inner) >> 6;
        const int kind = p.kinds[part];
        OutT* dst = reinterpret_cast<OutT*>(p.outs[part]);
        const int tstride = kind == PMHIP_PART_Q ? p.tokens : p.tokens_pad;
        const float sc = kind == PMHIP_PART_Q ? p.q_scale : 1.0f;
        auto stage = [&](int mi) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                *reinterpret_cast<uint2*>(eraw + (mi & 1) * BUFB + l15 * RS + ni * 32 + g * 8) =
                    make_uint2(pack_bf16x2(acc[mi][ni][0] * sc, acc[mi][ni][1] * sc), pack_bf16x2(acc[mi][ni][2] * sc, acc[mi][ni][3] * sc));
        };
        stage(0);
        // (batch, token) of the wave's first row once per tile, wave-uniform: a per-row `mm / tokens`, `mm % tokens` is two
        // ~40-instruction integer divisions per row and store and was most of this epilogue's non-store time
        const int b0 = mwave / p.tokens, t0 = mwave % p.tokens;
        const bool nowrap = t0 + MI * 16 <= p.tokens;              // the wave's rows stay inside one image (tokens % 128 == 0: always)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            uint4 v[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) v[it] = *reinterpret_cast<const uint4*>(eraw + (mi & 1) * BUFB + (it * 8 + (lane >> 3)) * RS + (lane & 7) * 16);
            if (mi + 1 < MI) stage(mi + 1);
#pragma unroll
            for (int it = 0; it < 2; ++it) {                       // 8 rows x 128 B per store instruction
                const int r = mi * 16 + it * 8 + (lane >> 3);
                int b = b0, t = t0 + r;
                if (!nowrap) { b = (mwave + r) / p.tokens; t = (mwave + r) % p.tokens; }
                nt_store16(dst + (((size_t)b * p.heads + h) * tstride + t) * 64 + (lane & 7) * 8, v[it]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return;
    }
    constexpr bool PIPE_RES = (EPI == EPI_STD) && FULL && RES == 1 && NPRE == 1 && sizeof(OutT) == 4;
    float4 rnext[ITERS] = {};
    if constexpr (PIPE_RES) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it)
            rnext[it] = *reinterpret_cast<const float4*>(p.residual + (size_t)((mwave + it * RPI + lane / LPR) % p.res_rows) * p.ldr + ncol);
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        float4 rcur[ITERS];
        if constexpr (PIPE_RES) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) rcur[it] = rnext[it];
            if (mi + 1 < MI) {
#pragma unroll
                for (int it = 0; it < ITERS; ++it)
                    rnext[it] = *reinterpret_cast<const float4*>(
                        p.residual + (size_t)((mwave + (mi + 1) * 16 + it * RPI + lane / LPR) % p.res_rows) * p.ldr + ncol);
            }
        }
        const int mbase = mwave + mi * 16;
        if constexpr (GATE_IN_REGS) {
            constexpr int RS = 80;                                 // bytes per staged row: 32 bf16 + pad, 16-B aligned
            unsigned char* row = eraw + l15 * RS;
            *reinterpret_cast<uint2*>(row + g * 8) = make_uint2(
                pack_bf16x2(silu_mul(acc[mi][0][0] + sb[0].x, acc[mi][1][0] + sb[1].x, p.fast_math),
                            silu_mul(acc[mi][0][1] + sb[0].y, acc[mi][1][1] + sb[1].y, p.fast_math)),
                pack_bf16x2(silu_mul(acc[mi][0][2] + sb[0].z, acc[mi][1][2] + sb[1].z, p.fast_math),
                            silu_mul(acc[mi][0][3] + sb[0].w, acc[mi][1][3] + sb[1].w, p.fast_math)));
            *reinterpret_cast<uint2*>(row + 32 + g * 8) = make_uint2(
                pack_bf16x2(silu_mul(acc[mi][2][0] + sb[2].x, acc[mi][3][0] + sb[3].x, p.fast_math),
                            silu_mul(acc[mi][2][1] + sb[2].y, acc[mi][3][1] + sb[3].y, p.fast_math)),
                pack_bf16x2(silu_mul(acc[mi][2][2] + sb[2].z, acc[mi][3][2] + sb[3].z, p.fast_math),
                            silu_mul(acc[mi][2][3] + sb[2].w, acc[mi][3][3] + sb[3].w, p.fast_math)));
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int erow = lane >> 2, q = lane & 3, m = mbase + erow;
            if (FULL || m < p.M)                                   // 4 lanes x 16 B = the row's 32 hidden columns
                nt_store16(reinterpret_cast<OutT*>(p.out) + (size_t)m * p.ldo + (nw >> 1) + q * 8, *reinterpret_cast<const uint4*>(eraw + erow * RS + q * 16));
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            continue;
        }
        if constexpr (EPI == EPI_HEADS && sizeof(OutT) == 2) {
            // Q / K part, bf16: scale and round in registers, transpose 64 bf16 columns per row (half the fp32 traffic)
            constexpr int RS = 144;                                // bytes per staged row: 64 bf16 + pad, 16-B aligned
            const int part = nw / p.inner;
            const int h = (nw % p.inner) >> 6;
            const int kind = p.kinds[part];
            OutT* dst = reinterpret_cast<OutT*>(p.outs[part]);
            const int tstride = kind == PMHIP_PART_Q ? p.tokens : p.tokens_pad;
            const float sc = kind == PMHIP_PART_Q ? p.q_scale : 1.0f;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                *reinterpret_cast<uint2*>(eraw + l15 * RS + ni * 32 + g * 8) =
                    make_uint2(pack_bf16x2(acc[mi][ni][0] * sc, acc[mi][ni][1] * sc), pack_bf16x2(acc[mi][ni][2] * sc, acc[mi][ni][3] * sc));
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 2; ++it) {                       // 8 rows x 128 B per store instruction
                const int r = it * 8 + (lane >> 3), mm = mbase + r, c16 = lane & 7;
                if (FULL || mm < p.M) {
                    const int b = mm / p.tokens, t = mm % p.tokens;
                    nt_store16(dst + (((size_t)b * p.heads + h) * tstride + t) * 64 + c16 * 8, *reinterpret_cast<const uint4*>(eraw + r * RS + c16 * 16));
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            continue;
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
            *reinterpret_cast<f32x4_t*>(ebuf + l15 * ESTRIDE + ni * 16 + g * 4) = acc[mi][ni];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (EPI == EPI_STD) {
            // one store instruction = RPI rows x 64 columns, LPR adjacent lanes per row (full 128-B lines)
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int r = it * RPI + lane / LPR;
                const int mm = mbase + r;
                if (FULL || (mm < p.M && ncol < p.N)) {
                    float v[CPL];
#pragma unroll
                    for (int j = 0; j < CPL; j += 4) {
                        const float4 t = *reinterpret_cast<const float4*>(ebuf + r * ESTRIDE + ccol + j);
                        v[j] = t.x + bias_v[j]; v[j + 1] = t.y + bias_v[j + 1]; v[j + 2] = t.z + bias_v[j + 2]; v[j + 3] = t.w + bias_v[j + 3];
                    }
                    if constexpr (sizeof(OutT) == 4 && RES != 0) {
                        if (RES == 1 || p.residual) {
                            float4 rr;
                            if constexpr (PIPE_RES) rr = rcur[it];
                            else if constexpr (NPRE > 1) rr = rpre[mi * ITERS + it]; // compile-time index: stays in registers
                            else rr = *reinterpret_cast<const float4*>(p.residual + (size_t)(mm % p.res_rows) * p.ldr + ncol);
                            v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                        }
                    }
                    nt_store_row(reinterpret_cast<OutT*>(p.out) + (size_t)mm * p.ldo + ncol, v);
                    if constexpr (EMIT && sizeof(OutT) == 4) {
                        // LayerNorm fold: the consumer GEMM reads this row as bf16; its statistics come from the f32 values
                        __builtin_nontemporal_store(nt_v2u{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}, reinterpret_cast<nt_v2u*>(p.xb_out + (size_t)mm * p.ldxb + ncol));
                        float s1 = (v[0] + v[1]) + (v[2] + v[3]);
                        float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                        s1 += dpp_mov<0xB1>(s1); s2 += dpp_mov<0xB1>(s2);            // the row's 64 columns sit in 16 adjacent lanes
                        s1 += dpp_mov<0x4E>(s1); s2 += dpp_mov<0x4E>(s2);
                        s1 += dpp_mov<0x141>(s1); s2 += dpp_mov<0x141>(s2);
                        s1 += dpp_mov<0x140>(s1); s2 += dpp_mov<0x140>(s2);
                        if ((lane & 15) == 0)
                            *reinterpret_cast<float2*>(p.stats_out + ((size_t)(nw >> 6) * p.M + mm) * 2) = make_float2(s1, s2);     // [chunk][row][2]
                    }
                }
            }
        } else if constexpr (EPI == EPI_SWIGLU) {
            // wave columns: [x1 0-15 | x2 0-15 | x1 16-31 | x2 16-31] of 32 hidden columns; lane (erow, q) gates
            // hidden columns q*8 .. q*8+7
            const int erow = lane >> 2, m = mbase + erow;
            if (FULL || m < p.M) {
                const int q = lane & 3;
                const int c1 = (q >> 1) * 32 + (q & 1) * 8;          // x1 column inside the wave tile
                const float* e1 = ebuf + erow * ESTRIDE + c1;
                const float4 a0 = *reinterpret_cast<const float4*>(e1), a1 = *reinterpret_cast<const float4*>(e1 + 4);
                const float4 g0 = *reinterpret_cast<const float4*>(e1 + 16), g1 = *reinterpret_cast<const float4*>(e1 + 20);
                const float4 ba0 = sb[0], ba1 = sb[1], bg0 = sb[2], bg1 = sb[3];
                float h[8];
                h[0] = silu_mul(a0.x + ba0.x, g0.x + bg0.x, p.fast_math); h[1] = silu_mul(a0.y + ba0.y, g0.y + bg0.y, p.fast_math);
                h[2] = silu_mul(a0.z + ba0.z, g0.z + bg0.z, p.fast_math); h[3] = silu_mul(a0.w + ba0.w, g0.w + bg0.w, p.fast_math);
                h[4] = silu_mul(a1.x + ba1.x, g1.x + bg1.x, p.fast_math); h[5] = silu_mul(a1.y + ba1.y, g1.y + bg1.y, p.fast_math);
                h[6] = silu_mul(a1.z + ba1.z, g1.z + bg1.z, p.fast_math); h[7] = silu_mul(a1.w + ba1.w, g1.w + bg1.w, p.fast_math);
                OutT* out = reinterpret_cast<OutT*>(p.out) + (size_t)m * p.ldo + (nw >> 1) + q * 8;
                if constexpr (sizeof(OutT) == 2) {
                    store_row(out, h);                               // one 16-byte store
                } else {
                    store4(out, h[0], h[1], h[2], h[3]);
                    store4(out + 4, h[4], h[5], h[6], h[7]);
                }
            }
        } else {  // EPI_HEADS, Q or K part: the wave's 64 columns are exactly one head
            const int part = nw / p.inner;
            const int h = (nw % p.inner) >> 6;
            const int kind = p.kinds[part];
            OutT* dst = reinterpret_cast<OutT*>(p.outs[part]);
            const int tstride = kind == PMHIP_PART_Q ? p.tokens : p.tokens_pad;
            const float sc = kind == PMHIP_PART_Q ? p.q_scale : 1.0f;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int r = it * RPI + lane / LPR;
                const int mm = mbase + r;
                if (FULL || mm < p.M) {
                    const int b = mm / p.tokens, t = mm % p.tokens;
                    float v[CPL];
#pragma unroll
                    for (int j = 0; j < CPL; j += 4) {
                        const float4 t4 = *reinterpret_cast<const float4*>(ebuf + r * ESTRIDE + ccol + j);
                        v[j] = t4.x * sc; v[j + 1] = t4.y * sc; v[j + 2] = t4.z * sc; v[j + 3] = t4.w * sc;
                    }
                    nt_store_row(dst + (((size_t)b * p.heads + h) * tstride + t) * 64 + ccol, v);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next 16 rows overwrite ebuf
    }
}

}  // namespace pmgemm
