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
    // bf16 hi/lo residual stream (EPI_STD, bf16 out, RES == 2): x = hi + lo, two bf16 planes.  `residual` then points at
    // the hi plane (bf16, row stride ldr), res_lo at the lo plane; `out` is the new hi plane, out_lo the new lo plane (row
    // stride ldo).  The hi plane alone is what the next GEMM consumes (LayerNorm folded into it), so the stream costs the
    // same 4 + 4 bytes per element as fp32 but needs no separate normalisation pass.
    const bf16_t* res_lo; bf16_t* out_lo;
    // ... and, optionally, the row statistics of the NEW hi plane, so that the LayerNorm folded into the next GEMM needs no pass
    // over the plane (pmhip_ln_coef reads 2 bytes per element: 14 us per launch at the bench shape, 4.4 % of GPU time):
    // row_stats[m][N/64][2] = (sum, centred sum of squares) of the 64 columns a wave owns; pmhip_ln_coef_parts combines them
    // (Chan's update, fixed order).  Needs N % 64 == 0.
    float* row_stats;
    // CENTRED hi plane (round 4).  The folded LayerNorm normalises bf16(x): a row whose common offset is large against its spread
    // loses 2^-9 |x| / std per element (tests/test_gpu_ops.py::test_layernorm_fold_accuracy_...: 7x the unfolded kernel's error
    // at offset 50 std).  LayerNorm does not see a per-row shift, so the producer stores the pair of x - c with c = the row mean
    // of the PREVIOUS hi plane, taken from the (rstd, -rstd * mean) pair the LayerNorm in front of this branch left in
    // `center_coef` ([M][2], pmhip_ln_coef / pmhip_ln_coef_parts): the new hi plane is centred up to the branch's own mean and
    // rounds like LN(x) does.  `shift` ([M], optional): running sum of the subtracted values, kept only where the absolute x is
    // needed again (the ViT encoder's prev_quant); shift_mode 1 = this producer opens the stream (shift <- 0), 2 = shift += c.
    // center_extra: a launch-wide constant added to c -- the mean of this producer's bias over its columns, the part of the new
    // row mean that is known before the row is computed.
    const float* center_coef; float center_extra; float* shift; int shift_mode;
    // LayerNorm fold, consumer side (256x256 kernel): A is the RAW bf16 row (the hi plane), W carries gamma, and the epilogue
    // applies out = rstd * acc - rstd * mean * c[n] + d[n]
    const float* ln_coef;                          // [M][2]: (rstd, -rstd * mean) per row, from pmhip_ln_coef
    const float* ln_c; const float* ln_d;          // [N]: c = sum_k bf16(gamma_k W_nk), d = sum_k beta_k W_nk
    // small-batch form (128x128 kernel only): the coefficients are computed in the prologue from the producer's partial statistics
    // (ln_coef_row, common.h) instead of being read; the workgroups of the first tile column also store them to ln_coef_out for
    // whoever needs them next (the centred producer that follows)
    const float* ln_parts; int ln_nparts; float ln_eps; float* ln_coef_out;
    // f32 EPI_STD output without a residual (the logits GEMM): softmax statistics (max, sum of exp) of every (row, 64-column block),
    // [M][N/64][2], computed from the values as they are stored (common.h, softmax_block_stat).  The sampling kernel then reads these
    // 8 bytes per block and the top-k blocks of a row instead of the whole row (sample.hip).  N % 64 == 0.
    float* block_stats;
};

// timing family of a launch (common.h)
inline int gemm_family(const GemmParams& p, int epi, bool two_wg = false) {
    if (epi == EPI_HEADS) return FAM_GEMM_HEADS;
    if (epi == EPI_SWIGLU) return FAM_GEMM_SWIGLU;
    if (p.residual || p.out_lo) return two_wg ? FAM_GEMM_RESID2B : FAM_GEMM_RESID;
    return FAM_GEMM;
}

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
// LayerNorm fold, consumer side.  ln_coef_issue / ln_coef_finish: (rstd, -rstd * mean) of the MI rows this lane owns in the
// accumulator layout (row mwave + mi*16 + l15).  ln_apply: acc = rstd * acc - rstd * mean * c[n] + d[n].
// ------------------------------------------------------------------------------------------------
// The consumer reads the per-row pair (rstd, -rstd * mean) that pmhip_ln_coef computed from the hi plane -- 8 eight-byte
// loads per lane plus one for c | d, issued (inline asm, invisible to hipcc's waits) in the
// first read slot of the tile's LAST K-tile and retired by that K-tile's closing vmcnt(0), so the fold costs the persistent,
// streamed K loop nothing but 20 registers in its last K-tile.
// The coefficients travel by LDS-DMA into the wave's private scratch -- no register is held across the K loop (held in
// registers, 20 of them, the head-split variant spilled in its last K-tile: 154 vs 124 us):
//   scratch[0 .. 255]    (rstd, -rstd * mean) of the wave's 128 rows: 1 KiB contiguous in coef[M][2], ONE 16-byte DMA
//   scratch[256 .. 319]  c[nw .. nw+63],  scratch[320 .. 383]  d[nw .. nw+63]: one 4-byte DMA each
// issued in the first read slot of the tile's LAST K-tile and retired by that K-tile's closing vmcnt(0).
constexpr int LN_COEF_LOADS = 3;
__device__ __forceinline__ void ln_coef_issue(const GemmParams& p, int mwave, int nw, int lane, float* scratch) {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    __builtin_amdgcn_global_load_lds((gbl_void*)(p.ln_coef + (size_t)mwave * 2 + lane * 4), (lds_void*)scratch, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(p.ln_c + nw + lane), (lds_void*)(scratch + 256), 4, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(p.ln_d + nw + lane), (lds_void*)(scratch + 320), 4, 0, 0);
}

// HARDWARE HAZARD (gfx950, tools/hwtests/pkfma_mfma.hip): a v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 whose LOW half
// takes SRC1 from the HIGH register of its pair (`op_sel:[_,1,_]`, what hipcc emits to broadcast a scalar that sits in an odd
// register) loses that operand in lanes 48-63 -- the product comes out as zero -- when the OTHER wave of the SIMD issues the first
// bf16 MFMA of a block in the same cycles.  The lead waves of gemm256 normalise exactly while their lag waves start the tile's last
// MFMA block, and hipcc compiled the plain C++ of this function to 64 such instructions: one accumulator register wrong in one
// 16-lane group once per ~10^9 wave-tiles.  The per-row coefficients are therefore splatted into REAL register pairs through
// opaque moves and the packed FMAs are written without operand selects; tests/test_isa_hazards.py refuses any build that
// contains the vulnerable form.  The arithmetic (two fused multiply-adds per element) and its rounding are unchanged.
typedef float pk2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk2_t pk_splat(float v) {
    float lo, hi;
    asm("v_mov_b32 %0, %1" : "=v"(lo) : "v"(v));
    asm("v_mov_b32 %0, %1" : "=v"(hi) : "v"(v));
    return pk2_t{lo, hi};
}
__device__ __forceinline__ pk2_t pk_fma_plain(pk2_t a, pk2_t b, pk2_t c) {
    pk2_t d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// acc[0..3] = x * acc[0..3] + (y * c[0..3] + d[0..3]) for one 16x16 tile slice of a lane; x | y are splatted pairs
__device__ __forceinline__ void ln_apply4(f32x4_t& acc, pk2_t xx, pk2_t yy, const float4& c, const float4& d) {
    const pk2_t t0 = pk_fma_plain(pk2_t{c.x, c.y}, yy, pk2_t{d.x, d.y}), t1 = pk_fma_plain(pk2_t{c.z, c.w}, yy, pk2_t{d.z, d.w});
    const pk2_t o0 = pk_fma_plain(xx, pk2_t{acc[0], acc[1]}, t0), o1 = pk_fma_plain(xx, pk2_t{acc[2], acc[3]}, t1);
    acc[0] = o0[0]; acc[1] = o0[1]; acc[2] = o1[0]; acc[3] = o1[1];
}

template <int MI>
__device__ __forceinline__ void ln_apply(const float* scratch, f32x4_t (&acc)[MI][4], int lane) {
    const int g = lane >> 4, l15 = lane & 15;
    float4 cc[4], dd[4];                                           // c | d of the lane's 4 x 4 columns
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        cc[ni] = *reinterpret_cast<const float4*>(scratch + 256 + ni * 16 + g * 4);
        dd[ni] = *reinterpret_cast<const float4*>(scratch + 320 + ni * 16 + g * 4);
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const float2 ab = *reinterpret_cast<const float2*>(scratch + (mi * 16 + l15) * 2);      // row mwave + mi*16 + l15
        const pk2_t xx = pk_splat(ab.x), yy = pk_splat(ab.y);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) ln_apply4(acc[mi][ni], xx, yy, cc[ni], dd[ni]);
    }
}

// diagnostic variant: the same coefficients by plain global loads at epilogue time (no LDS-DMA involved)
template <int MI>
__device__ __forceinline__ void ln_apply_direct(const GemmParams& p, int mwave, int nw, f32x4_t (&acc)[MI][4], int lane) {
    const int g = lane >> 4, l15 = lane & 15;
    float4 cc[4], dd[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        cc[ni] = *reinterpret_cast<const float4*>(p.ln_c + nw + ni * 16 + g * 4);
        dd[ni] = *reinterpret_cast<const float4*>(p.ln_d + nw + ni * 16 + g * 4);
    }
    float2 ab[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) ab[mi] = *reinterpret_cast<const float2*>(p.ln_coef + (size_t)(mwave + mi * 16 + l15) * 2);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const pk2_t xx = pk_splat(ab[mi].x), yy = pk_splat(ab[mi].y);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) ln_apply4(acc[mi][ni], xx, yy, cc[ni], dd[ni]);
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
// NT = false (the small-batch launches of gemm.hip): ordinary stores.  There the whole output is a megabyte that the NEXT kernel of
// the dependent chain reads at once: it should stay in the cache hierarchy instead of being streamed past it.
template <bool NT> __device__ __forceinline__ void st16(void* p, uint4 v) {
    if constexpr (NT) nt_store16(p, v);
    else *reinterpret_cast<uint4*>(p) = v;
}
template <bool NT> __device__ __forceinline__ void st_row(float* p, const float (&v)[4]) { st16<NT>(p, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]))); }
template <bool NT> __device__ __forceinline__ void st_row(bf16_t* p, const float (&v)[8]) {
    st16<NT>(p, make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])));
}

template <int EPI, typename OutT, int MI, int NPRE, bool FULL = false, int RES = -1, typename Hook = NoHook, bool NT = true>
__device__ __forceinline__ void wave_epilogue(const GemmParams& p, f32x4_t (&acc)[MI][4], unsigned char* eraw, int mwave,
                                              int nw, int lane, const float4 (&rpre)[NPRE], Hook hook = Hook()) {
    constexpr int CPL = 16 / (int)sizeof(OutT);                    // columns per lane per store (16 B)
    constexpr int LPR = 64 / CPL, RPI = 64 / LPR, ITERS = 16 / RPI;
    const int l15 = lane & 15, g = lane >> 4;
    const int ccol = (lane % LPR) * CPL, ncol = nw + ccol;
    float* ebuf = reinterpret_cast<float*>(eraw);
    if (!FULL && nw >= p.N) return;

    if constexpr (EPI == EPI_HEADS) {
        // V part: TB-token x 64-d blocks (TB = 64; 32 for the 32-row wave tiles of the eight-wave small-batch GEMM) are transposed
        // through LDS as OutT so that V^T[b,h,d,:] rows are written 128 B / 64 B (bf16) at a time; Q and K take the generic
        // 16-row path below.
        static_assert(MI % 4 == 0 || MI == 2, "the head-split epilogue transposes V in blocks of 64 (or 32) tokens");
        if (p.kinds[nw / p.inner] == PMHIP_PART_V) {
            hook();
            constexpr int TB = MI >= 4 ? 64 : MI * 16;                         // tokens per transposed block
            constexpr int MIB = TB / 16;                                         // accumulator row tiles per block
            constexpr int LPRV = TB / 8, RPIV = 64 / LPRV, ITV = 64 / RPIV;      // 16-byte lanes per d-row, d-rows per store, stores
            const int h = (nw % p.inner) >> 6;
            OutT* dst = reinterpret_cast<OutT*>(p.outs[nw / p.inner]);
#pragma unroll
            for (int blk = 0; blk < MI / MIB; ++blk) {
                const int mblk = mwave + blk * TB;
                const int b0 = mblk / p.tokens, t0 = mblk % p.tokens;
                const bool whole = (FULL || mblk + TB - 1 < p.M) && (t0 + TB - 1 < p.tokens) && (t0 % 8 == 0);
                if (whole && sizeof(OutT) == 2) {
                    OutT* vbuf = reinterpret_cast<OutT*>(eraw);                  // [64 d][TB tokens]
#pragma unroll
                    for (int mi = 0; mi < MIB; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                vbuf[(ni * 16 + g * 4 + r) * TB + mi * 16 + l15] = from_f32<OutT>(acc[blk * MIB + mi][ni][r]);
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    // one per-lane base on each side + wave-uniform steps: eight separate row indices kept live across the
                    // whole kernel cost the folded head-split variant two spills, reloaded here behind a vmcnt(0) that drained
                    // the next tile's DMA (+24 us per launch)
                    OutT* vptr = dst + (((size_t)b0 * p.heads + h) * 64 + (lane / LPRV)) * p.tokens_pad + t0 + (lane % LPRV) * 8;
                    const unsigned char* lptr = eraw + (lane / LPRV) * (TB * 2) + (lane % LPRV) * 16;
                    const size_t vstep = (size_t)RPIV * p.tokens_pad;
#pragma unroll
                    for (int it = 0; it < ITV; ++it)                             // RPIV d-rows x TB*2 bytes per store instruction
                        st16<NT>(vptr + it * vstep, *reinterpret_cast<const uint4*>(lptr + it * (RPIV * TB * 2)));
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                } else {
#pragma unroll
                    for (int mi = 0; mi < MIB; ++mi) {
                        const int mm = mblk + mi * 16 + l15;
                        if (mm < p.M) {
                            const int b = mm / p.tokens, t = mm % p.tokens;
#pragma unroll
                            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    dst[(((size_t)b * p.heads + h) * 64 + ni * 16 + g * 4 + r) * p.tokens_pad + t] =
                                        from_f32<OutT>(acc[blk * MIB + mi][ni][r]);
                        }
                    }
                }
            }
            return;
        }
    }

#ifdef PM_ABL_EPI_SKIP                       // timing ablation (tools/epilogue_cost.py): no epilogue at all
    if (p.M > 0) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[MI - 1][3])); return; }
#endif
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
            st16<NT>(reinterpret_cast<OutT*>(p.out) + (size_t)(mwave + mi * 16 + erow) * p.ldo + (nw >> 1) + q * 8, v);
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
                st16<NT>(dst + (((size_t)b * p.heads + h) * tstride + t) * 64 + (lane & 7) * 8, v[it]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return;
    }
    // hi/lo residual stream: residual and output are pairs of bf16 planes, 8 columns (16 B of each plane) per lane
    constexpr bool HILO = (EPI == EPI_STD) && RES == 2 && sizeof(OutT) == 2;
    constexpr bool PIPE_HILO = HILO && FULL;
    // The residual pair of slice mi + HD is requested while slice mi is transposed, added and stored.  HD = 1: two and three slices
    // ahead measured 2-6 % SLOWER on the out-projection and equal on w3 (profiles/r04_producer_ablation.txt): the epilogue is
    // bound by the HBM read + write mix of all CUs storing at once, not by the bytes one CU keeps in flight
#ifndef PM_HILO_DEPTH
#define PM_HILO_DEPTH 1
#endif
    constexpr int HD = PIPE_HILO ? (PM_HILO_DEPTH < MI ? PM_HILO_DEPTH : MI) : 1;
    [[maybe_unused]] uint4 hq[HD][ITERS] = {}, lq[HD][ITERS] = {};
    [[maybe_unused]] float2 cq[HD][ITERS] = {};                 // (rstd, -rstd * mean) of the previous hi plane, per row of the slice
    [[maybe_unused]] float sq[HD][ITERS] = {};                  // running shift of the row (only the wave of column block 0 keeps it)
    [[maybe_unused]] const bool keeps_shift = HILO && p.shift && p.shift_mode == 2 && nw == 0;
    [[maybe_unused]] const bf16_t* res_hi = reinterpret_cast<const bf16_t*>(p.residual);
    [[maybe_unused]] auto load_pair = [&](int slice, uint4 (&h)[ITERS], uint4 (&l)[ITERS], float2 (&c)[ITERS], float (&sh)[ITERS]) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            if (p.center_coef) c[it] = *reinterpret_cast<const float2*>(p.center_coef + (size_t)(mwave + slice * 16 + it * RPI + lane / LPR) * 2);
            if (keeps_shift) sh[it] = p.shift[mwave + slice * 16 + it * RPI + lane / LPR];
#ifdef PM_ABL_NO_RESLOAD
            h[it] = make_uint4(0x3f803f80u + slice, 0, 0, 0); l[it] = make_uint4(0, 0, 0, 0);
#else
            const size_t off = (size_t)((mwave + slice * 16 + it * RPI + lane / LPR) % p.res_rows) * p.ldr + ncol;
            h[it] = *reinterpret_cast<const uint4*>(res_hi + off);
            l[it] = *reinterpret_cast<const uint4*>(p.res_lo + off);
#endif
        }
    };
    if constexpr (PIPE_HILO) {
#pragma unroll
        for (int d = 0; d < HD; ++d) load_pair(d, hq[d], lq[d], cq[d], sq[d]);
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
        [[maybe_unused]] uint4 hcur[ITERS], lcur[ITERS];
        [[maybe_unused]] float2 ccur[ITERS] = {};
        [[maybe_unused]] float scur[ITERS] = {};
        if constexpr (PIPE_HILO) {                          // ring slot mi % HD (mi is a compile-time constant after unrolling)
#pragma unroll
            for (int it = 0; it < ITERS; ++it) { hcur[it] = hq[mi % HD][it]; lcur[it] = lq[mi % HD][it]; ccur[it] = cq[mi % HD][it]; scur[it] = sq[mi % HD][it]; }
            if (mi + HD < MI) load_pair(mi + HD, hq[mi % HD], lq[mi % HD], cq[mi % HD], sq[mi % HD]);
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
                st16<NT>(reinterpret_cast<OutT*>(p.out) + (size_t)m * p.ldo + (nw >> 1) + q * 8, *reinterpret_cast<const uint4*>(eraw + erow * RS + q * 16));
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
                    st16<NT>(dst + (((size_t)b * p.heads + h) * tstride + t) * 64 + c16 * 8, *reinterpret_cast<const uint4*>(eraw + r * RS + c16 * 16));
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
                    if constexpr (HILO) {
                        uint4 rh, rl;
                        float2 cc = make_float2(1.f, 0.f);
                        float sh = 0.f;
                        if constexpr (PIPE_HILO) { rh = hcur[it]; rl = lcur[it]; cc = ccur[it]; sh = scur[it]; }
                        else {
                            const size_t off = (size_t)(mm % p.res_rows) * p.ldr + ncol;
                            rh = *reinterpret_cast<const uint4*>(res_hi + off);
                            rl = *reinterpret_cast<const uint4*>(p.res_lo + off);
                            if (p.center_coef) cc = *reinterpret_cast<const float2*>(p.center_coef + (size_t)mm * 2);
                            if (keeps_shift) sh = p.shift[mm];
                        }
                        // the row mean of the previous hi plane (0 without centring); every wave of the row derives the same bits
                        const float cen = p.center_coef ? p.center_extra - cc.y * __builtin_amdgcn_rcpf(cc.x) : 0.f;
                        if (p.shift && nw == 0 && (lane & (LPR - 1)) == 0) {
                            if (p.shift_mode == 1) p.shift[mm] = 0.f;
                            else if (p.shift_mode == 2) p.shift[mm] = sh + cen;
                        }
                        const unsigned hw[4] = {rh.x, rh.y, rh.z, rh.w}, lw[4] = {rl.x, rl.y, rl.z, rl.w};
                        unsigned oh[4], ol[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {              // two columns per word; hi + lo is exact in f32
                            const float x0 = v[2 * j] + ((__uint_as_float(hw[j] << 16) + __uint_as_float(lw[j] << 16)) - cen);
                            const float x1 = v[2 * j + 1] + ((__uint_as_float(hw[j] & 0xffff0000u) + __uint_as_float(lw[j] & 0xffff0000u)) - cen);
                            oh[j] = pack_bf16x2(x0, x1);
                            ol[j] = pack_bf16x2(x0 - __uint_as_float(oh[j] << 16), x1 - __uint_as_float(oh[j] & 0xffff0000u));
                        }
#ifdef PM_ABL_NO_STORE
                        if (mm < 0)
#endif
                        {
                        st16<NT>(reinterpret_cast<bf16_t*>(p.out) + (size_t)mm * p.ldo + ncol, make_uint4(oh[0], oh[1], oh[2], oh[3]));
                        st16<NT>(p.out_lo + (size_t)mm * p.ldo + ncol, make_uint4(ol[0], ol[1], ol[2], ol[3]));
                        }
                        if (p.row_stats) {                          // wave-uniform.  The 8 lanes of a row hold its 64 hi values of this wave
                            float hv[8];
#pragma unroll
                            for (int j = 0; j < 4; ++j) { hv[2 * j] = __uint_as_float(oh[j] << 16); hv[2 * j + 1] = __uint_as_float(oh[j] & 0xffff0000u); }
                            float sm = ((hv[0] + hv[1]) + (hv[2] + hv[3])) + ((hv[4] + hv[5]) + (hv[6] + hv[7]));
                            sm += dpp_mov<0xB1>(sm); sm += dpp_mov<0x4E>(sm); sm += dpp_mov<0x141>(sm);
                            const float pm = sm * (1.0f / 64.0f);
                            float q2 = 0.f;
#pragma unroll
                            for (int j = 0; j < 8; ++j) { const float a = hv[j] - pm; q2 = fmaf(a, a, q2); }
                            q2 += dpp_mov<0xB1>(q2); q2 += dpp_mov<0x4E>(q2); q2 += dpp_mov<0x141>(q2);
                            if ((lane & 7) == 0)
                                *reinterpret_cast<float2*>(p.row_stats + ((size_t)mm * (p.N >> 6) + (nw >> 6)) * 2) = make_float2(sm, q2);
                        }
                        continue;
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
                    st_row<NT>(reinterpret_cast<OutT*>(p.out) + (size_t)mm * p.ldo + ncol, v);
                    if constexpr (sizeof(OutT) == 4) {
                        if (p.block_stats) {                       // wave-uniform; N % 64 == 0: the 16 lanes of a row are all active
                            const float2 st = softmax_block_stat(v[0], v[1], v[2], v[3]);
                            if ((lane & (LPR - 1)) == 0)
                                *reinterpret_cast<float2*>(p.block_stats + ((size_t)mm * (p.N >> 6) + (nw >> 6)) * 2) = st;
                        }
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
                    st_row<NT>(dst + (((size_t)b * p.heads + h) * tstride + t) * 64 + ccol, v);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next 16 rows overwrite ebuf
    }
}

}  // namespace pmgemm
