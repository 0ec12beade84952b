// EXPERIMENT, not built into libpaintmind_hip.so (round 2).  Result: correct (bit-level agreement with pmhip_gemm + pmhip_layernorm
// within one bf16 ulp, fp64 check), but SLOWER than the two kernels it replaces: 95.6 us vs 62.4 + 30 us at M = 65536, K = 512 and
// 162.8 vs 110.9 + 30 us at K = 1408.  The lockstep K loop below (all 8 waves issue DMA, read and multiply in the same phase)
// runs at ~2400 cycles per 32 MFMAs; to win, this tile shape needs the lead / lag slot structure of gemm256.hip (estimated
// 80 / 124 us, i.e. ~4 % of the bench step), which was not built.  Kept for the epilogue (row-major re-ownership through LDS,
// buffer_load / buffer_store with scalar row offsets, two-pass row statistics across waves).
// Residual GEMM with the FOLLOWING LayerNorm in its epilogue (bf16 mode, N = 512 = one whole row of the residual stream per
// workgroup).  Reference: every projection back into the residual stream is followed by a LayerNorm of the updated stream
// (stage1/layers.py:54-58, stage2/transformer.py:44-49):
//     x = A . W^T + bias + residual          (f32, written back in place of the residual)
//     y = LayerNorm(x) * gamma + beta        (bf16, the next projection's input)
// Why: as two kernels the pair moves 64 + 128 + 128 MB (GEMM) + 128 + 64 MB (LayerNorm) per launch at M = 65536 and both are
// HBM-bound; fused, the LayerNorm's 128 MB read of x disappears (384 instead of 512 MB), and so does its launch.
//
// Geometry: 512 threads = 8 waves as 2(m) x 4(n), tile 128 rows x 512 columns, a wave owns 64 rows x 128 columns = acc[4][8]
// MFMA tiles (128 accumulator registers).  (A first version with 64-row tiles and two 4-wave workgroups per CU re-loaded W
// twice as often: 9 DMA instructions per wave per 32 MFMAs, 99 us at K = 512 against 62 + 30 us for the separate kernels.)
// K advances in tiles of 32 (64 B per row) through a 3-stage LDS ring, prefetch distance 2:
//   LDS image of a K-tile: A [128 rows][64 B] then W [512 rows][64 B]; the 16-byte slot of (row, k-chunk g) is
//   g ^ perm[(row >> 2) & 3], perm = {0,3,2,1} (the 16 lanes the LDS serves together for a ds_read_b128 then touch 16 different
//   16-byte bank groups); the swizzle is applied on the DMA's per-lane SOURCE offset (the LDS side of the DMA is lane-linear).
//   per K-tile t:  vmcnt(5): DMA(t) landed, DMA(t+1) may fly | barrier | DMA(t+2) -> the stage read at t-1 | 12 fragment reads | 32 MFMAs
// Epilogue, per 16-row slice: accumulators -> LDS -> row-major registers (a lane owns 4 consecutive columns of 2 rows per step,
// so residual loads and x stores cover 512 contiguous bytes per row); row sums over the wave's 128 columns by DPP / permlane,
// over the 4 waves through LDS; mean first, then the centred second moment (the same two-pass arithmetic as layernorm_kernel).
#include "gemm_common.h"

using namespace pmgemm;

namespace {

constexpr int BM = 128, BN = 512, BK = 32, THREADS = 512;
constexpr int A_BYTES = BM * 64;                   // 8 KiB
constexpr int W_BYTES = BN * 64;                   // 32 KiB
constexpr int STAGE = A_BYTES + W_BYTES;           // one K-tile: 40 KiB
constexpr int LDS_BYTES = 3 * STAGE;               // 120 KiB
constexpr int ESTR = 132;                          // floats per staged row: 128 + pad

typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ unsigned slot_perm(unsigned rowblk) { return (0x1230u >> (4 * (rowblk & 3))) & 3u; }   // {0,3,2,1}

// sum over the 32 lanes of a half wave (lanes 0-31 / 32-63); every lane of the half gets the result
__device__ __forceinline__ float half_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}

struct RowLnParams {
    const void* A; const void* W;
    const float* bias; const float* residual; float* x_out;
    const float* gamma; const float* beta; bf16_t* y_out;
    int lda, ldw, M, K, res_rows;
    float eps;
};

__global__ __launch_bounds__(THREADS) void gemm_rowln_kernel(const RowLnParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * BM;

    // ---- DMA: one instruction = 16 rows x 64 B; lane -> (row lane >> 2, LDS slot lane & 3), source chunk = slot ^ perm
    const rsrc_t Ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, 0x7fffffff, 0x00020000);
    const rsrc_t Wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, 0x7fffffff, 0x00020000);
    const unsigned lda_b = (unsigned)p.lda * 2u, ldw_b = (unsigned)p.ldw * 2u;
    const unsigned lswz = (((unsigned)lane & 3u) ^ slot_perm((unsigned)lane >> 4)) << 4;
    const unsigned avoff = (unsigned)(lane >> 2) * lda_b + lswz, wvoff = (unsigned)(lane >> 2) * ldw_b + lswz;
    // wave w copies A rows [16 w, 16 w + 16) and W rows [64 w, 64 w + 64) of the tile: 5 instructions per K-tile
    const unsigned asoff = (unsigned)(m0 + wave * 16) * lda_b, wsoff = (unsigned)(wave * 64) * ldw_b;
#define LDSP(ptr) ((__attribute__((address_space(3))) void*)(ptr))
    auto issue = [&](int kt, int stage) {
        unsigned char* st = lds + stage * STAGE;
        const unsigned kb = (unsigned)kt * (BK * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(Ar, LDSP(st + wave * 1024), 16, avoff, asoff + kb, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Wr, LDSP(st + A_BYTES + (wave * 4 + i) * 1024), 16, wvoff, wsoff + (unsigned)i * 16u * ldw_b + kb, 0, 0);
    };

    // ---- fragment addresses: row l15 (mod 16) fixes the swizzled slot, the fragment index is an immediate offset
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned fslot = ((unsigned)g ^ slot_perm((unsigned)l15 >> 2)) << 4;
    const unsigned fa = lds_base + (unsigned)(wm * 64 + l15) * 64u + fslot;
    const unsigned fw = lds_base + A_BYTES + (unsigned)(wn * 128 + l15) * 64u + fslot;

    f32x4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // inline-asm LDS reads: hipcc would put `s_waitcnt vmcnt(0)` in front of every LDS read while a DMA is in flight
#define DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    const int nk = p.K / BK;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int st = 0;                                            // ring stage of K-tile kt
    for (int kt = 0; kt < nk; ++kt) {
        uint4 a[4], w[8];
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      // K-tile kt landed; the 5 instructions of kt+1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // K-tile kt visible; nobody still reads the stage of K-tile kt-1
        const int st2 = st == 0 ? 2 : st - 1;              // (kt + 2) % 3 == (kt - 1) % 3
        if (kt + 2 < nk) issue(kt + 2, st2);
        const unsigned ba = fa + (unsigned)st * STAGE, bw = fw + (unsigned)st * STAGE;
        st = st == 2 ? 0 : st + 1;
        DSR(w[0], bw, 0 * 1024); DSR(w[1], bw, 1 * 1024); DSR(w[2], bw, 2 * 1024); DSR(w[3], bw, 3 * 1024);
        DSR(a[0], ba, 0 * 1024); DSR(a[1], ba, 1 * 1024); DSR(a[2], ba, 2 * 1024); DSR(a[3], ba, 3 * 1024);
        DSR(w[4], bw, 4 * 1024); DSR(w[5], bw, 5 * 1024); DSR(w[6], bw, 6 * 1024); DSR(w[7], bw, 7 * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int nf = 0; nf < 8; ++nf) Mma<bf16_t>::run(acc[mi][nf], w[nf], a[mi]);
        __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_s_barrier();                          // all fragment reads done: LDS is free for the epilogue
#undef DSR

    // ---- epilogue.  LDS: per wave a 16 x 132 float staging slab, then the cross-wave row statistics [2][4 waves][64 rows].
    // Global accesses are buffer operations: one descriptor per tensor, ONE per-lane vector offset (row-in-pair, column), the
    // row in the scalar offset -- no 64-bit address VGPRs (the flat form spilled 336 bytes of them).  Stores are non-temporal.
    float* ebuf = reinterpret_cast<float*>(lds) + wave * (16 * ESTR);
    float* stat = reinterpret_cast<float*>(lds) + 8 * (16 * ESTR) + wm * 256;       // [pass: +512][wm][wn][row]
    const int hrow = lane >> 5, c4 = (lane & 31) * 4;                              // row-major ownership: 2 rows per step
    const int col = wn * 128 + c4;
    const int m0w = m0 + wm * 64;                                                  // first row of this wave
    const rsrc_t Rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual), 0, 0x7fffffff, 0x00020000);
    const rsrc_t Xr = __builtin_amdgcn_make_buffer_rsrc(p.x_out, 0, 0x7fffffff, 0x00020000);
    const rsrc_t Yr = __builtin_amdgcn_make_buffer_rsrc(p.y_out, 0, 0x7fffffff, 0x00020000);
    const unsigned voff4 = (unsigned)(hrow * BN + col) * 4u, voff2 = (unsigned)(hrow * BN + col) * 2u;
    const unsigned rrow0 = (unsigned)(m0w % p.res_rows);                            // res_rows is a multiple of 64 (or == M): no wrap inside a tile
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
    const float4 bias4 = *reinterpret_cast<const float4*>(p.bias + col);
    const float4 gm = *reinterpret_cast<const float4*>(p.gamma + col);
    const float4 bt = *reinterpret_cast<const float4*>(p.beta + col);
    float4 x[4][8];                                                                // [slice][step]: row = 16 slice + 2 step + hrow
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int nf = 0; nf < 8; ++nf) *reinterpret_cast<f32x4_t*>(ebuf + l15 * ESTR + nf * 16 + g * 4) = acc[mi][nf];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = it * 2 + hrow;
            const float4 t = *reinterpret_cast<const float4*>(ebuf + r * ESTR + c4);
            const v4u_t rr = __builtin_amdgcn_raw_buffer_load_b128(Rr, voff4, (rrow0 + mi * 16 + it * 2) * (BN * 4u), 0);
            x[mi][it] = make_float4(t.x + bias4.x + __uint_as_float(rr.x), t.y + bias4.y + __uint_as_float(rr.y),
                                    t.z + bias4.z + __uint_as_float(rr.z), t.w + bias4.w + __uint_as_float(rr.w));
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          // slab reads done before the next slice overwrites it
    }
    // x goes out as soon as it is final (512 contiguous bytes per row and wave)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int it = 0; it < 8; ++it)
            __builtin_amdgcn_raw_buffer_store_b128(v4u_t{__float_as_uint(x[mi][it].x), __float_as_uint(x[mi][it].y), __float_as_uint(x[mi][it].z),
                                                         __float_as_uint(x[mi][it].w)},
                                                   Xr, voff4, (unsigned)(m0w + mi * 16 + it * 2) * (BN * 4u), 2);
    // pass 1: row means
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const float s = half_sum((x[mi][it].x + x[mi][it].y) + (x[mi][it].z + x[mi][it].w));
            if ((lane & 31) == 0) stat[wn * 64 + mi * 16 + it * 2 + hrow] = s;
        }
    __syncthreads();
    float mean[4][8];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = mi * 16 + it * 2 + hrow;
            mean[mi][it] = ((stat[r] + stat[64 + r]) + (stat[128 + r] + stat[192 + r])) * (1.0f / BN);
        }
    // pass 2: centred second moments
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const float a = x[mi][it].x - mean[mi][it], b = x[mi][it].y - mean[mi][it], c = x[mi][it].z - mean[mi][it], d = x[mi][it].w - mean[mi][it];
            const float q = half_sum((a * a + b * b) + (c * c + d * d));
            if ((lane & 31) == 0) stat[512 + wn * 64 + mi * 16 + it * 2 + hrow] = q;
        }
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = mi * 16 + it * 2 + hrow;
            const float var = ((stat[512 + r] + stat[576 + r]) + (stat[640 + r] + stat[704 + r])) * (1.0f / BN);
            const float rstd = 1.0f / sqrtf(var + p.eps);
            const float mu = mean[mi][it];
            const v2u_t yv = {pack_bf16x2((x[mi][it].x - mu) * rstd * gm.x + bt.x, (x[mi][it].y - mu) * rstd * gm.y + bt.y),
                              pack_bf16x2((x[mi][it].z - mu) * rstd * gm.z + bt.z, (x[mi][it].w - mu) * rstd * gm.w + bt.w)};
            __builtin_amdgcn_raw_buffer_store_b64(yv, Yr, voff2, (unsigned)(m0w + mi * 16 + it * 2) * (BN * 2u), 2);
        }
#undef LDSP
}

}  // namespace

extern "C" int pmhip_gemm_res_ln_supported(int dtype, int M, int N, int K) {
    return dtype == PMHIP_BF16 && N == BN && M > 0 && M % BM == 0 && K >= BK && K % BK == 0 &&
                   (unsigned long long)M * K * 2 < (1ull << 31) && (unsigned long long)M * BN * 4 < (1ull << 31)
               ? 1 : 0;
}

extern "C" int pmhip_gemm_res_ln(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, const float* residual,
                                 int res_rows, float* x_out, const float* gamma, const float* beta, float eps, void* y_out, int M,
                                 int N, int K, pmhip_stream stream) {
    PM_REQUIRE(pmhip_gemm_res_ln_supported(dtype, M, N, K), "gemm_res_ln: needs bf16, N = 512, M %% 128 == 0, K %% 32 == 0 (got M=%d N=%d K=%d)", M, N, K);
    PM_REQUIRE(A && W && bias && residual && x_out && gamma && beta && y_out, "gemm_res_ln: null pointer");
    PM_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K, "gemm_res_ln: lda/ldw must be multiples of 8 and >= K");
    PM_REQUIRE(res_rows > 0 && (res_rows == M || res_rows % BM == 0), "gemm_res_ln: res_rows=%d must be M or a multiple of 128", res_rows);
    RowLnParams p{};
    p.A = A; p.W = W; p.bias = bias; p.residual = residual; p.x_out = x_out; p.gamma = gamma; p.beta = beta;
    p.y_out = reinterpret_cast<bf16_t*>(y_out);
    p.lda = lda; p.ldw = ldw; p.M = M; p.K = K; p.res_rows = res_rows; p.eps = eps;
    hipStream_t s = (hipStream_t)stream;
    PmTimer tm(FAM_GEMM, s);
    hipLaunchKernelGGL(gemm_rowln_kernel, dim3(M / BM), dim3(THREADS), 0, s, p);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}
