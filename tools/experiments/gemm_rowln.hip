// EXPERIMENT, not built into libpaintmind_hip.so (round 2).  Second version: the lead / lag K loop of gemm256.hip on a 128 x 512
// tile (1 x 8 waves), persistent and streamed, with the row-major LayerNorm epilogue.  Correct on the first run (tests below
// the kernel's former API: pmhip_gemm + pmhip_layernorm within one bf16 ulp, fp64 check, 258-tile persistent case), but NOT
// faster than the two kernels it replaces: 89.4-90.0 us vs 60.7 + 29.1 us at M = 65536, K = 512; 129.7 vs 107.4 + 28.8 us at
// K = 1408.  Its K loop runs at the speed of gemm256 (1.47 us per K-tile) and its epilogue streams 384 MB at 5.0 TB/s, but with
// one 8-wave workgroup per CU (160 KiB of LDS) the two do not overlap, whereas the separate kernels are each HBM-bound at 5.3 and
// 6.6 TB/s: saving a quarter of the bytes buys nothing.  Requesting the residual rows four slices ahead changed nothing.
// (The first version, 8 waves in lockstep, took 95.6 / 162.8 us.)
// Residual GEMM with the FOLLOWING LayerNorm in its epilogue (bf16 mode, N = 512 = one whole row of the residual stream per
// workgroup).  Reference: every projection back into the residual stream is followed by a LayerNorm of the updated stream
// (stage1/layers.py:54-58, stage2/transformer.py:44-49):
//     x = A . W^T + bias + residual          (f32, written back in place of the residual)
//     y = LayerNorm(x) * gamma + beta        (bf16, the next projection's input)
// Why: as two kernels the pair moves 64 + 128 + 128 MB (GEMM) + 128 + 64 MB (LayerNorm) per launch at M = 65536 and both are
// HBM-bound; fused, the LayerNorm's 128 MB read of x disappears (384 instead of 512 MB), and so does its launch.
//
// Geometry: 512 threads = 8 waves, tile 128 rows x 512 columns; wave w owns all 128 rows of columns [64 w, 64 w + 64) =
// acc[8][4] MFMA tiles -- the wave tile, the four phases of a K-tile, the lead / lag slot structure, the LDS swizzle, the
// streamed tiles and the vmcnt discipline are those of gemm256.hip (read its header first); what differs is the shape of
// the workgroup (1 x 8 waves instead of 2 x 4), hence the DMA pieces:
//     PA0 / PA1 : A rows [0,64) / [64,128)          8 KiB = 1 DMA instruction per wave
//     PW0 / PW1 : rows [0,32) / [32,64) of every wave's 64-row block of W     32 KiB = 4 instructions per wave
// and the counted waits (k_tile below).  LDS: two K-tile buffers of 80 KiB (A 16 + W 64) = all 160 KiB of the CU.
// (A first version of this kernel, `tools/experiments/gemm_rowln.hip`, ran its 8 waves in lockstep: 2400 cycles per 32 MFMAs,
// 95.6 us at K = 512 against 62 + 30 us for the separate kernels.)
// Epilogue, per 16-row slice: accumulators -> LDS -> row-major registers (16 lanes own the wave's 64 columns of a row, 4 rows
// per step), residual loads / x stores as buffer operations with the row in the scalar offset; row sums over the wave's 64
// columns by DPP, over the 8 waves through LDS; mean first, then the centred second moment (the two-pass arithmetic of
// layernorm_kernel, another summation order).
#include "gemm_common.h"

using namespace pmgemm;

namespace {

constexpr int BM = 128, BN = 512, THREADS = 512;
constexpr int A_BYTES = BM * ROWB;                 // 16 KiB
constexpr int W_BYTES = BN * ROWB;                 // 64 KiB
constexpr int BUF_BYTES = A_BYTES + W_BYTES;       // one K-tile: 80 KiB
constexpr int KSTEP = ROWB / 2;                    // 64 bf16

typedef __amdgpu_buffer_rsrc_t rsrc_t;
enum { PA0 = 0, PW0 = 1, PW1 = 2, PA1 = 3 };
enum { KT_STEADY = 0, KT_FIRST = 1, KT_LAST = 2 };

struct RowLnParams {
    const void* A; const void* W;
    const float* bias; const float* residual; float* x_out;
    const float* gamma; const float* beta; bf16_t* y_out;
    int lda, ldw, M, K, res_rows;
    float eps;
};

struct LoopCtx {
    rsrc_t Ar, Wr;
    unsigned aorg, aorg1;                               // byte offsets of this / the next tile's first row in A
    unsigned lda_b, ldw_b, laneoffA, laneoffW;
    unsigned fa0, fa1, fw0, fw1, lds_base;
    int nk, wave;
};

#define LDSP(ptr) ((__attribute__((address_space(3))) void*)(ptr))
// one DMA piece of K-tile starting at k0 into buffer `buf` (A pieces: `aorg` selects the tile)
__device__ __forceinline__ void issue_piece(int piece, const LoopCtx& c, unsigned aorg, int k0, unsigned char* buf) {
    const unsigned kb = (unsigned)k0 * 2u;
    if (piece == PA0 || piece == PA1) {
        const int row0 = (piece == PA1 ? 64 : 0) + c.wave * 8;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(c.Ar, LDSP(buf + row0 * ROWB), 16, c.laneoffA, aorg + (unsigned)row0 * c.lda_b + kb, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row0 = c.wave * 64 + (piece == PW1 ? 32 : 0) + i * 8;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(c.Wr, LDSP(buf + A_BYTES + row0 * ROWB), 16, c.laneoffW, (unsigned)row0 * c.ldw_b + kb, 0, 0);
        }
    }
}

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
#define LGKM0 asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0)
#define BAR __builtin_amdgcn_s_barrier()
#define VMW(n) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

// One K-tile for one wave (gemm256.hip: k_tile).  DMA instructions per wave: PA0 1, PW0 4, PW1 4, PA1 1.
//   KT_STEADY  issues the next K-tile of the same tile, one piece per phase.  The wait at the end of a phase retires what the
//              NEXT phase reads: after phase 1 PW1 (in flight behind it: PA1 + PA0' = 2), after phase 2 PA1 (PA0' + PW0' = 5),
//              after phase 4 PA0' and PW0' (PW1' + PA1' = 5)
//   KT_FIRST   all four pieces have landed; the only younger VMEM operations are the previous epilogue's stores: no wait in
//              phases 1-3 (the stores drain under the MFMAs), vmcnt(5) after phase 4
//   KT_LAST    issues the NEXT tile's first K-tile into the other buffer, PA0'' + PW0'' in phase 1, PW1'' + PA1'' in phase 2:
//              after phase 1 PW1 must be in (PA1 + 5 new = 6), after phase 2 PA1 (10 new), the closing wait is vmcnt(0)
template <bool LEAD, int KIND>
__device__ __forceinline__ void k_tile(const LoopCtx& c, unsigned char* lds, f32x4_t (&acc)[8][4], uint4 (&a)[4][2], uint4 (&w)[2][2],
                                       int kt, int pb) {
    constexpr bool FIRST = KIND == KT_FIRST;
    const unsigned boff = (unsigned)((kt + pb) & 1) * BUF_BYTES;
    const unsigned ba0 = c.lds_base + boff + c.fa0, ba1 = c.lds_base + boff + c.fa1;
    const unsigned bw0 = c.lds_base + boff + c.fw0, bw1 = c.lds_base + boff + c.fw1;
    unsigned char* nxt = lds + ((kt + 1 + pb) & 1) * BUF_BYTES;
#define DMA(phase)                                                                                                   \
    if constexpr (KIND != KT_LAST) {                                                                                 \
        issue_piece(phase == 1 ? PA0 : phase == 2 ? PW0 : phase == 3 ? PW1 : PA1, c, c.aorg, (kt + 1) * KSTEP, nxt); \
    } else {                                                                                                         \
        if constexpr (phase == 1) { issue_piece(PA0, c, c.aorg1, 0, nxt); issue_piece(PW0, c, c.aorg1, 0, nxt); }    \
        else if constexpr (phase == 2) { issue_piece(PW1, c, c.aorg1, 0, nxt); issue_piece(PA1, c, c.aorg1, 0, nxt); } \
    }
#define WAIT(phase)                                                                                                  \
    if constexpr (KIND == KT_STEADY) {                                                                               \
        if constexpr (phase == 1) VMW(2) else if constexpr (phase == 2) VMW(5) else if constexpr (phase == 4) VMW(5) \
    } else if constexpr (KIND == KT_FIRST) {                                                                         \
        if constexpr (phase == 4) VMW(5)                                                                             \
    } else {                                                                                                         \
        if constexpr (phase == 1) VMW(6) else if constexpr (phase == 2) VMW(10) else if constexpr (phase == 4) VMW(0) \
    }
    // ---------------- phase 1: A rows [0,64) x W rows [0,32)
    if constexpr (LEAD) { RD_A(0) RD_W(0) DMA(1) } else if constexpr (!FIRST) { LGKM0; MMA(1, 0) }
    BAR;
    if constexpr (LEAD) { LGKM0; MMA(0, 0) } else { RD_A(0) RD_W(0) DMA(1) }
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
    // ---------------- phase 4: A rows [64,128) x W rows [0,32) (re-read)
    if constexpr (LEAD) { RD_W(0) DMA(4) } else { LGKM0; MMA(1, 1) }
    BAR;
    if constexpr (LEAD) { LGKM0; MMA(1, 0) } else { RD_W(0) DMA(4) }
    WAIT(4)
    BAR;
#undef DMA
#undef WAIT
}

template <bool LEAD>
__device__ __forceinline__ void k_loop(const LoopCtx& c, unsigned char* lds, f32x4_t (&acc)[8][4], int pb) {
    uint4 a[4][2], w[2][2];
    k_tile<LEAD, KT_FIRST>(c, lds, acc, a, w, 0, pb);
    for (int kt = 1; kt + 1 < c.nk; ++kt) k_tile<LEAD, KT_STEADY>(c, lds, acc, a, w, kt, pb);
    k_tile<LEAD, KT_LAST>(c, lds, acc, a, w, c.nk - 1, pb);
    if constexpr (!LEAD) { LGKM0; MMA(1, 0) }
}
#undef DSR
#undef RD_A
#undef RD_W
#undef MMA
#undef LGKM0
#undef BAR
#undef VMW

// sum over the 16 lanes of a DPP row; every lane of the row gets the result
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}

__global__ __launch_bounds__(THREADS) void gemm_rowln_kernel(const RowLnParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * BUF_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;

    const int ntiles = p.M / BM;
    LoopCtx c;
    c.Ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, 0x7fffffff, 0x00020000);
    c.Wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, 0x7fffffff, 0x00020000);
    c.lda_b = (unsigned)p.lda * 2u; c.ldw_b = (unsigned)p.ldw * 2u;
    const unsigned lswz = (unsigned)(((lane & 7) ^ ((lane >> 3) & 7)) << 4);       // chunk rows start at multiples of 8
    c.laneoffA = (unsigned)(lane >> 3) * c.lda_b + lswz;
    c.laneoffW = (unsigned)(lane >> 3) * c.ldw_b + lswz;
    c.fa0 = (unsigned)(l15 * ROWB) + ((unsigned)((0 + g) ^ (l15 & 7)) << 4);
    c.fa1 = (unsigned)(l15 * ROWB) + ((unsigned)((4 + g) ^ (l15 & 7)) << 4);
    c.fw0 = (unsigned)(A_BYTES + (wave * 64 + l15) * ROWB) + ((unsigned)((0 + g) ^ (l15 & 7)) << 4);
    c.fw1 = (unsigned)(A_BYTES + (wave * 64 + l15) * ROWB) + ((unsigned)((4 + g) ^ (l15 & 7)) << 4);
    c.nk = p.K / KSTEP;
    c.wave = wave;
    c.lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;

    int tile = blockIdx.x;
    int m0 = tile * BM;
    c.aorg = (unsigned)m0 * c.lda_b;

    // prologue of the FIRST tile: its whole first K-tile (KT_FIRST takes no waits for it)
    issue_piece(PA0, c, c.aorg, 0, lds); issue_piece(PW0, c, c.aorg, 0, lds);
    issue_piece(PW1, c, c.aorg, 0, lds); issue_piece(PA1, c, c.aorg, 0, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

    int pb = 0;
    for (;;) {
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < ntiles;
        c.aorg1 = (unsigned)((has_next ? next : tile) * BM) * c.lda_b;     // no next tile: KT_LAST re-reads this tile's first K-tile

        if (wave < 4) k_loop<true>(c, lds, acc, pb); else k_loop<false>(c, lds, acc, pb);

        // ---- epilogue in the buffer of the LAST K-tile (the other one holds the next tile's first K-tile):
        //      8 staging slabs of 8 KiB, then the cross-wave row statistics [pass][row][wave]
        // (everything the epilogue needs is derived HERE, from the lane id: values computed before the tile loop stay live through
        // the K loop and are spilled around it, and a reload's compiler-inserted vmcnt(0) serialises the hand-placed DMA)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));                                               // opaque: not hoisted out of the tile loop
        const int hrow = lane_e >> 4, c4 = (lane_e & 15) * 4;                          // row-major ownership: 16 lanes x 4 columns, 4 rows per step
        const int col = wave * 64 + c4;
        const rsrc_t Rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual), 0, 0x7fffffff, 0x00020000);
        const rsrc_t Xr = __builtin_amdgcn_make_buffer_rsrc(p.x_out, 0, 0x7fffffff, 0x00020000);
        const rsrc_t Yr = __builtin_amdgcn_make_buffer_rsrc(p.y_out, 0, 0x7fffffff, 0x00020000);
        const unsigned voff4 = (unsigned)(hrow * BN + col) * 4u, voff2 = (unsigned)(hrow * BN + col) * 2u;
        unsigned char* fb = lds + ((pb + c.nk - 1) & 1) * BUF_BYTES;
        float* ebuf = reinterpret_cast<float*>(fb + wave * EPI_WAVE_BYTES);
        float* stat = reinterpret_cast<float*>(fb + 8 * EPI_WAVE_BYTES);                 // [2][128 rows][8 waves]
        const unsigned rrow0 = (unsigned)(m0 % p.res_rows);                            // res_rows == M or a multiple of 128: no wrap inside a tile
        const float4 bias4 = *reinterpret_cast<const float4*>(p.bias + col);
        float4 x[8][4];                                                                // [slice][step]: row = 16 slice + 4 step + hrow
        // the residual rows are requested FOUR slices ahead of their use (64 registers: as the accumulator slices die, x and
        // these take their place), so that the HBM latency is paid once per tile, not once per slice
        v4u_t rr[8][4];
        auto res_issue = [&](int mi) {
#pragma unroll
            for (int it = 0; it < 4; ++it) rr[mi][it] = __builtin_amdgcn_raw_buffer_load_b128(Rr, voff4, (rrow0 + mi * 16 + it * 4) * (BN * 4u), 0);
        };
        res_issue(0); res_issue(1); res_issue(2); res_issue(3);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) *reinterpret_cast<f32x4_t*>(ebuf + l15 * ESTRIDE + ni * 16 + g * 4) = acc[mi][ni];
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float4 t = *reinterpret_cast<const float4*>(ebuf + (it * 4 + hrow) * ESTRIDE + c4);
                x[mi][it] = make_float4(t.x + bias4.x + __uint_as_float(rr[mi][it].x), t.y + bias4.y + __uint_as_float(rr[mi][it].y),
                                        t.z + bias4.z + __uint_as_float(rr[mi][it].z), t.w + bias4.w + __uint_as_float(rr[mi][it].w));
            }
            if (mi + 4 < 8) res_issue(mi + 4);
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          // slab reads done before the next slice overwrites it
        }
        // x goes out as soon as it is final (non-temporal, 256 contiguous bytes per row and wave)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int it = 0; it < 4; ++it)
                __builtin_amdgcn_raw_buffer_store_b128(v4u_t{__float_as_uint(x[mi][it].x), __float_as_uint(x[mi][it].y), __float_as_uint(x[mi][it].z),
                                                             __float_as_uint(x[mi][it].w)},
                                                       Xr, voff4, (unsigned)(m0 + mi * 16 + it * 4) * (BN * 4u), 2);
        // pass 1: row means
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float s = row16_sum((x[mi][it].x + x[mi][it].y) + (x[mi][it].z + x[mi][it].w));
                if ((lane_e & 15) == 0) stat[(mi * 16 + it * 4 + hrow) * 8 + wave] = s;
            }
        __syncthreads();
        // pass 2: centred second moments (the mean is re-read from LDS where it is needed: 32 registers less than keeping it)
        auto row_mean = [&](int mi, int it) {
            const float* sp = stat + (mi * 16 + it * 4 + hrow) * 8;
            const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
            return (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) * (1.0f / BN);
        };
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float mu = row_mean(mi, it);
                const float a = x[mi][it].x - mu, b = x[mi][it].y - mu, cc = x[mi][it].z - mu, d = x[mi][it].w - mu;
                const float q = row16_sum((a * a + b * b) + (cc * cc + d * d));
                if ((lane_e & 15) == 0) stat[1024 + (mi * 16 + it * 4 + hrow) * 8 + wave] = q;
            }
        __syncthreads();
        const float4 gm = *reinterpret_cast<const float4*>(p.gamma + col);
        const float4 bt = *reinterpret_cast<const float4*>(p.beta + col);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float* sp = stat + 1024 + (mi * 16 + it * 4 + hrow) * 8;
                const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
                const float var = (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) * (1.0f / BN);
                const float rstd = 1.0f / sqrtf(var + p.eps);
                const float mu = row_mean(mi, it);
                const v2u_t yv = {pack_bf16x2((x[mi][it].x - mu) * rstd * gm.x + bt.x, (x[mi][it].y - mu) * rstd * gm.y + bt.y),
                                  pack_bf16x2((x[mi][it].z - mu) * rstd * gm.z + bt.z, (x[mi][it].w - mu) * rstd * gm.w + bt.w)};
                __builtin_amdgcn_raw_buffer_store_b64(yv, Yr, voff2, (unsigned)(m0 + mi * 16 + it * 4) * (BN * 2u), 2);
            }
        if (!has_next) break;
        __builtin_amdgcn_s_barrier();                  // every wave is done with the free buffer (the next K-tile 1 lands there)
        tile = next; m0 = tile * BM;
        c.aorg = c.aorg1;
        pb = (pb + c.nk) & 1;
    }
}
#undef LDSP

}  // namespace

extern "C" int pmhip_gemm_res_ln_supported(int dtype, int M, int N, int K) {
    return dtype == PMHIP_BF16 && N == BN && M > 0 && M % BM == 0 && K >= 2 * KSTEP && K % KSTEP == 0 &&
                   (unsigned long long)M * K * 2 < (1ull << 31) && (unsigned long long)M * BN * 4 < (1ull << 31)
               ? 1 : 0;
}

extern "C" int pmhip_gemm_res_ln(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, const float* residual,
                                 int res_rows, float* x_out, const float* gamma, const float* beta, float eps, void* y_out, int M,
                                 int N, int K, pmhip_stream stream) {
    PM_REQUIRE(pmhip_gemm_res_ln_supported(dtype, M, N, K), "gemm_res_ln: needs bf16, N = 512, M %% 128 == 0, K %% 64 == 0, K >= 128 (got M=%d N=%d K=%d)", M, N, K);
    PM_REQUIRE(A && W && bias && residual && x_out && gamma && beta && y_out, "gemm_res_ln: null pointer");
    PM_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K, "gemm_res_ln: lda/ldw must be multiples of 8 and >= K");
    PM_REQUIRE(res_rows > 0 && (res_rows == M || res_rows % BM == 0), "gemm_res_ln: res_rows=%d must be M or a multiple of 128", res_rows);
    RowLnParams p{};
    p.A = A; p.W = W; p.bias = bias; p.residual = residual; p.x_out = x_out; p.gamma = gamma; p.beta = beta;
    p.y_out = reinterpret_cast<bf16_t*>(y_out);
    p.lda = lda; p.ldw = ldw; p.M = M; p.K = K; p.res_rows = res_rows; p.eps = eps;
    hipStream_t s = (hipStream_t)stream;
    static int cus = 0;
    if (!cus) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount; if (cus <= 0) cus = 256; }
    const int tiles = M / BM;
    PmTimer tm(FAM_GEMM, s);
    hipLaunchKernelGGL(gemm_rowln_kernel, dim3(tiles < cus ? tiles : cus), dim3(THREADS), 0, s, p);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}
