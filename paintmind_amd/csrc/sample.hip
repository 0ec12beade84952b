// MaskGIT sampling tail (reference generate.py:163-179 with helpers :29-46), HBM-bound.
//
// sample_rows: ONE read of each logits row (32 KiB for 8192 fp32 classes).  One wave owns a row,
// the row lives in registers (16 B per lane per load, 1 KiB per wave-instruction, fully
// coalesced).  From that single residency the wave derives: the softmax normaliser (max, sum exp),
// the top-k candidates (k rounds of a wave-wide lexicographic arg-max -- no sort, no scatter of a
// -inf tensor as the reference does at :33-37), the gumbel-perturbed arg-max among the k candidates
// (noise is only needed at those k positions: every other position is -inf in the reference), the
// confidence of the unfiltered softmax at the sampled id, and the merge into the masked positions.
//
// sample_tiles (round 5, top-k <= 8 and V a multiple of 64: every launch of the decode loop): the same step from the softmax
// statistics of the row's 64-column blocks -- (max, sum of exp) per block, 8 bytes, left behind by the logits GEMM's epilogue
// (gemm_common.h, GemmParams::block_stats) -- and the k blocks with the largest maxima, which contain the k largest elements:
// 1 KiB + k x 256 B per row instead of 32 KiB.  Without statistics the kernel derives them from the stored row by the SAME
// arithmetic (common.h softmax_block_stat), so which kernel produced them never shows in the result.
//
// remask: per image, the num_mask highest scores get the mask id (generate.py:175-179); bitonic sort
// of 64-bit (score, reversed index) keys gives the exact (score desc, index asc) order (round 5: the keys stay in registers,
// remask_reg_kernel; the all-LDS sort is kept behind PMHIP_REMASK_REG=0).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int THREADS = 256;

struct Cand { float v; int i; };

// total order used everywhere: larger value first, then smaller index
__device__ __forceinline__ bool before(float av, int ai, float bv, int bi) {
    return (av > bv) || (av == bv && ai < bi);
}

// wave-wide arg-max under `before`: a butterfly on the VALU only (DPP inside a row of 16 lanes, then the row / half swaps;
// nothing goes through the LDS pipe -- DESIGN.md section 4a).  Every lane ends with the same winner: the order is total.
__device__ __forceinline__ Cand wave_best(Cand c) {
    const int lane = threadIdx.x & 63;
#define PM_STEP(OV, OI) { const float ov = (OV); const int oi = (OI); if (before(ov, oi, c.v, c.i)) { c.v = ov; c.i = oi; } }
    PM_STEP(dpp_mov<0xB1>(c.v), dpp_mov<0xB1>(c.i))            // lane ^ 1
    PM_STEP(dpp_mov<0x4E>(c.v), dpp_mov<0x4E>(c.i))            // lane ^ 2
    PM_STEP(dpp_mov<0x141>(c.v), dpp_mov<0x141>(c.i))          // the other quad of the half row (quads are uniform now)
    PM_STEP(dpp_mov<0x140>(c.v), dpp_mov<0x140>(c.i))          // the other half row
    PM_STEP(__uint_as_float(other16(__float_as_uint(c.v), lane)), (int)other16((unsigned)c.i, lane))
    PM_STEP(__uint_as_float(other32(__float_as_uint(c.v), lane)), (int)other32((unsigned)c.i, lane))
#undef PM_STEP
    return c;
}

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

// log(t.clamp(min=1e-20)) twice, negated (generate.py:29-30,40-42)
__device__ __forceinline__ float gumbel_from_uniform(float u) {
    const float inner = -logf(fmaxf(u, 1e-20f));
    return -logf(fmaxf(inner, 1e-20f));
}

// Row layout in registers: lane l holds float4 group g (g = 0..NV4-1) = columns (g*64 + l)*4 .. +3,
// so a lane's columns increase with g and groups are disjoint contiguous column ranges: ordering
// candidates by (value desc, first column of their group asc) equals (value desc, column asc).
template <int NV4>
__global__ __launch_bounds__(THREADS) void sample_rows_kernel(
    const float* __restrict__ logits, int ldl, const int64_t* __restrict__ ids_in, int64_t mask_id, int topk,
    float temperature, const float* __restrict__ noise, uint64_t seed, uint32_t step, uint64_t row_base,
    int64_t* __restrict__ pred_out, int64_t* __restrict__ ids_out, float* __restrict__ score_out, int M, int V,
    const PmGenParams* __restrict__ gp) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    if (row >= M) return;                                  // whole wave exits together
    if (gp) {                                              // graph replay: per-call scalars live in device memory
        temperature = gp->temps[step];
        seed = gp->seed;
        row_base = gp->row_base;
    }
    const float* lrow = logits + (size_t)row * ldl;

    float4 x[NV4];
#pragma unroll
    for (int g = 0; g < NV4; ++g) {
        const int col = (g * 64 + lane) * 4;
        x[g] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (col < V) x[g] = *reinterpret_cast<const float4*>(lrow + col);
    }
    // ---- softmax normaliser of the UNfiltered row (generate.py:170)
    float mx = -INFINITY;
#pragma unroll
    for (int g = 0; g < NV4; ++g) mx = fmaxf(mx, fmaxf(fmaxf(x[g].x, x[g].y), fmaxf(x[g].z, x[g].w)));
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int g = 0; g < NV4; ++g)
        se += (__expf(x[g].x - mx) + __expf(x[g].y - mx)) + (__expf(x[g].z - mx) + __expf(x[g].w - mx));
    se = wave_sum(se);

    // ---- top-k: k rounds of {per-lane best group, wave arg-max, winner removes its element}
    Cand mine{-INFINITY, 0x7fffffff};                      // candidate r is kept by lane r
#pragma unroll 1
    for (int r = 0; r < topk; ++r) {
        float bv = -INFINITY;
        int bg = 0;
#pragma unroll
        for (int g = 0; g < NV4; ++g) {
            const float m4 = fmaxf(fmaxf(x[g].x, x[g].y), fmaxf(x[g].z, x[g].w));
            if (m4 > bv) { bv = m4; bg = g; }              // strict: the first (lowest-column) group wins ties
        }
        Cand c{bv, (bg * 64 + lane) * 4};
        c = wave_best(c);
        // the owner lane finds the element inside the winning group and retires it
        const int wg = __builtin_amdgcn_readfirstlane(c.i >> 8);          // c.i = (g*64+lane)*4 -> g = c.i / 256
        const int wl = (c.i >> 2) & 63;
        int col = c.i;
#pragma unroll
        for (int g = 0; g < NV4; ++g) {
            if (g == wg) {                                                // wave-uniform branch
                if (lane == wl) {
                    if (x[g].x == c.v) { x[g].x = -INFINITY; col = c.i; }
                    else if (x[g].y == c.v) { x[g].y = -INFINITY; col = c.i + 1; }
                    else if (x[g].z == c.v) { x[g].z = -INFINITY; col = c.i + 2; }
                    else { x[g].w = -INFINITY; col = c.i + 3; }
                }
            }
        }
        col = __builtin_amdgcn_readlane(col, wl);                         // wl is wave-uniform
        if (lane == r) { mine.v = c.v; mine.i = col; }
    }
    // ---- gumbel arg-max among the candidates (lane r evaluates candidate r)
    Cand pert{-INFINITY, 0x7fffffff};
    if (lane < topk && mine.i < V) {
        float u;
        if (noise) {
            u = noise[(size_t)row * V + mine.i];
        } else {
            const uint64_t grow = row_base + (uint64_t)row;
            const uint4 rnd = philox4x32_10(make_uint4((uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)mine.i, step),
                                            make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
            u = (float)(rnd.x >> 8) * (1.0f / 16777216.0f);
        }
        pert.v = mine.v / fmaxf(temperature, 1e-10f) + gumbel_from_uniform(u);
        pert.i = mine.i;
    }
    const Cand win = wave_best(pert);
    const unsigned long long owner = __ballot(lane < topk && mine.i == win.i);
    const int src = owner ? __ffsll((long long)owner) - 1 : 0;
    const float raw = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(mine.v), src));   // src is wave-uniform

    if (lane == 0) {
        const int64_t pred = win.i;
        const int64_t cur = ids_in[row];
        const bool is_mask = (cur == mask_id);
        const float p = expf(raw - mx) / se;
        if (pred_out) pred_out[row] = pred;
        ids_out[row] = is_mask ? pred : cur;
        if (score_out) score_out[row] = is_mask ? (1.0f - p) : -1e5f;
    }
}


// ---- the step from block statistics -----------------------------------------------------------------------------------------
constexpr int KT_MAX = 8;                                  // top-k served by sample_tiles_kernel

// NB2 = blocks per lane (block b is kept by lane b % 64).  DENSE: no statistics were handed in -- one pass over the stored row
// computes them (lane l, group g: columns (g*64 + l)*4 .. +3, i.e. block g*4 + l/16 in the layout softmax_block_stat defines).
template <int NB2, bool DENSE>
__global__ __launch_bounds__(THREADS) void sample_tiles_kernel(
    const float* __restrict__ logits, int ldl, const float2* __restrict__ stats, const int64_t* __restrict__ ids_in, int64_t mask_id,
    int topk, float temperature, const float* __restrict__ noise, uint64_t seed, uint32_t step, uint64_t row_base,
    int64_t* __restrict__ pred_out, int64_t* __restrict__ ids_out, float* __restrict__ score_out, int M, int V,
    const PmGenParams* __restrict__ gp) {
    __shared__ float2 sh[DENSE ? THREADS / 64 : 1][DENSE ? NB2 * 64 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * (THREADS / 64) + wave;
    if (row >= M) return;                                  // whole wave exits together (no workgroup barrier below)
    if (gp) {
        temperature = gp->temps[step];
        seed = gp->seed;
        row_base = gp->row_base;
    }
    const float* lrow = logits + (size_t)row * ldl;
    const int nblk = V >> 6;

    float2 st[NB2];
    if constexpr (DENSE) {
        const int ngroups = (V + 255) >> 8;
        for (int g0 = 0; g0 < ngroups; g0 += 8) {          // eight 1-KiB loads in flight per wave
            float4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int col = ((g0 + u) * 64 + lane) * 4;
                x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (g0 + u < ngroups && col < V) x[u] = *reinterpret_cast<const float4*>(lrow + col);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int col = ((g0 + u) * 64 + lane) * 4;
                const float2 b = softmax_block_stat(x[u].x, x[u].y, x[u].z, x[u].w);
                if (g0 + u < ngroups && col < V && (lane & 15) == 0) sh[wave][(g0 + u) * 4 + (lane >> 4)] = b;
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < NB2; ++j) {
            const int b = lane + 64 * j;
            st[j] = b < nblk ? sh[wave][b] : make_float2(-INFINITY, 0.f);
        }
    } else {
        const float2* srow = stats + (size_t)row * nblk;
#pragma unroll
        for (int j = 0; j < NB2; ++j) {
            const int b = lane + 64 * j;
            st[j] = b < nblk ? srow[b] : make_float2(-INFINITY, 0.f);
        }
    }
    // ---- softmax normaliser of the UNfiltered row (generate.py:170) from the blocks: max M, S = sum_b s_b 2^((m_b - M) log2 e)
    float mx = st[0].x;
#pragma unroll
    for (int j = 1; j < NB2; ++j) mx = fmaxf(mx, st[j].x);
    mx = wave_max(mx);
    const float nmx = __fmul_rn(mx, -PM_LOG2E);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < NB2; ++j) se = __fmaf_rn(st[j].y, __builtin_amdgcn_exp2f(__fmaf_rn(st[j].x, PM_LOG2E, nmx)), se);
    se = wave_sum(se);

    // ---- the k blocks with the largest maxima, (max desc, block asc): they hold the k first elements of (value desc, column asc)
    // (an element outside them has k blocks in front of its own, each with an element that comes before it)
    int tile[KT_MAX];
    float x[KT_MAX];
#pragma unroll
    for (int r = 0; r < KT_MAX; ++r) {
        tile[r] = -1;
        x[r] = -INFINITY;
        if (r < topk) {                                    // wave-uniform
            float bv = -INFINITY;
            int bj = 0;
#pragma unroll
            for (int j = 0; j < NB2; ++j)
                if (st[j].x > bv) { bv = st[j].x; bj = j; }   // strict: the lowest block wins ties
            Cand c{bv, lane + 64 * bj};
            c = wave_best(c);
            if (c.v > -INFINITY) {                         // (fewer blocks than k: the remaining rounds find nothing)
                tile[r] = __builtin_amdgcn_readfirstlane(c.i);
#pragma unroll
                for (int j = 0; j < NB2; ++j)
                    if (lane + 64 * j == c.i) st[j].x = -INFINITY;
                x[r] = lrow[tile[r] * 64 + lane];
            }
        }
    }
    // ---- top-k elements of the k x 64 values: k rounds of {per-lane best, wave arg-max, winner retires its element}
    Cand mine{-INFINITY, 0x7fffffff};                      // candidate r is kept by lane r
#pragma unroll
    for (int r = 0; r < KT_MAX; ++r) {
        if (r < topk) {
            Cand c{-INFINITY, 0x7fffffff};
#pragma unroll
            for (int q = 0; q < KT_MAX; ++q) {
                const int ci = tile[q] * 64 + lane;
                if (tile[q] >= 0 && before(x[q], ci, c.v, c.i)) { c.v = x[q]; c.i = ci; }
            }
            c = wave_best(c);
#pragma unroll
            for (int q = 0; q < KT_MAX; ++q)
                if (tile[q] >= 0 && tile[q] * 64 + lane == c.i) x[q] = -INFINITY;
            if (lane == r) mine = c;
        }
    }
    // ---- gumbel arg-max among the candidates (lane r evaluates candidate r)
    Cand pert{-INFINITY, 0x7fffffff};
    if (lane < topk && mine.i < V) {
        float u;
        if (noise) {
            u = noise[(size_t)row * V + mine.i];
        } else {
            const uint64_t grow = row_base + (uint64_t)row;
            const uint4 rnd = philox4x32_10(make_uint4((uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)mine.i, step),
                                            make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
            u = (float)(rnd.x >> 8) * (1.0f / 16777216.0f);
        }
        pert.v = mine.v / fmaxf(temperature, 1e-10f) + gumbel_from_uniform(u);
        pert.i = mine.i;
    }
    const Cand win = wave_best(pert);
    const unsigned long long owner = __ballot(lane < topk && mine.i == win.i);
    const int src = owner ? __ffsll((long long)owner) - 1 : 0;
    const float raw = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(mine.v), src));   // src is wave-uniform

    if (lane == 0) {
        const int64_t pred = win.i;
        const int64_t cur = ids_in[row];
        const bool is_mask = (cur == mask_id);
        const float p = __fdiv_rn(__builtin_amdgcn_exp2f(__fmaf_rn(raw, PM_LOG2E, nmx)), se);
        if (pred_out) pred_out[row] = pred;
        ids_out[row] = is_mask ? pred : cur;
        if (score_out) score_out[row] = is_mask ? (1.0f - p) : -1e5f;
    }
}

__device__ __forceinline__ uint32_t orderable(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(THREADS) void remask_kernel(int64_t* __restrict__ ids, const float* __restrict__ scores,
                                                         int num_mask, int64_t mask_id, int N, int Npow2,
                                                         const PmGenParams* __restrict__ gp, int step) {
    if (gp) num_mask = gp->nmask[step];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* key = reinterpret_cast<unsigned long long*>(smem);
    const int b = blockIdx.x;
    const float* sc = scores + (size_t)b * N;
    for (int i = threadIdx.x; i < Npow2; i += THREADS)
        key[i] = i < N ? (((unsigned long long)orderable(sc[i]) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)i)) : 0ull;
    __syncthreads();
    // bitonic sort, descending
    for (int k = 2; k <= Npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < Npow2; i += THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = key[i], c = key[ixj];
                    const bool desc = ((i & k) == 0);
                    if (desc ? (a < c) : (a > c)) { key[i] = c; key[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    const int nm = num_mask < 1 ? 1 : (num_mask > N ? N : num_mask);
    const unsigned long long thr = key[nm - 1];
    for (int i = threadIdx.x; i < N; i += THREADS) {
        const unsigned long long k = ((unsigned long long)orderable(sc[i]) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)i);
        if (k >= thr) ids[(size_t)b * N + i] = mask_id;
    }
}


// Round 5: the same selection with the sort in registers.  Thread t keeps the E consecutive elements t*E .. t*E+E-1; a bitonic
// compare-exchange with distance j is in-thread for j < E, a wave shuffle for j < 64*E and goes through LDS (two barriers) only
// beyond that: 3 of the 55 stages at N = 1024 (the all-LDS kernel above: one barrier per stage, 40 us per launch on B <= 64
// workgroups -- latency, not work).  Same keys, same total order, same threshold test.
template <int E>
__global__ __launch_bounds__(THREADS) void remask_reg_kernel(int64_t* __restrict__ ids, const float* __restrict__ scores,
                                                             int num_mask, int64_t mask_id, int N, const PmGenParams* __restrict__ gp,
                                                             int step) {
    constexpr int NP = THREADS * E;
    __shared__ unsigned long long xs[NP];
    if (gp) num_mask = gp->nmask[step];
    const int tid = threadIdx.x, base = tid * E;
    const float* sc = scores + (size_t)blockIdx.x * N;
    unsigned long long key[E], mine[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = base + e;
        key[e] = i < N ? (((unsigned long long)orderable(sc[i]) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)i)) : 0ull;
        mine[e] = key[e];
    }
#pragma unroll
    for (int k = 2; k <= NP; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < E) {                                   // both elements in this thread
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & j) == 0) {
                        const bool desc = ((base + e) & k) == 0;
                        const unsigned long long a = key[e], b = key[e | j];
                        const bool swap = desc ? (a < b) : (a > b);
                        key[e] = swap ? b : a;
                        key[e | j] = swap ? a : b;
                    }
                }
            } else {
                unsigned long long other[E];
                if (j < 64 * E) {                          // the partner thread is in this wave
#pragma unroll
                    for (int e = 0; e < E; ++e) other[e] = __shfl_xor(key[e], j / E, 64);
                } else {
#pragma unroll
                    for (int e = 0; e < E; ++e) xs[base + e] = key[e];
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < E; ++e) other[e] = xs[(base + e) ^ j];
                    __syncthreads();
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = base + e;
                    const bool want_max = ((i & j) == 0) == ((i & k) == 0);   // descending block: the lower index keeps the larger key
                    const unsigned long long a = key[e], b = other[e];
                    key[e] = want_max ? (a > b ? a : b) : (a < b ? a : b);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) xs[base + e] = key[e];
    __syncthreads();
    const int nm = num_mask < 1 ? 1 : (num_mask > N ? N : num_mask);
    const unsigned long long thr = xs[nm - 1];
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (base + e < N && mine[e] >= thr) ids[(size_t)blockIdx.x * N + base + e] = mask_id;
}

}  // namespace

static int sample_env_int(const char* name, int dflt) { return pm_dev_knob(name, dflt); }     // development builds only (common.h)

int pm_sample_rows(const float* logits, int ldl, const float* block_stats, const int64_t* ids_in, int64_t mask_id, int topk,
                   float temperature, const float* noise, uint64_t seed, uint32_t step, uint64_t row_base, int64_t* pred_out,
                   int64_t* ids_out, float* score_out, int M, int V, const PmGenParams* gp, pmhip_stream stream) {
    PM_REQUIRE(logits && ids_in && ids_out, "sample_rows: null pointer");
    PM_REQUIRE(M > 0 && V > 0 && V % 4 == 0 && ldl % 4 == 0 && ldl >= V, "sample_rows: bad shape M=%d V=%d ldl=%d", M, V, ldl);
    PM_REQUIRE(topk >= 1 && topk <= 64 && topk <= V, "sample_rows: topk=%d must be in [1, min(64,V)]", topk);
    PM_REQUIRE(V <= 16384, "sample_rows: V=%d > 16384 unsupported", V);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(ceil_div(M, THREADS / 64)), block(THREADS);
    PmTimer tm(FAM_SAMPLE, s);
    // Which kernel runs depends on (V, topk) ONLY -- never on whether statistics were handed in: the two differ in the last bits
    // of the confidence (a different summation order of the softmax denominator).
    static const int g_tiles = sample_env_int("PMHIP_SAMPLE_TILES", 1);     // 0: the one-read row kernel everywhere (A/B)
    if (g_tiles && topk <= KT_MAX && V % 64 == 0) {
        const float2* st = reinterpret_cast<const float2*>(block_stats);
#define PM_TILES(NB2)                                                                                                         \
    do {                                                                                                                      \
        if (st) hipLaunchKernelGGL((sample_tiles_kernel<NB2, false>), grid, block, 0, s, logits, ldl, st, ids_in, mask_id, topk, \
                                   temperature, noise, seed, step, row_base, pred_out, ids_out, score_out, M, V, gp);          \
        else hipLaunchKernelGGL((sample_tiles_kernel<NB2, true>), grid, block, 0, s, logits, ldl, st, ids_in, mask_id, topk,   \
                                temperature, noise, seed, step, row_base, pred_out, ids_out, score_out, M, V, gp);             \
    } while (0)
        if (V <= 4096) PM_TILES(1);
        else if (V <= 8192) PM_TILES(2);
        else PM_TILES(4);
#undef PM_TILES
        PM_HIP(hipGetLastError());
        return PMHIP_OK;
    }
#define PM_SAMPLE(NV4)                                                                                              \
    hipLaunchKernelGGL((sample_rows_kernel<NV4>), grid, block, 0, s, logits, ldl, ids_in, mask_id, topk, temperature, \
                       noise, seed, step, row_base, pred_out, ids_out, score_out, M, V, gp)
    if (V <= 256) PM_SAMPLE(1);
    else if (V <= 1024) PM_SAMPLE(4);
    else if (V <= 8192) PM_SAMPLE(32);
    else PM_SAMPLE(64);
#undef PM_SAMPLE
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

extern "C" int pmhip_sample_rows(const float* logits, int ldl, const int64_t* ids_in, int64_t mask_id, int topk,
                                 float temperature, const float* noise, uint64_t seed, uint32_t step,
                                 uint64_t row_base, int64_t* pred_out, int64_t* ids_out, float* score_out, int M,
                                 int V, pmhip_stream stream) {
    return pm_sample_rows(logits, ldl, nullptr, ids_in, mask_id, topk, temperature, noise, seed, step, row_base, pred_out, ids_out,
                          score_out, M, V, nullptr, stream);
}

// the same step with the block statistics pmhip_gemm_softmax_stats (or pmhip_guidance_combine_stats) left behind: [M][V/64][2]
extern "C" int pmhip_sample_rows_stats(const float* logits, int ldl, const float* block_stats, const int64_t* ids_in, int64_t mask_id,
                                       int topk, float temperature, const float* noise, uint64_t seed, uint32_t step,
                                       uint64_t row_base, int64_t* pred_out, int64_t* ids_out, float* score_out, int M, int V,
                                       pmhip_stream stream) {
    PM_REQUIRE(block_stats, "sample_rows_stats: null statistics");
    PM_REQUIRE(V % 64 == 0, "sample_rows_stats: V=%d must be a multiple of 64", V);
    return pm_sample_rows(logits, ldl, block_stats, ids_in, mask_id, topk, temperature, noise, seed, step, row_base, pred_out, ids_out,
                          score_out, M, V, nullptr, stream);
}

int pm_remask(int64_t* ids, const float* scores, int num_mask, int64_t mask_id, int B, int N, const PmGenParams* gp, int step,
              pmhip_stream stream) {
    PM_REQUIRE(ids && scores, "remask: null pointer");
    PM_REQUIRE(B > 0 && N > 0 && N <= 4096, "remask: bad shape B=%d N=%d (N <= 4096)", B, N);
    int np2 = 1;
    while (np2 < N) np2 <<= 1;
    hipStream_t s = (hipStream_t)stream;
    PmTimer tm(FAM_SAMPLE, s);
    static const int g_reg = sample_env_int("PMHIP_REMASK_REG", 1);      // 0: the all-LDS sort (A/B)
#define PM_REMASK(E) hipLaunchKernelGGL(remask_reg_kernel<E>, dim3(B), dim3(THREADS), 0, s, ids, scores, num_mask, mask_id, N, gp, step)
    if (!g_reg) hipLaunchKernelGGL(remask_kernel, dim3(B), dim3(THREADS), (size_t)np2 * 8, s, ids, scores, num_mask, mask_id, N, np2, gp, step);
    else if (np2 <= 256) PM_REMASK(1);
    else if (np2 <= 512) PM_REMASK(2);
    else if (np2 <= 1024) PM_REMASK(4);
    else if (np2 <= 2048) PM_REMASK(8);
    else PM_REMASK(16);
#undef PM_REMASK
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

extern "C" int pmhip_remask(int64_t* ids, const float* scores, int num_mask, int64_t mask_id, int B, int N,
                            pmhip_stream stream) {
    return pm_remask(ids, scores, num_mask, mask_id, B, N, nullptr, 0, stream);
}
