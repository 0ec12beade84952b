// L2-normalised nearest-neighbour vector quantiser (reference stage1/quantize.py:18-44).
//
// The arithmetic ORDER is part of the contract, because idx must be bit-exact against the oracle
// (oracle/vq_ref.c restates exactly this sequence):
//   ss  = fmaf(z[k], z[k], ss)            k = 0..E-1, ss starts at +0
//   zn  = z / fmaxf(sqrtf(ss), 1e-12f)    IEEE sqrt and divide (hipcc default: correctly rounded)
//   sz  = fmaf(zn[k], zn[k], sz)          k = 0..E-1
//   dot = fmaf(zn[k], en[j][k], dot)      k = 0..E-1
//   d_j = (sz + sq[j]) - 2*dot            (the reference's "(z^2 + e^2) - 2 z.e", quantize.py:24-26)
//   idx = first j with the smallest d_j   (torch.argmin = first occurrence)
// The dot products run on the exact-f32 matrix core: v_mfma_f32_16x16x4_f32 is bit for bit a k-ordered fmaf chain
// (MI355X_MICROARCH.md), so E/4 chained MFMAs per 16 rows x 16 codes reproduce the sequence above exactly, at the
// rate of the f32 vector ALU's peak but with both operands in registers: a wave keeps its 64 normalised rows as row
// operands (E registers) and streams the codebook through LDS as column operands, one ds_read_b32 per lane per 4 MFMAs.
// (The round-1/2 kernel ran the same chains as VALU fmaf with one LDS broadcast read per 4 fmaf: 59 TFLOP/s, 582 us at
// 65 536 rows x 8192 codes; this one 95 TFLOP/s, 340 us, matrix pipe 0.65 busy: the waves of a SIMD fall into step and
// idle the pipe through their distance / first-minimum VALU phases.)
// The codebook is split 4-ways across blockIdx.y to fill the chip (M/256 x 4 workgroups); a finishing kernel merges
// the splits in index order.
#include "common.h"

namespace {

constexpr int THREADS = 256;
constexpr int TC = 128;   // codes per LDS tile

inline int vq_splits(int V) { return V >= 2048 ? 4 : 1; }

__global__ __launch_bounds__(THREADS) void vq_prepare_kernel(const float* __restrict__ w, float* __restrict__ en,
                                                             float* __restrict__ sq, int V, int E) {
    const int j = blockIdx.x * THREADS + threadIdx.x;
    if (j >= V) return;
    const float* row = w + (size_t)j * E;
    float ss = 0.f;
    for (int k = 0; k < E; ++k) ss = fmaf(row[k], row[k], ss);
    const float den = fmaxf(sqrtf(ss), 1e-12f);
    float s2 = 0.f;
    for (int k = 0; k < E; ++k) {
        const float v = row[k] / den;
        en[(size_t)j * E + k] = v;
        s2 = fmaf(v, v, s2);
    }
    sq[j] = s2;
}

template <int E>
__device__ __forceinline__ float normalize_row(const float* __restrict__ zrow, float (&zn)[E]) {
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < E; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(zrow + k);
        zn[k] = v.x; zn[k + 1] = v.y; zn[k + 2] = v.z; zn[k + 3] = v.w;
    }
#pragma unroll
    for (int k = 0; k < E; ++k) ss = fmaf(zn[k], zn[k], ss);
    const float den = fmaxf(sqrtf(ss), 1e-12f);
    float sz = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        zn[k] = zn[k] / den;
        sz = fmaf(zn[k], zn[k], sz);
    }
    return sz;
}

// (distance, index) minimum over the 16 lanes of a DPP row (the 16 codes of a tile column group): smaller distance wins,
// equal distances keep the smaller index -- together with each lane's strict '<' scan this is the first minimum
template <int CTRL> __device__ __forceinline__ void row_min_step(float& d, int& i) {
    const float od = dpp_mov<CTRL>(d);
    const int oi = dpp_mov<CTRL>(i);
    const bool take = od < d || (od == d && oi < i);
    d = take ? od : d;
    i = take ? oi : i;
}

template <int E>
__global__ __launch_bounds__(THREADS, 3) void vq_scan_kernel(const float* __restrict__ z, const float* __restrict__ en,
                                                             const float* __restrict__ sq, float* __restrict__ best_d,
                                                             int* __restrict__ best_i, int M, int V, int codes_per_split) {
    constexpr int KG = E / 4;                    // k-groups of 4 = MFMAs per chain
    constexpr int TCP = TC + 16;                 // k-major code tile [E][TCP]: the +16 floats put lane groups g, g+1 on disjoint banks
    constexpr int ZP = E + 1;                    // row-major z tile [64][ZP] per wave (conflict-free column reads)
    // ONE LDS array, two lives: first the waves' normalised rows on their way into the operand layout (64 x ZP + 64 sz per
    // wave), then -- behind a barrier -- two buffers of a code tile + its squared norms
    constexpr int ZL_FLOATS = 4 * 64 * ZP + 4 * 64, TILE_FLOATS = E * TCP + TC;
    __shared__ __attribute__((aligned(16))) float lds[ZL_FLOATS > 2 * TILE_FLOATS ? ZL_FLOATS : 2 * TILE_FLOATS];
    float* tile = lds;
    float* zl = lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int m = blockIdx.x * THREADS + threadIdx.x;
    const int mm = m < M ? m : M - 1;
    float* zw = zl + wave * 64 * ZP;
    float* szw = zl + 4 * 64 * ZP + wave * 64;
    {
        float zn[E];
        const float sz = normalize_row<E>(z + (size_t)mm * E, zn);          // thread t: row t of the wave, exact sequential order
#pragma unroll
        for (int k = 0; k < E; ++k) zw[lane * ZP + k] = zn[k];
        szw[lane] = sz;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // row operand of v_mfma_f32_16x16x4_f32: lane (l15, g) holds A[row l15][k = g]; 4 row tiles x KG k-groups
    float a[4][KG];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) a[rt][kg] = zw[(rt * 16 + l15) * ZP + kg * 4 + g];
    // the accumulator layout puts row 4 g + r of a tile into register r: the sz of the rows this lane compares
    float szr[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) szr[rt][r] = szw[rt * 16 + 4 * g + r];

    const int v0 = blockIdx.y * codes_per_split;
    const int v1 = min(V, v0 + codes_per_split);
    float bd[4][4];
    int bi[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { bd[rt][r] = INFINITY; bi[rt][r] = v0; }

    // Code tiles go global [code][k] -> registers -> LDS [k][code], double-buffered: the loads of tile t+1 are issued before
    // tile t is multiplied and written behind it, one barrier per tile.  Thread -> (code, k) is chosen so that a wave's
    // transposing stores hit 32 different banks (consecutive codes, same k).
    constexpr int FILL = TC * E / 4 / THREADS;   // float4 per thread per tile
    float4 pre[FILL];
    float presq = INFINITY;
    auto fetch = [&](int t0) {
#pragma unroll
        for (int f = 0; f < FILL; ++f) {
            const int i = f * THREADS + threadIdx.x, code = i % TC, k = (i / TC) * 4;
            pre[f] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t0 + code < v1) pre[f] = *reinterpret_cast<const float4*>(en + (size_t)(t0 + code) * E + k);
        }
        if (threadIdx.x < TC) presq = (t0 + threadIdx.x < v1) ? sq[t0 + threadIdx.x] : INFINITY;
    };
    auto commit = [&](int buf) {
        float* tl = tile + buf * TILE_FLOATS;
#pragma unroll
        for (int f = 0; f < FILL; ++f) {
            const int i = f * THREADS + threadIdx.x, code = i % TC, k = (i / TC) * 4;
            tl[(k + 0) * TCP + code] = pre[f].x; tl[(k + 1) * TCP + code] = pre[f].y;
            tl[(k + 2) * TCP + code] = pre[f].z; tl[(k + 3) * TCP + code] = pre[f].w;
        }
        if (threadIdx.x < TC) tl[E * TCP + threadIdx.x] = presq;
    };
    fetch(v0);
    __syncthreads();                             // every wave has its operands out of the z staging area
    commit(0);
    int buf = 0;
    for (int t0 = v0; t0 < v1; t0 += TC, buf ^= 1) {
        __syncthreads();                         // tile `buf` is complete; nobody reads buffer buf^1 any more
        const bool more = t0 + TC < v1;
        if (more) fetch(t0 + TC);
        const float* tl = tile + buf * TILE_FLOATS;
        const float* tq = tl + E * TCP;
#pragma unroll 2
        for (int ct = 0; ct < TC / 16; ++ct) {
            // column operand: lane (l15, g) holds B[k = g][code l15] of each k-group
            float bcol[KG];
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) bcol[kg] = tl[(kg * 4 + g) * TCP + ct * 16 + l15];
            const float sqc = tq[ct * 16 + l15];
            f32x4_t acc[4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kg = 0; kg < KG; ++kg)                                   // k ascending: the fmaf chain's order
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)                                // 4 independent chains hide the MFMA latency
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][kg], bcol[kg], acc[rt], 0, 0, 0);
            const int code = t0 + ct * 16 + l15;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dist = (szr[rt][r] + sqc) - 2.0f * acc[rt][r];
                    if (dist < bd[rt][r]) { bd[rt][r] = dist; bi[rt][r] = code; }   // codes ascend per lane: first minimum
                }
        }
        if (more) commit(buf ^ 1);
    }
    // the 16 lanes of a row group hold disjoint code residues of the same rows
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            row_min_step<0xB1>(bd[rt][r], bi[rt][r]);      // lane ^ 1
            row_min_step<0x4E>(bd[rt][r], bi[rt][r]);      // lane ^ 2
            row_min_step<0x141>(bd[rt][r], bi[rt][r]);     // other quad of the half row
            row_min_step<0x140>(bd[rt][r], bi[rt][r]);     // other half row
        }
    if (l15 == 0) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = blockIdx.x * THREADS + wave * 64 + rt * 16 + 4 * g + r;
                if (row < M) {
                    best_d[(size_t)blockIdx.y * M + row] = bd[rt][r];
                    best_i[(size_t)blockIdx.y * M + row] = bi[rt][r];
                }
            }
    }
}

// merge the splits (ascending code ranges, strict '<' keeps the first minimum), emit idx, the
// straight-through value z + (zq - z) (quantize.py:36) and per-block partial sums of (zq - z)^2
template <int E>
__global__ __launch_bounds__(THREADS) void vq_finish_kernel(const float* __restrict__ z, const float* __restrict__ en,
                                                            const float* __restrict__ best_d, const int* __restrict__ best_i,
                                                            int splits, float* __restrict__ z_out,
                                                            int64_t* __restrict__ idx_out, float* __restrict__ partial, int M) {
    __shared__ float red[THREADS / 64];
    const int m = blockIdx.x * THREADS + threadIdx.x;
    float acc = 0.f;
    if (m < M) {
        float zn[E];
        normalize_row<E>(z + (size_t)m * E, zn);
        float bd = best_d[m];
        int bi = best_i[m];
        for (int s = 1; s < splits; ++s) {
            const float d = best_d[(size_t)s * M + m];
            if (d < bd) { bd = d; bi = best_i[(size_t)s * M + m]; }
        }
        idx_out[m] = bi;
        const float* q = en + (size_t)bi * E;
#pragma unroll
        for (int k = 0; k < E; k += 4) {
            const float4 qv = *reinterpret_cast<const float4*>(q + k);
            const float d0 = qv.x - zn[k], d1 = qv.y - zn[k + 1], d2 = qv.z - zn[k + 2], d3 = qv.w - zn[k + 3];
            acc += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            if (z_out) store4(z_out + (size_t)m * E + k, zn[k] + d0, zn[k + 1] + d1, zn[k + 2] + d2, zn[k + 3] + d3);
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// loss = beta*mean + mean (quantize.py:33), partial sums added in a fixed order
__global__ __launch_bounds__(THREADS) void vq_loss_kernel(const float* __restrict__ partial, int nblocks, float beta,
                                                          float count, float* __restrict__ loss_out) {
    __shared__ float red[THREADS / 64];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += THREADS) acc += partial[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / count;
        loss_out[0] = beta * mean + mean;
    }
}

template <int E>
int run_quantize(const float* z, const float* en, const float* sq, float beta, float* z_out, int64_t* idx_out,
                 float* loss_out, void* scratch, int M, int V, hipStream_t s) {
    const int splits = vq_splits(V);
    int cps = ceil_div(V, splits);
    cps = ceil_div(cps, TC) * TC;
    const int mblocks = ceil_div(M, THREADS);
    float* best_d = reinterpret_cast<float*>(scratch);
    int* best_i = reinterpret_cast<int*>(best_d + (size_t)splits * M);
    float* partial = reinterpret_cast<float*>(best_i + (size_t)splits * M);
    PmTimer tm(FAM_VQ, s);
    hipLaunchKernelGGL((vq_scan_kernel<E>), dim3(mblocks, splits), dim3(THREADS), 0, s, z, en, sq, best_d, best_i, M, V, cps);
    hipLaunchKernelGGL((vq_finish_kernel<E>), dim3(mblocks), dim3(THREADS), 0, s, z, en, best_d, best_i, splits, z_out,
                       idx_out, partial, M);
    if (loss_out)
        hipLaunchKernelGGL(vq_loss_kernel, dim3(1), dim3(THREADS), 0, s, partial, mblocks, beta, (float)M * (float)E, loss_out);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

}  // namespace

extern "C" int pmhip_vq_prepare(const float* codebook, float* en, float* sq, int V, int E, pmhip_stream stream) {
    PM_REQUIRE(codebook && en && sq && V > 0 && E > 0, "vq_prepare: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    PmTimer tm(FAM_VQ, s);
    hipLaunchKernelGGL(vq_prepare_kernel, dim3(ceil_div(V, THREADS)), dim3(THREADS), 0, s, codebook, en, sq, V, E);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

extern "C" size_t pmhip_vq_scratch_bytes(int M, int V) {
    const size_t splits = (size_t)vq_splits(V);
    return splits * (size_t)M * 8 + (size_t)ceil_div(M, THREADS) * 4 + 256;
}

extern "C" int pmhip_vq_quantize(const float* z, const float* en, const float* sq, float beta, float* z_out,
                                 int64_t* idx_out, float* loss_out, void* scratch, int M, int V, int E,
                                 pmhip_stream stream) {
    PM_REQUIRE(z && en && sq && idx_out && scratch, "vq_quantize: null pointer");
    PM_REQUIRE(M > 0 && V > 0, "vq_quantize: empty problem");
    hipStream_t s = (hipStream_t)stream;
    switch (E) {
        case 8: return run_quantize<8>(z, en, sq, beta, z_out, idx_out, loss_out, scratch, M, V, s);
        case 16: return run_quantize<16>(z, en, sq, beta, z_out, idx_out, loss_out, scratch, M, V, s);
        case 32: return run_quantize<32>(z, en, sq, beta, z_out, idx_out, loss_out, scratch, M, V, s);
        case 64: return run_quantize<64>(z, en, sq, beta, z_out, idx_out, loss_out, scratch, M, V, s);
        default: pm_set_error("vq_quantize: embed_dim %d not in {8,16,32,64}", E); return PMHIP_EINVAL;
    }
}
