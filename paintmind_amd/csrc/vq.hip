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
// One thread owns one z row (E floats in registers) and scans codes from an LDS tile that every
// lane reads at the same address (broadcast, conflict-free); two codes are in flight per thread so
// the dependent fmaf chains overlap.  The codebook is split 4-ways across blockIdx.y to fill the
// chip (M/256 x 4 workgroups); a finishing kernel merges the splits in index order.
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

template <int E>
__global__ __launch_bounds__(THREADS) void vq_scan_kernel(const float* __restrict__ z, const float* __restrict__ en,
                                                          const float* __restrict__ sq, float* __restrict__ best_d,
                                                          int* __restrict__ best_i, int M, int V, int codes_per_split) {
    __shared__ __attribute__((aligned(16))) float tile[TC * E];
    __shared__ float tsq[TC];
    const int m = blockIdx.x * THREADS + threadIdx.x;
    const int mm = m < M ? m : M - 1;
    float zn[E];
    const float sz = normalize_row<E>(z + (size_t)mm * E, zn);

    const int v0 = blockIdx.y * codes_per_split;
    const int v1 = min(V, v0 + codes_per_split);
    float bd = INFINITY;
    int bi = v0;
    for (int t0 = v0; t0 < v1; t0 += TC) {
        __syncthreads();
        for (int i = threadIdx.x; i < TC * E / 4; i += THREADS) {
            const int code = t0 + (i * 4) / E;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (code < v1) v = *reinterpret_cast<const float4*>(en + (size_t)t0 * E + i * 4);
            *reinterpret_cast<float4*>(tile + i * 4) = v;
        }
        if (threadIdx.x < TC) tsq[threadIdx.x] = (t0 + threadIdx.x < v1) ? sq[t0 + threadIdx.x] : INFINITY;
        __syncthreads();
#pragma unroll 1
        for (int j = 0; j < TC; j += 2) {
            float d0 = 0.f, d1 = 0.f;
            const float* e0 = tile + j * E;
            const float* e1 = e0 + E;
#pragma unroll
            for (int k = 0; k < E; k += 4) {
                const float4 a = *reinterpret_cast<const float4*>(e0 + k);
                const float4 b = *reinterpret_cast<const float4*>(e1 + k);
                d0 = fmaf(zn[k], a.x, d0); d1 = fmaf(zn[k], b.x, d1);
                d0 = fmaf(zn[k + 1], a.y, d0); d1 = fmaf(zn[k + 1], b.y, d1);
                d0 = fmaf(zn[k + 2], a.z, d0); d1 = fmaf(zn[k + 2], b.z, d1);
                d0 = fmaf(zn[k + 3], a.w, d0); d1 = fmaf(zn[k + 3], b.w, d1);
            }
            const float dist0 = (sz + tsq[j]) - 2.0f * d0;
            const float dist1 = (sz + tsq[j + 1]) - 2.0f * d1;
            if (dist0 < bd) { bd = dist0; bi = t0 + j; }
            if (dist1 < bd) { bd = dist1; bi = t0 + j + 1; }
        }
    }
    if (m < M) {
        best_d[(size_t)blockIdx.y * M + m] = bd;
        best_i[(size_t)blockIdx.y * M + m] = bi;
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
