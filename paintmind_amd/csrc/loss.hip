// Masked-token modelling objective, forward only (reference generate.py:78-146): the per-sample random
// masking of the latent sequence and the label-smoothed cross entropy over the masked positions.
// Both are HBM-bound row kernels.
//
// random_mask: reference :78-110.  The reference sorts one uniform per position, keeps the len_keep
// smallest and replaces the rest by the mask token.  Only the RANK of each position matters, so no
// permutation is materialised: a block owns a sample, the noise row sits in LDS and every position
// counts the positions ordered before it (smaller noise, ties to the smaller index -- the order a
// stable ascending sort produces).
//
// masked_ce: reference :112-125 (F.cross_entropy(label_smoothing=eps, reduction='none'), times the mask,
// summed, divided by the number of masked positions).  ONE read of each logits row, held in registers
// by one wave exactly as in sample.hip; per row
//     loss = (1-eps) * (lse - x[label]) + eps * (lse - mean(x)),   lse = max + log(sum exp(x - max)).
// The final reduction is a single fixed-order block, so the scalar is run-to-run deterministic.
#include "common.h"

namespace {

constexpr int THREADS = 256;

__global__ __launch_bounds__(1024) void random_mask_kernel(const float* __restrict__ z, const float* __restrict__ noise,
                                                            const float* __restrict__ mask_token, int len_keep,
                                                            float* __restrict__ x_out, float* __restrict__ mask_out, int N,
                                                            int E) {
    extern __shared__ float smem[];
    float* nz = smem;                                      // [N] noise of this sample
    float* flag = smem + N;                                // [N] 1 = masked
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < N; i += blockDim.x) nz[i] = noise[(size_t)b * N + i];
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const float mine = nz[i];
        int rank = 0;
        for (int j = 0; j < N; ++j) {                      // LDS broadcast reads
            const float o = nz[j];
            rank += (o < mine) || (o == mine && j < i);
        }
        const float m = rank >= len_keep ? 1.f : 0.f;
        flag[i] = m;
        mask_out[(size_t)b * N + i] = m;
    }
    __syncthreads();
    const size_t base = (size_t)b * N * E;
    for (int idx = threadIdx.x; idx < N * E; idx += blockDim.x) {
        const int row = idx / E, col = idx - row * E;
        x_out[base + idx] = flag[row] != 0.f ? mask_token[col] : z[base + idx];
    }
}

// lane l holds float4 group g = columns (g*64 + l)*4 .. +3 (same residency as sample_rows_kernel)
template <int NV4>
__global__ __launch_bounds__(THREADS) void masked_ce_rows_kernel(const float* __restrict__ logits, int ldl,
                                                                 const int64_t* __restrict__ labels,
                                                                 const float* __restrict__ mask, float eps,
                                                                 float* __restrict__ row_loss, int M, int V) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* lrow = logits + (size_t)row * ldl;
    float4 x[NV4];
#pragma unroll
    for (int g = 0; g < NV4; ++g) {
        const int col = (g * 64 + lane) * 4;
        x[g] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (col < V) x[g] = *reinterpret_cast<const float4*>(lrow + col);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int g = 0; g < NV4; ++g) mx = fmaxf(mx, fmaxf(fmaxf(x[g].x, x[g].y), fmaxf(x[g].z, x[g].w)));
    mx = wave_max(mx);
    float se = 0.f, sx = 0.f;
#pragma unroll
    for (int g = 0; g < NV4; ++g) {
        const int col = (g * 64 + lane) * 4;
        se += (__expf(x[g].x - mx) + __expf(x[g].y - mx)) + (__expf(x[g].z - mx) + __expf(x[g].w - mx));
        if (col < V) sx += (x[g].x + x[g].y) + (x[g].z + x[g].w);
    }
    se = wave_sum(se);
    sx = wave_sum(sx);
    if (lane == 0) {
        const float lse = mx + logf(se);
        const float nll = lse - lrow[labels[row]];
        const float smooth = lse - sx / (float)V;
        row_loss[row] = ((1.f - eps) * nll + eps * smooth) * mask[row];
    }
}

// loss = sum(row_loss) / sum(mask): one block, fixed order (double partials, tree in LDS)
__global__ __launch_bounds__(1024) void masked_ce_reduce_kernel(const float* __restrict__ row_loss,
                                                                 const float* __restrict__ mask, float* __restrict__ out,
                                                                 int M) {
    __shared__ double sl[1024];
    __shared__ double sm[1024];
    double a = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < M; i += 1024) {
        a += (double)row_loss[i];
        c += (double)mask[i];
    }
    sl[threadIdx.x] = a;
    sm[threadIdx.x] = c;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sl[threadIdx.x] += sl[threadIdx.x + s];
            sm[threadIdx.x] += sm[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(sl[0] / sm[0]);     // 0/0 = nan, as in the reference when nothing is masked
}

}  // namespace

extern "C" int pmhip_random_mask(const float* z, const float* noise, const float* mask_token, int len_keep, float* x_out,
                                 float* mask_out, int B, int N, int E, pmhip_stream stream) {
    PM_REQUIRE(z && noise && mask_token && x_out && mask_out, "random_mask: null pointer");
    PM_REQUIRE(B > 0 && N > 0 && E > 0, "random_mask: bad shape B=%d N=%d E=%d", B, N, E);
    PM_REQUIRE(len_keep >= 0 && len_keep <= N, "random_mask: len_keep=%d outside [0, N=%d]", len_keep, N);
    PM_REQUIRE(N <= 16384, "random_mask: N=%d > 16384 (the noise row must fit LDS)", N);
    hipStream_t s = (hipStream_t)stream;
    PmTimer tm(FAM_SAMPLE, s);
    const int threads = N >= 1024 ? 1024 : ceil_div(N, 64) * 64;
    hipLaunchKernelGGL(random_mask_kernel, dim3(B), dim3(threads), (size_t)2 * N * sizeof(float), s, z, noise, mask_token,
                       len_keep, x_out, mask_out, N, E);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

extern "C" int pmhip_masked_ce(const float* logits, int ldl, const int64_t* labels, const float* mask, float label_smoothing,
                               float* row_loss, float* loss_out, int M, int V, pmhip_stream stream) {
    PM_REQUIRE(logits && labels && mask && row_loss && loss_out, "masked_ce: null pointer");
    PM_REQUIRE(M > 0 && V > 0 && V % 4 == 0 && ldl % 4 == 0 && ldl >= V, "masked_ce: bad shape M=%d V=%d ldl=%d", M, V, ldl);
    PM_REQUIRE(V <= 16384, "masked_ce: V=%d > 16384 unsupported", V);
    PM_REQUIRE(label_smoothing >= 0.f && label_smoothing <= 1.f, "masked_ce: label_smoothing=%f outside [0,1]",
               (double)label_smoothing);
    hipStream_t s = (hipStream_t)stream;
    PmTimer tm(FAM_SAMPLE, s);
    dim3 grid(ceil_div(M, THREADS / 64)), block(THREADS);
#define PM_CE(NV4) \
    hipLaunchKernelGGL((masked_ce_rows_kernel<NV4>), grid, block, 0, s, logits, ldl, labels, mask, label_smoothing, row_loss, M, V)
    if (V <= 256) PM_CE(1);
    else if (V <= 1024) PM_CE(4);
    else if (V <= 8192) PM_CE(32);
    else PM_CE(64);
#undef PM_CE
    PM_HIP(hipGetLastError());
    hipLaunchKernelGGL(masked_ce_reduce_kernel, dim3(1), dim3(1024), 0, s, row_loss, mask, loss_out, M);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}
