// HBM-bound row kernels: LayerNorm, patchify / un-patchify(+clamp), pad-convert, embedding gather.
// All are one pass over their input with 16-byte accesses; one wave owns one row where a
// reduction is involved (no LDS, no barriers).
#include "common.h"

namespace {

constexpr int THREADS = 256;

// ------------------------------------------------------------------------------------------------
// LayerNorm (torch.nn.LayerNorm: biased variance, eps inside the sqrt).
// Reference call sites: stage1/layers.py:55-56,109,148; stage2/transformer.py:45-47,90.
// NV4 > 0: the row (D = 256*NV4 floats) lives in registers, two-pass mean / variance.
// NV4 == 0: generic D (multiple of 4), three passes over global/L1.
// ------------------------------------------------------------------------------------------------
template <typename OutT, int NV4>
__global__ __launch_bounds__(THREADS) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            OutT* __restrict__ out, int M, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * D;
    OutT* orow = out + (size_t)row * D;
    const float invD = 1.0f / (float)D;
    if constexpr (NV4 > 0) {
        float4 v[NV4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            v[i] = *reinterpret_cast<const float4*>(xr + (i * 64 + lane) * 4);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mean = wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * invD + eps);
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int c0 = (i * 64 + lane) * 4;
            const float4 gm = *reinterpret_cast<const float4*>(gamma + c0);
            const float4 bt = *reinterpret_cast<const float4*>(beta + c0);
            store4(orow + c0, (v[i].x - mean) * rstd * gm.x + bt.x, (v[i].y - mean) * rstd * gm.y + bt.y,
                   (v[i].z - mean) * rstd * gm.z + bt.z, (v[i].w - mean) * rstd * gm.w + bt.w);
        }
    } else {
        float s = 0.f;
        for (int c0 = lane * 4; c0 < D; c0 += 256) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c0);
            s += (v.x + v.y) + (v.z + v.w);
        }
        const float mean = wave_sum(s) * invD;
        float q = 0.f;
        for (int c0 = lane * 4; c0 < D; c0 += 256) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c0);
            const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * invD + eps);
        for (int c0 = lane * 4; c0 < D; c0 += 256) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c0);
            const float4 gm = *reinterpret_cast<const float4*>(gamma + c0);
            const float4 bt = *reinterpret_cast<const float4*>(beta + c0);
            store4(orow + c0, (v.x - mean) * rstd * gm.x + bt.x, (v.y - mean) * rstd * gm.y + bt.y,
                   (v.z - mean) * rstd * gm.z + bt.z, (v.w - mean) * rstd * gm.w + bt.w);
        }
    }
}

template <typename OutT>
int launch_ln(const float* x, const float* g, const float* b, float eps, void* out, int M, int D, hipStream_t s) {
    dim3 grid(ceil_div(M, THREADS / 64)), block(THREADS);
    OutT* o = reinterpret_cast<OutT*>(out);
    PmTimer tm(FAM_LAYERNORM, s);
    switch (D) {
        case 512: hipLaunchKernelGGL((layernorm_kernel<OutT, 2>), grid, block, 0, s, x, g, b, eps, o, M, D); break;
        case 768: hipLaunchKernelGGL((layernorm_kernel<OutT, 3>), grid, block, 0, s, x, g, b, eps, o, M, D); break;
        case 1024: hipLaunchKernelGGL((layernorm_kernel<OutT, 4>), grid, block, 0, s, x, g, b, eps, o, M, D); break;
        default: hipLaunchKernelGGL((layernorm_kernel<OutT, 0>), grid, block, 0, s, x, g, b, eps, o, M, D); break;
    }
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// patchify: out[(b,ph,pw), (c,i,j)] = img[b,c,ph*P+i,pw*P+j]; one thread = 8 consecutive j
// ------------------------------------------------------------------------------------------------
template <typename OutT>
__global__ __launch_bounds__(THREADS) void patchify_kernel(const float* __restrict__ img, OutT* __restrict__ out, int B,
                                                           int C, int H, int W, int P) {
    const int Wp = W / P, Hp = H / P;
    const int K = C * P * P, K8 = K / 8;
    const size_t total = (size_t)B * Hp * Wp * K8;
    for (size_t idx = (size_t)blockIdx.x * THREADS + threadIdx.x; idx < total; idx += (size_t)gridDim.x * THREADS) {
        const int k8 = (int)(idx % K8);
        const size_t m = idx / K8;
        const int pw = (int)(m % Wp), ph = (int)((m / Wp) % Hp), b = (int)(m / ((size_t)Wp * Hp));
        const int k = k8 * 8;
        const int c = k / (P * P), i = (k / P) % P, j = k % P;
        const float* src = img + (((size_t)b * C + c) * H + (ph * P + i)) * W + pw * P + j;
        const float4 v0 = *reinterpret_cast<const float4*>(src);
        const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
        OutT* dst = out + m * K + k;
        store4(dst, v0.x, v0.y, v0.z, v0.w);
        store4(dst + 4, v1.x, v1.y, v1.z, v1.w);
    }
}

// un-patchify + clamp: img[b,c,ph*P+i,pw*P+j] = clamp(y[(b,ph,pw), (i*P+j)*C + c]); thread = 4 j
__global__ __launch_bounds__(THREADS) void unpatchify_kernel(const float* __restrict__ y, float* __restrict__ img, int B,
                                                             int C, int H, int W, int P, float lo, float hi) {
    const int Wp = W / P, Hp = H / P, W4 = W / 4;
    const int K = C * P * P;
    const size_t total = (size_t)B * C * H * W4;
    for (size_t idx = (size_t)blockIdx.x * THREADS + threadIdx.x; idx < total; idx += (size_t)gridDim.x * THREADS) {
        const int x4 = (int)(idx % W4);
        const int yy = (int)((idx / W4) % H);
        const int c = (int)((idx / ((size_t)W4 * H)) % C);
        const int b = (int)(idx / ((size_t)W4 * H * C));
        const int xx = x4 * 4;
        const int ph = yy / P, i = yy % P, pw = xx / P, j = xx % P;
        const float* src = y + (((size_t)b * Hp + ph) * Wp + pw) * K + (size_t)(i * P + j) * C + c;
        float4 v;
        v.x = fminf(fmaxf(src[0], lo), hi);
        v.y = fminf(fmaxf(src[C], lo), hi);
        v.z = fminf(fmaxf(src[2 * C], lo), hi);
        v.w = fminf(fmaxf(src[3 * C], lo), hi);
        *reinterpret_cast<float4*>(img + (((size_t)b * C + c) * H + yy) * W + xx) = v;
    }
}

// fp32 [M,K] -> OutT [M,Kpad] with zero padding; one thread = 4 output columns
template <typename OutT>
__global__ __launch_bounds__(THREADS) void convert_pad_kernel(const float* __restrict__ in, int K, OutT* __restrict__ out,
                                                              int Kpad, int M) {
    const int K4 = Kpad / 4;
    const size_t total = (size_t)M * K4;
    for (size_t idx = (size_t)blockIdx.x * THREADS + threadIdx.x; idx < total; idx += (size_t)gridDim.x * THREADS) {
        const int c0 = (int)(idx % K4) * 4;
        const size_t m = idx / K4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + 3 < K) v = *reinterpret_cast<const float4*>(in + m * K + c0);
        else {
            if (c0 + 0 < K) v.x = in[m * K + c0];
            if (c0 + 1 < K) v.y = in[m * K + c0 + 1];
            if (c0 + 2 < K) v.z = in[m * K + c0 + 2];
        }
        store4(out + m * Kpad + c0, v.x, v.y, v.z, v.w);
    }
}

// out[m,:] = table[ids[m],:], zero-padded to Kpad; one thread = 4 output columns
template <typename OutT>
__global__ __launch_bounds__(THREADS) void embed_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids,
                                                             OutT* __restrict__ out, int Kpad, int M, int V, int E) {
    const int K4 = Kpad / 4;
    const size_t total = (size_t)M * K4;
    for (size_t idx = (size_t)blockIdx.x * THREADS + threadIdx.x; idx < total; idx += (size_t)gridDim.x * THREADS) {
        const int c0 = (int)(idx % K4) * 4;
        const size_t m = idx / K4;
        int64_t id = ids[m];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);     // torch would raise; stay in bounds
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + 3 < E) v = *reinterpret_cast<const float4*>(table + (size_t)id * E + c0);
        store4(out + m * Kpad + c0, v.x, v.y, v.z, v.w);
    }
}

// out[m,:] = x[m,:] + table[m % table_rows,:]   (x + position_embedding, stage1/layers.py:146)
__global__ __launch_bounds__(THREADS) void add_rows_kernel(const float* __restrict__ x, const float* __restrict__ table,
                                                           int table_rows, float* __restrict__ out, int M, int D) {
    const int D4 = D / 4;
    const size_t total = (size_t)M * D4;
    for (size_t idx = (size_t)blockIdx.x * THREADS + threadIdx.x; idx < total; idx += (size_t)gridDim.x * THREADS) {
        const int c0 = (int)(idx % D4) * 4;
        const size_t m = idx / D4;
        const float4 a = *reinterpret_cast<const float4*>(x + m * D + c0);
        const float4 b = *reinterpret_cast<const float4*>(table + (m % table_rows) * D + c0);
        *reinterpret_cast<float4*>(out + m * D + c0) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

inline int grid_for(size_t total) {
    size_t blocks = (total + THREADS - 1) / THREADS;
    if (blocks > 256 * 8) blocks = 256 * 8;       // grid-stride the rest
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ stats, int nc, int M, float eps, float2* __restrict__ coef) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < nc; ++c) {                              // chunk order: deterministic
        const float2 t = *reinterpret_cast<const float2*>(stats + ((size_t)c * M + m) * 2);
        s1 += t.x; s2 += t.y;
    }
    const float invD = 1.0f / (float)(nc * 64);
    const float mean = s1 * invD;
    const float var = fmaxf(s2 * invD - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    coef[m] = make_float2(rstd, -rstd * mean);
}
}  // namespace

int pm_ln_finalize(const float* stats, int nc, int M, float eps, float* coef, pmhip_stream stream) {
    PM_REQUIRE(stats && coef && nc > 0 && M > 0, "ln_finalize: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    PmTimer tm(FAM_ROWOPS, s);
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((M + 255) / 256), dim3(256), 0, s, stats, nc, M, eps, reinterpret_cast<float2*>(coef));
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

extern "C" int pmhip_layernorm(const float* x, const float* gamma, const float* beta, float eps, void* out,
                               int out_dtype, int M, int D, pmhip_stream stream) {
    PM_REQUIRE(x && gamma && beta && out, "layernorm: null pointer");
    PM_REQUIRE(M > 0 && D > 0 && D % 4 == 0, "layernorm: bad shape M=%d D=%d (D must be a multiple of 4)", M, D);
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == PMHIP_F32) return launch_ln<float>(x, gamma, beta, eps, out, M, D, s);
    if (out_dtype == PMHIP_BF16) return launch_ln<bf16_t>(x, gamma, beta, eps, out, M, D, s);
    pm_set_error("layernorm: bad out dtype %d", out_dtype);
    return PMHIP_EINVAL;
}

extern "C" int pmhip_patchify(const float* img, void* out, int out_dtype, int B, int C, int H, int W, int P,
                              pmhip_stream stream) {
    PM_REQUIRE(img && out, "patchify: null pointer");
    PM_REQUIRE(B > 0 && C > 0 && P % 8 == 0 && H % P == 0 && W % P == 0, "patchify: bad geometry (P must be a multiple of 8)");
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)B * (H / P) * (W / P) * (C * P * P / 8);
    PmTimer tm(FAM_ROWOPS, s);
    if (out_dtype == PMHIP_F32)
        hipLaunchKernelGGL((patchify_kernel<float>), dim3(grid_for(total)), dim3(THREADS), 0, s, img, (float*)out, B, C, H, W, P);
    else
        hipLaunchKernelGGL((patchify_kernel<bf16_t>), dim3(grid_for(total)), dim3(THREADS), 0, s, img, (bf16_t*)out, B, C, H, W, P);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

extern "C" int pmhip_unpatchify_clamp(const float* y, float* img, int B, int C, int H, int W, int P, float lo,
                                      float hi, pmhip_stream stream) {
    PM_REQUIRE(y && img, "unpatchify: null pointer");
    PM_REQUIRE(B > 0 && C > 0 && P % 4 == 0 && H % P == 0 && W % P == 0, "unpatchify: bad geometry");
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)B * C * H * (W / 4);
    PmTimer tm(FAM_ROWOPS, s);
    hipLaunchKernelGGL(unpatchify_kernel, dim3(grid_for(total)), dim3(THREADS), 0, s, y, img, B, C, H, W, P, lo, hi);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

extern "C" int pmhip_convert_pad(const float* in, int K, void* out, int out_dtype, int Kpad, int M,
                                 pmhip_stream stream) {
    PM_REQUIRE(in && out, "convert_pad: null pointer");
    PM_REQUIRE(M > 0 && K > 0 && Kpad >= K && Kpad % 4 == 0, "convert_pad: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)M * (Kpad / 4);
    PmTimer tm(FAM_ROWOPS, s);
    if (out_dtype == PMHIP_F32)
        hipLaunchKernelGGL((convert_pad_kernel<float>), dim3(grid_for(total)), dim3(THREADS), 0, s, in, K, (float*)out, Kpad, M);
    else
        hipLaunchKernelGGL((convert_pad_kernel<bf16_t>), dim3(grid_for(total)), dim3(THREADS), 0, s, in, K, (bf16_t*)out, Kpad, M);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

extern "C" int pmhip_embed_rows(const float* table, const int64_t* ids, void* out, int out_dtype, int Kpad,
                                int M, int V, int E, pmhip_stream stream) {
    PM_REQUIRE(table && ids && out, "embed_rows: null pointer");
    PM_REQUIRE(M > 0 && V > 0 && E > 0 && E % 4 == 0 && Kpad >= E && Kpad % 4 == 0, "embed_rows: bad shape (E multiple of 4)");
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)M * (Kpad / 4);
    PmTimer tm(FAM_ROWOPS, s);
    if (out_dtype == PMHIP_F32)
        hipLaunchKernelGGL((embed_rows_kernel<float>), dim3(grid_for(total)), dim3(THREADS), 0, s, table, ids, (float*)out, Kpad, M, V, E);
    else
        hipLaunchKernelGGL((embed_rows_kernel<bf16_t>), dim3(grid_for(total)), dim3(THREADS), 0, s, table, ids, (bf16_t*)out, Kpad, M, V, E);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

extern "C" int pmhip_add_rows(const float* x, const float* table, int table_rows, float* out, int M, int D,
                              pmhip_stream stream) {
    PM_REQUIRE(x && table && out, "add_rows: null pointer");
    PM_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && table_rows > 0, "add_rows: bad shape");
    hipStream_t s = (hipStream_t)stream;
    PmTimer tm(FAM_ROWOPS, s);
    hipLaunchKernelGGL(add_rows_kernel, dim3(grid_for((size_t)M * (D / 4))), dim3(THREADS), 0, s, x, table, table_rows, out, M, D);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}
