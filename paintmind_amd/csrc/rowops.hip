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

// out = uncond + scale * (cond - uncond): logits of a text-conditioned and an unconditional forward combined for guidance
// (the reference TRAINS for this -- 10 % text drop, utils/trainer.py:379,387-388 -- but its inference never combines them; an
// extension behind an explicit keyword).  One fused multiply-add per element in fp32, in place when out aliases cond or uncond.
// (no __restrict__: out is documented to alias cond or uncond; every element is read and then written by the same lane)
// STATS: additionally the softmax statistics of every 64-element block of the result (16 consecutive lanes hold one block, four
// consecutive elements each -- the layout of common.h softmax_block_stat): what the logits GEMM leaves behind for an unguided step.
template <bool STATS>
__global__ __launch_bounds__(THREADS) void guidance_kernel(const float* cond, const float* uncond, float scale, float* out, size_t n4,
                                                           float2* block_stats) {
    for (size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x; i < n4; i += (size_t)gridDim.x * THREADS) {
        const float4 c = reinterpret_cast<const float4*>(cond)[i], u = reinterpret_cast<const float4*>(uncond)[i];
        const float4 r = make_float4(fmaf(scale, c.x - u.x, u.x), fmaf(scale, c.y - u.y, u.y), fmaf(scale, c.z - u.z, u.z),
                                     fmaf(scale, c.w - u.w, u.w));
        reinterpret_cast<float4*>(out)[i] = r;
        if constexpr (STATS) {                             // n4 % 16 == 0: the 16 lanes of a block run the same iterations
            const float2 st = softmax_block_stat(r.x, r.y, r.z, r.w);
            if ((threadIdx.x & 15) == 0) block_stats[i >> 4] = st;
        }
    }
}

inline int grid_for(size_t total) {
    size_t blocks = (total + THREADS - 1) / THREADS;
    if (blocks > 256 * 8) blocks = 256 * 8;       // grid-stride the rest
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

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

// ------------------------------------------------------------------------------------------------
// bf16 hi/lo residual stream (bf16-perf mode): x = hi + lo, two bf16 planes [M, D].
//   hilo_rows_kernel  one wave per row, the row in registers.  IN: f32 row, or hi + lo.  Statistics two-pass like
//                     layernorm_kernel.  OUT (any subset): LayerNorm(x) as f32 / bf16, LayerNorm(x) split into hi + lo,
//                     x itself split (no normalisation), x joined to f32, or only the row's (rstd, -rstd * mean) pair
//                     computed from the HI plane alone -- what a GEMM that consumes hi with the LayerNorm folded in needs
//                     (it multiplies bf16(x) = hi, so the statistics of hi are the self-consistent ones).
// ------------------------------------------------------------------------------------------------
enum { HL_LN_F32 = 0, HL_LN_BF16 = 1, HL_LN_HILO = 2, HL_SPLIT = 3, HL_JOIN = 4, HL_COEF = 5 };

template <bool IN_HILO, int MODE>
__global__ __launch_bounds__(THREADS) void hilo_rows_kernel(const float* __restrict__ x, const bf16_t* __restrict__ xh,
                                                            const bf16_t* __restrict__ xl, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, void* __restrict__ out,
                                                            bf16_t* __restrict__ out_lo, int M, int D, const float* __restrict__ shift) {
    constexpr int MAXV = 4;                                      // 4 columns per lane per step, D <= 1024
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    if (row >= M) return;
    const size_t base = (size_t)row * D;
    const float sh = (IN_HILO && shift) ? shift[row] : 0.f;     // centred stream (gemm_common.h): x = hi + lo + shift[row]
    const int nv = (D + 255) / 256;
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c0 = (i * 64 + lane) * 4;
        if (i < nv && c0 < D) {
            if constexpr (IN_HILO) {
                const uint2 h = *reinterpret_cast<const uint2*>(xh + base + c0);
                uint2 l = make_uint2(0u, 0u);
                if constexpr (MODE != HL_COEF) l = *reinterpret_cast<const uint2*>(xl + base + c0);
                v[i].x = __uint_as_float(h.x << 16) + __uint_as_float(l.x << 16);
                v[i].y = __uint_as_float(h.x & 0xffff0000u) + __uint_as_float(l.x & 0xffff0000u);
                v[i].z = __uint_as_float(h.y << 16) + __uint_as_float(l.y << 16);
                v[i].w = __uint_as_float(h.y & 0xffff0000u) + __uint_as_float(l.y & 0xffff0000u);
                if (shift) { v[i].x += sh; v[i].y += sh; v[i].z += sh; v[i].w += sh; }
            } else {
                v[i] = *reinterpret_cast<const float4*>(x + base + c0);
            }
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    float mean = 0.f, rstd = 1.f;
    if constexpr (MODE != HL_SPLIT && MODE != HL_JOIN) {
        const float invD = 1.0f / (float)D;
        mean = wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c0 = (i * 64 + lane) * 4;
            if (i < nv && c0 < D) {
                const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + b * b) + (c * c + d * d);
            }
        }
        rstd = 1.0f / sqrtf(wave_sum(q) * invD + eps);
    }
    if constexpr (MODE == HL_COEF) {
        if (lane == 0) *reinterpret_cast<float2*>(reinterpret_cast<float*>(out) + (size_t)row * 2) = make_float2(rstd, -rstd * mean);
        return;
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c0 = (i * 64 + lane) * 4;
        if (i < nv && c0 < D) {
            float4 y = v[i];
            if constexpr (MODE == HL_LN_F32 || MODE == HL_LN_BF16 || MODE == HL_LN_HILO) {
                const float4 gm = *reinterpret_cast<const float4*>(gamma + c0);
                const float4 bt = *reinterpret_cast<const float4*>(beta + c0);
                y = make_float4((v[i].x - mean) * rstd * gm.x + bt.x, (v[i].y - mean) * rstd * gm.y + bt.y,
                                (v[i].z - mean) * rstd * gm.z + bt.z, (v[i].w - mean) * rstd * gm.w + bt.w);
            }
            if constexpr (MODE == HL_LN_F32 || MODE == HL_JOIN) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + base + c0) = y;
            } else if constexpr (MODE == HL_LN_BF16) {
                store4(reinterpret_cast<bf16_t*>(out) + base + c0, y.x, y.y, y.z, y.w);
            } else {                                             // split into hi + lo
                const uint2 h = make_uint2(pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
                const uint2 l = make_uint2(pack_bf16x2(y.x - __uint_as_float(h.x << 16), y.y - __uint_as_float(h.x & 0xffff0000u)),
                                           pack_bf16x2(y.z - __uint_as_float(h.y << 16), y.w - __uint_as_float(h.y & 0xffff0000u)));
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + base + c0) = h;
                *reinterpret_cast<uint2*>(out_lo + base + c0) = l;
            }
        }
    }
}

// (rstd, -rstd * mean) of the rows of the hi plane alone.  RPW rows per wave, every row's 16-byte loads issued before the
// first is consumed (a wave with one 1-KiB row in flight is latency-bound: 3.9 TB/s); two-pass statistics in registers.
template <int RPW>
__global__ __launch_bounds__(THREADS) void ln_coef_kernel(const bf16_t* __restrict__ xh, float eps, float2* __restrict__ coef, int M, int D) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)) * RPW;
    if (row0 >= M) return;
    uint4 u[RPW][2];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r < M ? row0 + r : M - 1;
        const bf16_t* xr = xh + (size_t)row * D;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c0 = (i * 64 + lane) * 8;
            u[r][i] = make_uint4(0u, 0u, 0u, 0u);
            if (c0 < D) u[r][i] = *reinterpret_cast<const uint4*>(xr + c0);
        }
    }
    const float invD = 1.0f / (float)D;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float v[2][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned w[4] = {u[r][i].x, u[r][i].y, u[r][i].z, u[r][i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[i][2 * j] = __uint_as_float(w[j] << 16); v[i][2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
            s += ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) + ((v[i][4] + v[i][5]) + (v[i][6] + v[i][7]));     // absent chunks are zeros
        }
        const float mean = wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if ((i * 64 + lane) * 8 < D) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float a = v[i][j] - mean; q += a * a; }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * invD + eps);
        if (lane == 0 && row0 + r < M) coef[row0 + r] = make_float2(rstd, -rstd * mean);
    }
}

// The same coefficients from the partial statistics a hi/lo producer GEMM left behind (gemm_common.h: per row and per 64-column
// part the sum and the sum of squares centred on the part's own mean): Chan's combination, eight lanes per row.
// 16 bytes per part and row instead of a pass over the plane.
__global__ __launch_bounds__(THREADS) void ln_coef_parts_kernel(const float2* __restrict__ parts, int nparts, float eps,
                                                                float2* __restrict__ coef, int M) {
    // 8 lanes per row: lane j takes parts j and j + 8 (a wave reads 8 rows = 512 or 1024 contiguous bytes), DPP sums over the
    // 8 lanes in a fixed order
    const int t = blockIdx.x * THREADS + threadIdx.x;
    const int row = t >> 3, j = t & 7;
    const bool live = row < M;
    const float2* pr = parts + (size_t)(live ? row : 0) * nparts;
    const float2 p0 = (live && j < nparts) ? pr[j] : make_float2(0.f, 0.f);
    const float2 p1 = (live && j + 8 < nparts) ? pr[j + 8] : make_float2(0.f, 0.f);
    // the arithmetic is common.h's (lnp_*): the butterfly below is lnp_tree8's ((0+1)+(2+3)) + ((4+5)+(6+7)), and a lane-pair sum
    // is commutative -- ln_coef_row (one lane per row, in the small-batch folded GEMM) gives the same bits
    float s = __fadd_rn(p0.x, p1.x);
    s = __fadd_rn(s, dpp_mov<0xB1>(s)); s = __fadd_rn(s, dpp_mov<0x4E>(s)); s = __fadd_rn(s, dpp_mov<0x141>(s));
    const float mean = lnp_mean(s, nparts);
    float m2 = __fadd_rn(j < nparts ? lnp_m2_term(p0, mean) : 0.f, j + 8 < nparts ? lnp_m2_term(p1, mean) : 0.f);
    m2 = __fadd_rn(m2, dpp_mov<0xB1>(m2)); m2 = __fadd_rn(m2, dpp_mov<0x4E>(m2)); m2 = __fadd_rn(m2, dpp_mov<0x141>(m2));
    if (live && j == 0) coef[row] = lnp_finish(m2, mean, nparts, eps);
}

template <bool IN_HILO, int MODE>
static int launch_hilo(const float* x, const void* xh, const void* xl, const float* g, const float* b, float eps, void* out, void* out_lo,
                       int M, int D, hipStream_t s, const float* shift = nullptr) {
    PM_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "hi/lo row operator: D=%d must be a multiple of 4, <= 1024", D);
    PmTimer tm(FAM_LAYERNORM, s);
    hipLaunchKernelGGL((hilo_rows_kernel<IN_HILO, MODE>), dim3(ceil_div(M, THREADS / 64)), dim3(THREADS), 0, s, x,
                       reinterpret_cast<const bf16_t*>(xh), reinterpret_cast<const bf16_t*>(xl), g, b, eps, out,
                       reinterpret_cast<bf16_t*>(out_lo), M, D, shift);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

// LayerNorm of a hi/lo row -> f32 or bf16 (the unfolded consumer path: small batches, the decoder's final norm)
extern "C" int pmhip_layernorm_hilo(const void* x_hi, const void* x_lo, const float* gamma, const float* beta, float eps, void* out,
                                    int out_dtype, int M, int D, pmhip_stream stream) {
    PM_REQUIRE(x_hi && x_lo && gamma && beta && out, "layernorm_hilo: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == PMHIP_F32) return launch_hilo<true, HL_LN_F32>(nullptr, x_hi, x_lo, gamma, beta, eps, out, nullptr, M, D, s);
    if (out_dtype == PMHIP_BF16) return launch_hilo<true, HL_LN_BF16>(nullptr, x_hi, x_lo, gamma, beta, eps, out, nullptr, M, D, s);
    pm_set_error("layernorm_hilo: bad out dtype %d", out_dtype);
    return PMHIP_EINVAL;
}

// LayerNorm of an f32 row, result split into hi + lo (the encoder's norm_pre opens the residual stream)
extern "C" int pmhip_layernorm_to_hilo(const float* x, const float* gamma, const float* beta, float eps, void* out_hi, void* out_lo,
                                       int M, int D, pmhip_stream stream) {
    PM_REQUIRE(x && gamma && beta && out_hi && out_lo, "layernorm_to_hilo: null pointer");
    return launch_hilo<false, HL_LN_HILO>(x, nullptr, nullptr, gamma, beta, eps, out_hi, out_lo, M, D, (hipStream_t)stream);
}

// x (f32) -> hi + lo, and back
extern "C" int pmhip_split_hilo(const float* x, void* out_hi, void* out_lo, int M, int D, pmhip_stream stream) {
    PM_REQUIRE(x && out_hi && out_lo, "split_hilo: null pointer");
    return launch_hilo<false, HL_SPLIT>(x, nullptr, nullptr, nullptr, nullptr, 0.f, out_hi, out_lo, M, D, (hipStream_t)stream);
}
extern "C" int pmhip_join_hilo(const void* x_hi, const void* x_lo, float* out, int M, int D, pmhip_stream stream) {
    PM_REQUIRE(x_hi && x_lo && out, "join_hilo: null pointer");
    return launch_hilo<true, HL_JOIN>(nullptr, x_hi, x_lo, nullptr, nullptr, 0.f, out, nullptr, M, D, (hipStream_t)stream);
}

// centred stream -> plain pair, in place: (hi, lo) <- split(hi + lo + shift[row]) (where the absolute x is needed again)
extern "C" int pmhip_unshift_hilo(void* x_hi, void* x_lo, const float* shift, int M, int D, pmhip_stream stream) {
    PM_REQUIRE(x_hi && x_lo && shift, "unshift_hilo: null pointer");
    return launch_hilo<true, HL_SPLIT>(nullptr, x_hi, x_lo, nullptr, nullptr, 0.f, x_hi, x_lo, M, D, (hipStream_t)stream, shift);
}

// per-row (rstd, -rstd * mean) of the hi plane: the coefficients of a GEMM with the LayerNorm folded in (pmhip_lnfold)
extern "C" int pmhip_ln_coef(const void* x_hi, float eps, float* coef, int M, int D, pmhip_stream stream) {
    PM_REQUIRE(x_hi && coef, "ln_coef: null pointer");
    PM_REQUIRE(M > 0 && D > 0 && D % 8 == 0 && D <= 1024, "ln_coef: D=%d must be a multiple of 8, <= 1024", D);
    hipStream_t s = (hipStream_t)stream;
    PmTimer tm(FAM_LAYERNORM, s);
    constexpr int RPW = 4;
    hipLaunchKernelGGL((ln_coef_kernel<RPW>), dim3(ceil_div(M, (THREADS / 64) * RPW)), dim3(THREADS), 0, s,
                       reinterpret_cast<const bf16_t*>(x_hi), eps, reinterpret_cast<float2*>(coef), M, D);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

extern "C" int pmhip_ln_coef_parts(const float* row_stats, int nparts, float eps, float* coef, int M, pmhip_stream stream) {
    PM_REQUIRE(row_stats && coef, "ln_coef_parts: null pointer");
    PM_REQUIRE(M > 0 && nparts > 0 && nparts <= 16, "ln_coef_parts: nparts=%d must be in [1,16] (D = 64 * nparts <= 1024)", nparts);
    hipStream_t s = (hipStream_t)stream;
    PmTimer tm(FAM_LAYERNORM, s);
    hipLaunchKernelGGL(ln_coef_parts_kernel, dim3(ceil_div(M * 8, THREADS)), dim3(THREADS), 0, s,
                       reinterpret_cast<const float2*>(row_stats), nparts, eps, reinterpret_cast<float2*>(coef), M);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
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

static int guidance_impl(const float* cond, const float* uncond, float scale, float* out, size_t n, float* block_stats, pmhip_stream stream) {
    PM_REQUIRE(cond && uncond && out, "guidance_combine: null pointer");
    PM_REQUIRE(n > 0 && n % 4 == 0, "guidance_combine: n must be a positive multiple of 4");
    PM_REQUIRE(!block_stats || n % 64 == 0, "guidance_combine_stats: n must be a multiple of 64 (rows of whole 64-column blocks)");
    hipStream_t s = (hipStream_t)stream;
    PmTimer tm(FAM_ROWOPS, s);
    if (block_stats)
        hipLaunchKernelGGL(guidance_kernel<true>, dim3(grid_for(n / 4)), dim3(THREADS), 0, s, cond, uncond, scale, out, n / 4,
                           reinterpret_cast<float2*>(block_stats));
    else
        hipLaunchKernelGGL(guidance_kernel<false>, dim3(grid_for(n / 4)), dim3(THREADS), 0, s, cond, uncond, scale, out, n / 4, (float2*)nullptr);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

extern "C" int pmhip_guidance_combine(const float* cond, const float* uncond, float scale, float* out, size_t n, pmhip_stream stream) {
    return guidance_impl(cond, uncond, scale, out, n, nullptr, stream);
}

// the same, plus block_stats [n / 64][2]: softmax statistics of the combined logits for pmhip_sample_rows_stats (contiguous rows
// whose length is a multiple of 64)
extern "C" int pmhip_guidance_combine_stats(const float* cond, const float* uncond, float scale, float* out, size_t n, float* block_stats,
                                            pmhip_stream stream) {
    PM_REQUIRE(block_stats, "guidance_combine_stats: null statistics buffer");
    return guidance_impl(cond, uncond, scale, out, n, block_stats, stream);
}
