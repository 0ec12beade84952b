// Attention for dim_head != 64 (reference modules/attention.py:27-33 takes any dim_head; every shipped config uses 64,
// which the tuned kernels in attention.hip / the head-split GEMM epilogue are built for).  This file is the plain path
// behind pmhip_gemm_heads_dh / pmhip_attention_dh: correct for dim_head = 16, 32, ..., 128, not tuned.
//
//   head split : the projection is an ordinary GEMM into an f32 scratch [M, nparts*heads*dh]; one thread per element
//                moves it into Q [B,H,t,dh] (scaled), K [B,H,tp,dh] or V^T [B,H,dh,tp] and rounds to the compute type once.
//   attention  : flash-style online softmax on the vector ALU.  4 lanes share one query row, each owning dh/4 contiguous
//                dims of q and of the output; the score is a quad reduction (two DPP steps).  All lanes of a workgroup
//                walk the keys in the same order, so K / V^T addresses are wave-uniform per quad position and come out of
//                L1/L2 as broadcasts.  The score matrix is never materialised.
#include "common.h"

namespace {

struct SplitParams {
    const float* src;
    void* outs[3];
    int kinds[3];
    int M, nparts, heads, dh, tokens, tokens_pad;
    float q_scale;
};

template <typename T>
__global__ __launch_bounds__(256) void head_split_kernel(SplitParams p) {
    const int inner = p.heads * p.dh, N = p.nparts * inner;
    const long long total = (long long)p.M * N;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int m = (int)(idx / N), col = (int)(idx % N);
        const int part = col / inner, hc = col % inner, h = hc / p.dh, d = hc % p.dh;
        const int b = m / p.tokens, t = m % p.tokens;
        const float v = p.src[idx];
        T* o = reinterpret_cast<T*>(p.outs[part]);
        const size_t bh = (size_t)b * p.heads + h;
        if (p.kinds[part] == PMHIP_PART_Q)
            o[(bh * p.tokens + t) * p.dh + d] = from_f32<T>(v * p.q_scale);
        else if (p.kinds[part] == PMHIP_PART_K)
            o[(bh * p.tokens_pad + t) * p.dh + d] = from_f32<T>(v);
        else
            o[(bh * p.dh + d) * p.tokens_pad + t] = from_f32<T>(v);
    }
}

__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    return v;
}

// DPL = dims per lane = dh / 4
template <typename T, int DPL, bool EXP2>
__global__ __launch_bounds__(256) void attention_dh_kernel(const T* __restrict__ Q, const T* __restrict__ K, const T* __restrict__ Vt,
                                                          T* __restrict__ out, int ldo, int heads, int Nq, int Nkv, int Nkp) {
    constexpr int DH = DPL * 4;
    const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
    const int qi = blockIdx.x * 64 + (threadIdx.x >> 2), part = threadIdx.x & 3;
    const int qr = qi < Nq ? qi : Nq - 1;                       // rows past the end compute a copy and do not store
    float q[DPL], o[DPL];
    const T* qp = Q + ((size_t)bh * Nq + qr) * DH + part * DPL;
#pragma unroll
    for (int i = 0; i < DPL; ++i) { q[i] = to_f32<T>(qp[i]); o[i] = 0.f; }
    const T* kp = K + (size_t)bh * Nkp * DH + part * DPL;
    const T* vp = Vt + ((size_t)bh * DH + part * DPL) * Nkp;
    float m = -INFINITY, l = 0.f;
    for (int j = 0; j < Nkv; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < DPL; ++i) s = fmaf(q[i], to_f32<T>(kp[(size_t)j * DH + i]), s);
        s = quad_sum(s);
        const float mn = fmaxf(m, s);
        const float a = EXP2 ? exp2f(m - mn) : expf(m - mn);
        const float pj = EXP2 ? exp2f(s - mn) : expf(s - mn);
        l = fmaf(l, a, pj);
#pragma unroll
        for (int i = 0; i < DPL; ++i) o[i] = fmaf(o[i], a, pj * to_f32<T>(vp[(size_t)i * Nkp + j]));
        m = mn;
    }
    if (qi < Nq) {
        const float inv = 1.f / l;
        T* op = out + ((size_t)b * Nq + qi) * ldo + h * DH + part * DPL;
#pragma unroll
        for (int i = 0; i < DPL; ++i) op[i] = from_f32<T>(o[i] * inv);
    }
}

template <typename T, int DPL>
void launch_dh(const void* Q, const void* K, const void* Vt, void* out, int ldo, int B, int heads, int Nq, int Nkv, int Nkp,
               int use_exp2, hipStream_t s) {
    dim3 grid((Nq + 63) / 64, B * heads), block(256);
    if (use_exp2)
        hipLaunchKernelGGL((attention_dh_kernel<T, DPL, true>), grid, block, 0, s, (const T*)Q, (const T*)K, (const T*)Vt, (T*)out, ldo,
                           heads, Nq, Nkv, Nkp);
    else
        hipLaunchKernelGGL((attention_dh_kernel<T, DPL, false>), grid, block, 0, s, (const T*)Q, (const T*)K, (const T*)Vt, (T*)out, ldo,
                           heads, Nq, Nkv, Nkp);
}

template <typename T>
int dispatch_dh(int dh, const void* Q, const void* K, const void* Vt, void* out, int ldo, int B, int heads, int Nq, int Nkv, int Nkp,
                int use_exp2, hipStream_t s) {
    switch (dh / 4) {
#define PM_DH_CASE(n) case n: launch_dh<T, n>(Q, K, Vt, out, ldo, B, heads, Nq, Nkv, Nkp, use_exp2, s); break;
        PM_DH_CASE(4) PM_DH_CASE(8) PM_DH_CASE(12) PM_DH_CASE(16) PM_DH_CASE(20) PM_DH_CASE(24) PM_DH_CASE(28) PM_DH_CASE(32)
#undef PM_DH_CASE
        default: PM_REQUIRE(false, "attention_dh: dim_head=%d not served", dh);
    }
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

int check_dh(const char* who, int heads, int dim_head) {
    PM_REQUIRE(dim_head >= 16 && dim_head <= 128 && dim_head % 16 == 0, "%s: dim_head=%d must be a multiple of 16 in [16,128]", who, dim_head);
    PM_REQUIRE(heads > 0 && (heads * dim_head) % 64 == 0, "%s: heads*dim_head=%d must be a multiple of 64", who, heads * dim_head);
    return PMHIP_OK;
}

}  // namespace

extern "C" int pmhip_gemm_heads_dh(int dtype, const void* A, int lda, const void* W, int ldw, int M, int K, int heads, int dim_head,
                                   int tokens, int tokens_pad, int nparts, const int* part_kinds_host, void* const* part_outs_host,
                                   float q_scale, float* scratch, pmhip_stream stream) {
    if (dim_head == 64)
        return pmhip_gemm_heads(dtype, A, lda, W, ldw, M, K, heads, tokens, tokens_pad, nparts, part_kinds_host, part_outs_host, q_scale,
                                stream);
    PM_TRY(check_dh("gemm_heads_dh", heads, dim_head));
    PM_REQUIRE(dtype == PMHIP_F32 || dtype == PMHIP_BF16, "gemm_heads_dh: bad dtype %d", dtype);
    PM_REQUIRE(nparts >= 1 && nparts <= 3 && part_kinds_host && part_outs_host, "gemm_heads_dh: nparts=%d", nparts);
    PM_REQUIRE(tokens > 0 && tokens_pad >= tokens && M > 0 && M % tokens == 0, "gemm_heads_dh: bad token geometry M=%d tokens=%d", M, tokens);
    PM_REQUIRE(scratch, "gemm_heads_dh: dim_head=%d needs the f32 scratch [M, nparts*heads*dim_head]", dim_head);
    SplitParams p{};
    p.src = scratch; p.M = M; p.nparts = nparts; p.heads = heads; p.dh = dim_head; p.tokens = tokens; p.tokens_pad = tokens_pad;
    p.q_scale = q_scale;
    for (int i = 0; i < nparts; ++i) {
        PM_REQUIRE(part_outs_host[i] && part_kinds_host[i] >= 0 && part_kinds_host[i] <= 2, "gemm_heads_dh: bad part %d", i);
        p.outs[i] = part_outs_host[i]; p.kinds[i] = part_kinds_host[i];
    }
    const int N = nparts * heads * dim_head;
    PM_TRY(pmhip_gemm(dtype, A, lda, W, ldw, nullptr, nullptr, 0, 0, scratch, N, PMHIP_F32, M, N, K, stream));
    hipStream_t s = (hipStream_t)stream;
    PmTimer tm(FAM_ROWOPS, s);
    const long long total = (long long)M * N;
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    if (dtype == PMHIP_F32) hipLaunchKernelGGL(head_split_kernel<float>, dim3(blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(head_split_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, p);
    PM_HIP(hipGetLastError());
    return PMHIP_OK;
}

extern "C" int pmhip_attention_dh(int dtype, const void* Q, const void* K, const void* Vt, void* out, int ldo, int B, int heads,
                                  int dim_head, int Nq, int Nkv, int Nkv_pad, int use_exp2, pmhip_stream stream) {
    if (dim_head == 64) return pmhip_attention(dtype, Q, K, Vt, out, ldo, B, heads, Nq, Nkv, Nkv_pad, use_exp2, stream);
    PM_TRY(check_dh("attention_dh", heads, dim_head));
    PM_REQUIRE(dtype == PMHIP_F32 || dtype == PMHIP_BF16, "attention_dh: bad dtype %d", dtype);
    PM_REQUIRE(Q && K && Vt && out, "attention_dh: null pointer");
    PM_REQUIRE(B > 0 && Nq > 0 && Nkv > 0 && Nkv_pad >= Nkv, "attention_dh: empty problem");
    PM_REQUIRE((long long)B * heads <= 65535, "attention_dh: B*heads=%lld exceeds the grid's y range", (long long)B * heads);
    hipStream_t s = (hipStream_t)stream;
    PmTimer tm(FAM_ATTENTION, s);
    if (dtype == PMHIP_F32) return dispatch_dh<float>(dim_head, Q, K, Vt, out, ldo, B, heads, Nq, Nkv, Nkv_pad, use_exp2, s);
    return dispatch_dh<bf16_t>(dim_head, Q, K, Vt, out, ldo, B, heads, Nq, Nkv, Nkv_pad, use_exp2, s);
}
