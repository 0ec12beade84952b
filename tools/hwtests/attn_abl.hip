// Ablation timing of the bf16 attention kernel (csrc/attention_bf16.hip): the kernel source is compiled several times
// with different ABL masks, each in its own namespace; results of the ablated variants are garbage, only time counts.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o attn_abl tools/hwtests/attn_abl.hip && ./attn_abl [B H N]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>
#include "../../paintmind_amd/csrc/common.h"
void pm_set_error(const char*, ...) {}
#define VARIANT(ns, mask) namespace ns {
#define ENDVARIANT }
#define ABL 0
namespace a0 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL
#define ABL 1
namespace a1 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL
#define ABL 2
namespace a2 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL
#define ABL 4
namespace a4 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL
#define ABL 8
namespace a8 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL
#define ABL 16
namespace a16 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL
#define ABL 32
namespace a32 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL
#define ABL 64
namespace a64 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL
#define ABL (32 + 64)
namespace a96 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL
#define ABL (2 + 4 + 8)
namespace a14 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL
#define ABL (1 + 2 + 4 + 8 + 16)
namespace a31 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL
#define ABL (1 + 2 + 4 + 8 + 16 + 32 + 64)
namespace a127 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL

// the round-2 kernel for reference
bool g_pm_timing_on = false;
PmTimer::PmTimer(int fam, hipStream_t s) : family(fam), stream(s), e0(nullptr), on(false) {}
PmTimer::~PmTimer() {}
namespace old {
#include "../../paintmind_amd/csrc/attention.hip"
int pm_attention_bf16(const void*, const void*, const void*, void*, int, int, int, int, int, int, int, hipStream_t) { return -1; }
int run(const void* Q, const void* K, const void* Vt, void* out, int ldo, int B, int heads, int Nq, int Nkv, int Nkv_pad, int use_exp2, hipStream_t s) {
    return pmhip_attention(PMHIP_BF16, Q, K, Vt, out, ldo, B, heads, Nq, Nkv, Nkv_pad, use_exp2, s);
}
}

typedef int (*fn_t)(const void*, const void*, const void*, void*, int, int, int, int, int, int, int, hipStream_t);
struct V { const char* name; fn_t fn; };

int main(int argc, char** argv) {
    int B = 64, H = 8, N = 1024;
    if (argc > 3) { B = atoi(argv[1]); H = atoi(argv[2]); N = atoi(argv[3]); }
    const size_t n = (size_t)B * H * N * 64;
    std::vector<unsigned short> hq(n), hk(n), hv(n);
    unsigned s = 12345;
    auto rnd = [&](float scale) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale; unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    for (size_t i = 0; i < n; ++i) { hq[i] = rnd(0.5f); hk[i] = rnd(1.0f); hv[i] = rnd(1.0f); }
    void *q, *k, *v, *o;
    hipMalloc(&q, n * 2); hipMalloc(&k, n * 2); hipMalloc(&v, n * 2); hipMalloc(&o, n * 2);
    hipMemcpy(q, hq.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(k, hk.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(v, hv.data(), n * 2, hipMemcpyHostToDevice);
    setenv("PMHIP_ATTN_OLD", "1", 1);
    V vs[] = {{"round-2 kernel", old::run}, {"full", a0::pm_attention_bf16}, {"no exp (mul)", a1::pm_attention_bf16}, {"no fragment reads", a2::pm_attention_bf16},
              {"no DMA", a4::pm_attention_bf16}, {"no barrier", a8::pm_attention_bf16}, {"no max check", a16::pm_attention_bf16},
              {"no output store", a32::pm_attention_bf16}, {"no Q load", a64::pm_attention_bf16}, {"no Q load, no store", a96::pm_attention_bf16},
              {"no reads/DMA/barrier", a14::pm_attention_bf16}, {"MFMA + pack only", a31::pm_attention_bf16}, {"MFMA + pack, no Q/store", a127::pm_attention_bf16}};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int round = 0; round < 2; ++round)
        for (auto& x : vs) {
            for (int i = 0; i < 3; ++i) x.fn(q, k, v, o, H * 64, B, H, N, N, N, 1, 0);
            hipEventRecord(e0, 0);
            const int reps = 20;
            for (int i = 0; i < reps; ++i) x.fn(q, k, v, o, H * 64, B, H, N, N, N, 1, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / reps;
            printf("%-28s %8.1f us  %7.1f TFLOP/s\n", x.name, us, 4.0 * N * N * 64 * B * H / us / 1e6);
        }
    return 0;
}
