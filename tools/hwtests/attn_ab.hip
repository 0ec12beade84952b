// Same-process A/B of attention kernel generations and ablations (csrc/attention_bf16.hip compiled several times, each in its
// own namespace; tools/experiments/attention_bf16_gen3.hip = the round-3/4 kernel).  Variants are timed in alternation, several
// rounds, and the un-ablated ones are compared with each other element by element.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I paintmind_amd/csrc -o tools/hwtests/attn_ab tools/hwtests/attn_ab.hip
//   ./attn_ab [B H N [rounds]]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>
#include "../../paintmind_amd/csrc/common.h"
void pm_set_error(const char*, ...) {}
#define PM_ATTN_NO_ABI 1
#define ABL 0
namespace gen3 {
#include "../experiments/attention_bf16_gen3.hip"
}
namespace gen4 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#define PM_ATTN_FORCE_QF 4
namespace g4_qf4 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef PM_ATTN_FORCE_QF
#define PM_ATTN_FORCE_QF 2
namespace g4_qf2 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef PM_ATTN_FORCE_QF
#define PM_ATTN_FORCE_QF 1
namespace g4_qf1 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef PM_ATTN_FORCE_QF
#define PM_ATTN_KSWZ 0
namespace g4_kswz0 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef PM_ATTN_KSWZ
#define PM_ATTN_RING 3
namespace g4_ring3 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef PM_ATTN_RING
#undef ABL
#define ABL 1
namespace g4_noexp {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL
#define ABL 2
namespace g4_noreads {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL
#define ABL 4
namespace g4_nodma {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL
#define ABL 8
namespace g4_nobarrier {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL
#define ABL (2 + 4 + 8)
namespace g4_nomove {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL
#define ABL (1 + 2 + 4 + 8)
namespace g4_skeleton {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef ABL

typedef int (*fn_t)(const void*, const void*, const void*, void*, int, int, int, int, int, int, int, hipStream_t);
struct V { const char* name; fn_t fn; bool exact; double us; };

int main(int argc, char** argv) {
    int B = 64, H = 8, N = 1024, rounds = 6;
    if (argc > 3) { B = atoi(argv[1]); H = atoi(argv[2]); N = atoi(argv[3]); }
    if (argc > 4) rounds = atoi(argv[4]);
    const size_t n = (size_t)B * H * N * 64;
    std::vector<unsigned short> hq(n), hk(n), hv(n);
    unsigned s = 12345;
    auto rnd = [&](float scale) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale; unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    for (size_t i = 0; i < n; ++i) { hq[i] = rnd(0.5f); hk[i] = rnd(1.0f); hv[i] = rnd(1.0f); }
    void *q, *k, *v, *o, *oref;
    hipMalloc(&q, n * 2); hipMalloc(&k, n * 2); hipMalloc(&v, n * 2); hipMalloc(&o, n * 2); hipMalloc(&oref, n * 2);
    hipMemcpy(q, hq.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(k, hk.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(v, hv.data(), n * 2, hipMemcpyHostToDevice);
    V vs[] = {{"gen3 (round 4)", gen3::pm_attention_bf16, true, 0}, {"gen4", gen4::pm_attention_bf16, true, 0},
              {"gen4 256 queries / WG", g4_qf4::pm_attention_bf16, true, 0}, {"gen4 128 queries / WG", g4_qf2::pm_attention_bf16, true, 0},
              {"gen4  64 queries / WG", g4_qf1::pm_attention_bf16, true, 0},
              {"gen4 old K swizzle", g4_kswz0::pm_attention_bf16, true, 0},
              {"gen4 3-stage ring", g4_ring3::pm_attention_bf16, true, 0},
              {"gen4 no exp (mul)", g4_noexp::pm_attention_bf16, false, 0}, {"gen4 no fragment reads", g4_noreads::pm_attention_bf16, false, 0},
              {"gen4 no DMA", g4_nodma::pm_attention_bf16, false, 0}, {"gen4 no barrier", g4_nobarrier::pm_attention_bf16, false, 0},
              {"gen4 no reads/DMA/barrier", g4_nomove::pm_attention_bf16, false, 0}, {"gen4 MFMA + pack only", g4_skeleton::pm_attention_bf16, false, 0}};
    // element-wise comparison of the exact variants (bf16 outputs)
    std::vector<unsigned short> ha(n), hb(n);
    gen3::pm_attention_bf16(q, k, v, oref, H * 64, B, H, N, N, N, 1, 0);
    gen4::pm_attention_bf16(q, k, v, o, H * 64, B, H, N, N, N, 1, 0);
    hipDeviceSynchronize();
    hipMemcpy(ha.data(), oref, n * 2, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), o, n * 2, hipMemcpyDeviceToHost);
    double maxd = 0, maxv = 0; size_t ndiff = 0;
    for (size_t i = 0; i < n; ++i) {
        unsigned ua = (unsigned)ha[i] << 16, ub = (unsigned)hb[i] << 16; float fa, fb; memcpy(&fa, &ua, 4); memcpy(&fb, &ub, 4);
        if (ha[i] != hb[i]) ++ndiff;
        if (!(fabs(fa - fb) <= maxd)) maxd = fabs(fa - fb);
        if (fabs(fa) > maxv) maxv = fabs(fa);
    }
    printf("gen4 vs gen3: %zu of %zu bf16 outputs differ, max abs diff %.3g (max |out| %.3g)\n", ndiff, n, maxd, maxv);
    {   // the workgroup size must not change a single bit (batch invariance of the model rests on it)
        fn_t qfs[3] = {g4_qf4::pm_attention_bf16, g4_qf2::pm_attention_bf16, g4_qf1::pm_attention_bf16};
        const char* names[3] = {"256", "128", "64"};
        for (int i = 0; i < 3; ++i) {
            hipMemset(oref, 0xff, n * 2);
            qfs[i](q, k, v, oref, H * 64, B, H, N, N, N, 1, 0);
            hipDeviceSynchronize();
            hipMemcpy(ha.data(), oref, n * 2, hipMemcpyDeviceToHost);
            size_t nd = 0;
            for (size_t j = 0; j < n; ++j) nd += ha[j] != hb[j];
            printf("gen4 (%s queries per workgroup) vs gen4 (auto): %zu of %zu outputs differ\n", names[i], nd, n);
        }
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int round = 0; round < rounds; ++round)
        for (auto& x : vs) {
            for (int i = 0; i < 3; ++i) x.fn(q, k, v, o, H * 64, B, H, N, N, N, 1, 0);
            hipEventRecord(e0, 0);
            const int reps = 10;
            for (int i = 0; i < reps; ++i) x.fn(q, k, v, o, H * 64, B, H, N, N, N, 1, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / reps;
            x.us += us / rounds;
            printf("%-28s %8.1f us  %7.1f TFLOP/s\n", x.name, us, 4.0 * N * N * 64 * B * H / us / 1e6);
        }
    printf("---- mean over %d rounds\n", rounds);
    for (auto& x : vs) printf("%-28s %8.1f us  %7.1f TFLOP/s\n", x.name, x.us, 4.0 * N * N * 64 * B * H / x.us / 1e6);
    return 0;
}
