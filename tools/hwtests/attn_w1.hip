// Same-process A/B of the one-wave-per-SIMD attention kernel (round 6; shelved: tools/experiments/attention_bf16_w1.hip) against the
// fourth generation at 256 queries per workgroup (the product's csrc/attention_bf16.hip):
// bit comparison on ordinary data and through the overflow fallback, then alternating timings.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I paintmind_amd/csrc -o tools/hwtests/attn_w1 tools/hwtests/attn_w1.hip
//   ./attn_w1 [B H N [rounds]]
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
#define PM_ATTN_FORCE_QF 4
namespace g4 {
#include "../../paintmind_amd/csrc/attention_bf16.hip"
}
#undef PM_ATTN_FORCE_QF
#define PM_ATTN_FORCE_QF 8
#ifdef W1_TIMING_ABL                       // the timed kernel itself ablated (cycles per phase of an ablation)
#undef ABL
#define ABL W1_TIMING_ABL
#endif
namespace g5 {
#include "../experiments/attention_bf16_w1.hip"
}
#undef ABL
#define ABL 0
#undef PM_ATTN_FORCE_QF
#ifdef W1_ABLATIONS
#undef ABL
#define ABL 1
#define PM_ATTN_FORCE_QF 8
namespace g5_noexp {
#include "../experiments/attention_bf16_w1.hip"
}
#undef ABL
#define ABL 2
namespace g5_noreads {
#include "../experiments/attention_bf16_w1.hip"
}
#undef ABL
#define ABL 4
namespace g5_nodma {
#include "../experiments/attention_bf16_w1.hip"
}
#undef ABL
#define ABL 8
namespace g5_nobar {
#include "../experiments/attention_bf16_w1.hip"
}
#undef ABL
#define ABL 12
namespace g5_nodmabar {
#include "../experiments/attention_bf16_w1.hip"
}
#undef ABL
#define ABL 15
namespace g5_skel {
#include "../experiments/attention_bf16_w1.hip"
}
#undef ABL
#define ABL 0
#undef PM_ATTN_FORCE_QF
#endif
#ifdef W1_VARIANT
#ifndef W1_VARIANT_QF
#define W1_VARIANT_QF 8
#endif
#define PM_ATTN_FORCE_QF W1_VARIANT_QF
#ifdef W1_VARIANT_DEF
#define W1_VARIANT_DEF_ON 1
#endif
namespace g5b {
#include W1_VARIANT
}
#undef PM_ATTN_FORCE_QF
#endif

#ifdef W1_VARIANT
#define NVAR 1
#else
#define NVAR 0
#endif
typedef int (*fn_t)(const void*, const void*, const void*, void*, int, int, int, int, int, int, int, hipStream_t);
struct V { const char* name; fn_t fn; double us; };

int main(int argc, char** argv) {
    int B = 64, H = 8, N = 1024, rounds = 6;
    if (argc > 3) { B = atoi(argv[1]); H = atoi(argv[2]); N = atoi(argv[3]); }
    if (argc > 4) rounds = atoi(argv[4]);
    const size_t n = (size_t)B * H * N * 64;
    std::vector<unsigned short> hq(n), hk(n), hv(n), ha(n), hb(n);
    unsigned s = 12345;
    auto rnd = [&](float scale) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale; unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    for (size_t i = 0; i < n; ++i) { hq[i] = rnd(0.5f); hk[i] = rnd(1.0f); hv[i] = rnd(1.0f); }
    void *q, *k, *v, *o, *oref;
    (void)hipMalloc(&q, n * 2); (void)hipMalloc(&k, n * 2); (void)hipMalloc(&v, n * 2); (void)hipMalloc(&o, n * 2); (void)hipMalloc(&oref, n * 2);
    hipMemcpy(q, hq.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(v, hv.data(), n * 2, hipMemcpyHostToDevice);
    V vs[] = {{"gen4 (256 queries / WG)", g4::pm_attention_bf16, 0}, {"gen5 (one wave per SIMD)", g5::pm_attention_bf16, 0},
#ifdef W1_VARIANT
              {"gen5 variant", g5b::pm_attention_bf16, 0},
#endif
#ifdef W1_ABLATIONS
              {"gen5 no exp (mul)", g5_noexp::pm_attention_bf16, 0}, {"gen5 no fragment reads", g5_noreads::pm_attention_bf16, 0},
              {"gen5 no DMA", g5_nodma::pm_attention_bf16, 0}, {"gen5 no barrier", g5_nobar::pm_attention_bf16, 0},
              {"gen5 no DMA, no barrier", g5_nodmabar::pm_attention_bf16, 0}, {"gen5 MFMA + pack only", g5_skel::pm_attention_bf16, 0},
#endif
    };
    const int nv = sizeof(vs) / sizeof(vs[0]);
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<unsigned short> hk2(hk);
        if (pass == 1)      // keys far outside the fast path's range, late in the context of (batch 0, head 1) and of the last (batch, head)
            for (int d = 0; d < 64; ++d) {
                hk2[((size_t)1 * N + 700) * 64 + d] = 0x4700;                                  // 32768.0
                hk2[(((size_t)B * H - 1) * N + 77) * 64 + d] = 0xc700;
            }
        hipMemcpy(k, hk2.data(), n * 2, hipMemcpyHostToDevice);
        hipMemset(oref, 0xee, n * 2);
        vs[0].fn(q, k, v, oref, H * 64, B, H, N, N, N, 1, 0);
        hipDeviceSynchronize();
        hipMemcpy(ha.data(), oref, n * 2, hipMemcpyDeviceToHost);
        for (int i = 1; i < nv && i < 2 + NVAR; ++i) {
            hipMemset(o, 0xff, n * 2);
            vs[i].fn(q, k, v, o, H * 64, B, H, N, N, N, 1, 0);
            hipError_t rc = hipDeviceSynchronize();
            hipMemcpy(hb.data(), o, n * 2, hipMemcpyDeviceToHost);
            size_t nd = 0, nan = 0, first = n;
            for (size_t j = 0; j < n; ++j) { if (ha[j] != hb[j]) { if (first == n) first = j; ++nd; } nan += (hb[j] & 0x7f80) == 0x7f80; }
            if (nd) {       // where: by 16-query tile inside the 512-query workgroup; how much
                size_t by_tile[32] = {0}; double maxd = 0;
                for (size_t j = 0; j < n; ++j) if (ha[j] != hb[j]) {
                    const size_t row = j / (H * 64);          // b * N + query
                    by_tile[(row % N % 512) / 16]++;
                    unsigned ua = (unsigned)ha[j] << 16, ub = (unsigned)hb[j] << 16; float fa, fb; memcpy(&fa, &ua, 4); memcpy(&fb, &ub, 4);
                    if (fabs(fa - fb) > maxd) maxd = fabs(fa - fb);
                }
                size_t by_q[16] = {0}, by_d[8] = {0}, by_b[8] = {0};
                for (size_t j = 0; j < n; ++j) if (ha[j] != hb[j]) { const size_t row = j / (H * 64); by_q[row % 16]++; by_d[(j % 64) / 8]++; by_b[(row / N) % 8]++; }
                printf("  by query in tile:"); for (int t = 0; t < 16; ++t) printf(" %zu", by_q[t]);
                printf("\n  by 8-wide d block:"); for (int t = 0; t < 8; ++t) printf(" %zu", by_d[t]);
                printf("\n  by batch %% 8:"); for (int t = 0; t < 8; ++t) printf(" %zu", by_b[t]);
                printf("\n  max abs diff %.4g; differing outputs by tile of the workgroup:", maxd);
                for (int t = 0; t < 32; ++t) printf(" %zu", by_tile[t]);
                printf("\n");
            }
            unsigned long long f4 = 0, f5 = 0;
            hipMemcpyFromSymbol(&f4, HIP_SYMBOL(g4::g_attn_fallbacks), 8); hipMemcpyFromSymbol(&f5, HIP_SYMBOL(g5::g_attn_fallbacks), 8);
            printf("%s vs gen4 (%s): %zu of %zu outputs differ (first at %zu), %zu non-finite, rc %d; fallbacks so far gen4 %llu gen5 %llu\n", vs[i].name,
                   pass ? "overflowing keys" : "ordinary data", nd, n, first, nan, (int)rc, f4, f5);
        }
    }
    hipMemcpy(k, hk.data(), n * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int round = 0; round < rounds; ++round)
        for (auto& x : vs) {
            for (int i = 0; i < 3; ++i) x.fn(q, k, v, o, H * 64, B, H, N, N, N, 1, 0);
            hipEventRecord(e0, 0);
            const int reps = 20;
            for (int i = 0; i < reps; ++i) x.fn(q, k, v, o, H * 64, B, H, N, N, N, 1, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            x.us += ms * 1e3 / reps / rounds;
        }
    for (auto& x : vs) printf("%-28s %8.1f us  %7.1f TFLOP/s\n", x.name, x.us, 4.0 * N * N * 64 * B * H / x.us / 1e6);
    {   // the same kernels writing a head-major output [B * H, N, 64] (heads = 1, ldo = 64: 16 KiB contiguous per 128 queries
        // instead of 128-byte pieces at a 1 KiB stride): what the output layout costs
        for (auto& x : vs) x.us = 0;
        for (int round = 0; round < rounds; ++round)
            for (auto& x : vs) {
                for (int i = 0; i < 3; ++i) x.fn(q, k, v, o, 64, B * H, 1, N, N, N, 1, 0);
                hipEventRecord(e0, 0);
                const int reps = 20;
                for (int i = 0; i < reps; ++i) x.fn(q, k, v, o, 64, B * H, 1, N, N, N, 1, 0);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                x.us += ms * 1e3 / reps / rounds;
            }
        for (auto& x : vs) printf("%-28s %8.1f us  %7.1f TFLOP/s   (head-major output)\n", x.name, x.us, 4.0 * N * N * 64 * B * H / x.us / 1e6);
    }
#ifdef PM_ATTN_W1_TIMING
    {   // cycles per phase of the one-wave-per-SIMD kernel, per wave (one launch)
        unsigned long long z[8] = {0}, t[8];
        hipMemcpyToSymbol(HIP_SYMBOL(g5::g_w1_times), z, sizeof z);
        g5::pm_attention_bf16(q, k, v, o, H * 64, B, H, N, N, N, 1, 0);
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(t, HIP_SYMBOL(g5::g_w1_times), sizeof t);
        const double w = (double)t[0];
        printf("gen5 cycles per (wave, item): start (cold: Q + tile 0; warm: ~0) %.0f, first half-tile %.0f, loop %.0f (%.1f per half-tile step, MFMA floor 1152), "
               "seam (vote, O -> LDS) %.0f; final flush per wave %.0f; pairs %llu\n",
               t[1] / w, t[2] / w, t[3] / w, t[3] / w / (N / 32), t[4] / w, t[5] / (double)(256 * 4), t[0]);
    }
#endif
    return 0;
}
