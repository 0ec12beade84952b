// Hardware question: the LayerNorm-folded 256x256 GEMM showed, once per ~10^9 wave-tiles, ONE accumulator
// register wrong in lanes 48-63 -- always in a LEAD wave (its normalisation, 128 back-to-back v_pk_fma_f32, runs while the LAG
// wave of the same SIMD issues its last 16 MFMAs), always the LOW half of a packed result, always in the part of the burst that
// overlaps those MFMAs.  Does a dense v_pk_fma_f32 burst return wrong results when the sibling wave of the SIMD runs MFMAs?
//
// Waves 0-3 of a 512-thread workgroup (one per SIMD) run the burst (the exact instruction forms of the kernel: op_sel:[0,1,0]
// and op_sel_hi:[0,1,1]) on fixed register inputs and compare every result bit for bit with the result of the same burst
// computed while the whole workgroup was quiet; waves 4-7 (the siblings) either idle or run v_mfma_f32_16x16x32_bf16 blocks,
// started by the same s_barrier as the burst.
//   hipcc --offload-arch=gfx950 -O2 -o pkfma_mfma tools/hwtests/pkfma_mfma.hip && ./pkfma_mfma [seconds_per_mode] [quick]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Report {
    unsigned long long bursts;           // bursts checked (per wave)
    unsigned bad[4][2];                  // mismatching results by 16-lane group and half of the packed pair
    unsigned hist[64];                   // ... and by result index (order of the burst: idx = ((mi*2 + ni)*2 + pair)*2 + half)
    unsigned samples;
    struct { unsigned wg, wave, lane, idx, iter; float got, want; } sample[32];
};

#define PK_T(t, cc, ab, dd) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(t) : "v"(cc), "v"(ab), "v"(dd))
#define PK_O(o, ab, acc, t) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(o) : "v"(ab), "v"(acc), "v"(t))

#define F_T(t, cc, ab, dd) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(cc), "v"(ab), "v"(dd))
#define NOP1 asm volatile("s_nop 0")

// 64 packed FMAs: 8 "rows" (ab[mi]) x 2 "column groups" x the two register pairs of a 4-column slice, as in ln_apply.
// AV (consumer side): 0 = the kernel's order (t0, t1, o0, o1: a result is consumed two instructions later); 1 = the same
// arithmetic with v_fma_f32 (eight per step, a result consumed four instructions later); 2 = as 0 with one s_nop 0 in front of the
// consumers; 3 = all 32 producers first, then the 32 consumers; 4 = compiler-generated packed FMAs (no inline asm).
template <int AV>
__device__ __forceinline__ void burst(const f32x2 (&ab)[8], const f32x2 (&cc)[2][2], const f32x2 (&dd)[2][2], const f32x2 (&acc)[8][2][2],
                                      f32x2 (&out)[8][2][2]) {
    if constexpr (AV == 3) {
        f32x2 t[8][2][2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) { PK_T(t[mi][ni][1], cc[ni][1], ab[mi], dd[ni][1]); PK_T(t[mi][ni][0], cc[ni][0], ab[mi], dd[ni][0]); }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) { PK_O(out[mi][ni][1], ab[mi], acc[mi][ni][1], t[mi][ni][1]); PK_O(out[mi][ni][0], ab[mi], acc[mi][ni][0], t[mi][ni][0]); }
        return;
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            if constexpr (AV == 1) {
                float t[4], o[4];
                F_T(t[2], cc[ni][1][0], ab[mi][1], dd[ni][1][0]); F_T(t[3], cc[ni][1][1], ab[mi][1], dd[ni][1][1]);
                F_T(t[0], cc[ni][0][0], ab[mi][1], dd[ni][0][0]); F_T(t[1], cc[ni][0][1], ab[mi][1], dd[ni][0][1]);
                F_T(o[2], ab[mi][0], acc[mi][ni][1][0], t[2]); F_T(o[3], ab[mi][0], acc[mi][ni][1][1], t[3]);
                F_T(o[0], ab[mi][0], acc[mi][ni][0][0], t[0]); F_T(o[1], ab[mi][0], acc[mi][ni][0][1], t[1]);
                out[mi][ni][1] = f32x2{o[2], o[3]}; out[mi][ni][0] = f32x2{o[0], o[1]};
            } else if constexpr (AV == 4) {
                const f32x2 y = f32x2{ab[mi][1], ab[mi][1]}, x = f32x2{ab[mi][0], ab[mi][0]};
                out[mi][ni][1] = __builtin_elementwise_fma(x, acc[mi][ni][1], __builtin_elementwise_fma(cc[ni][1], y, dd[ni][1]));
                out[mi][ni][0] = __builtin_elementwise_fma(x, acc[mi][ni][0], __builtin_elementwise_fma(cc[ni][0], y, dd[ni][0]));
            } else if constexpr (AV == 5) {              // the fix: both halves of every broadcast operand are real registers, no op_sel
                float x0, x1, y0, y1;
                asm volatile("v_mov_b32 %0, %1" : "=v"(x0) : "v"(ab[mi][0])); asm volatile("v_mov_b32 %0, %1" : "=v"(x1) : "v"(ab[mi][0]));
                asm volatile("v_mov_b32 %0, %1" : "=v"(y0) : "v"(ab[mi][1])); asm volatile("v_mov_b32 %0, %1" : "=v"(y1) : "v"(ab[mi][1]));
                const f32x2 xx = f32x2{x0, x1}, yy = f32x2{y0, y1};
                f32x2 t0, t1;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(cc[ni][1]), "v"(yy), "v"(dd[ni][1]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(cc[ni][0]), "v"(yy), "v"(dd[ni][0]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(out[mi][ni][1]) : "v"(xx), "v"(acc[mi][ni][1]), "v"(t0));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(out[mi][ni][0]) : "v"(xx), "v"(acc[mi][ni][0]), "v"(t1));
            } else if constexpr (AV == 10 || AV == 11) { // HIGH half taking src1 / src2 from the LOW register (what hipcc emits to broadcast an even register)
                f32x2 t0, t1;
                if constexpr (AV == 10) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t0) : "v"(cc[ni][1]), "v"(ab[mi]), "v"(dd[ni][1]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t1) : "v"(cc[ni][0]), "v"(ab[mi]), "v"(dd[ni][0]));
                } else {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(t0) : "v"(cc[ni][1]), "v"(dd[ni][1]), "v"(ab[mi]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(t1) : "v"(cc[ni][0]), "v"(dd[ni][0]), "v"(ab[mi]));
                }
                PK_O(out[mi][ni][1], ab[mi], acc[mi][ni][1], t0);
                PK_O(out[mi][ni][0], ab[mi], acc[mi][ni][0], t1);
            } else if constexpr (AV >= 6 && AV <= 9) {   // other low-half-from-high-register forms
                f32x2 t0, t1;
                if constexpr (AV == 6) {
                    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t0) : "v"(cc[ni][1]), "v"(ab[mi]));
                    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t1) : "v"(cc[ni][0]), "v"(ab[mi]));
                } else if constexpr (AV == 7) {
                    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t0) : "v"(cc[ni][1]), "v"(ab[mi]));
                    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t1) : "v"(cc[ni][0]), "v"(ab[mi]));
                } else if constexpr (AV == 8) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(t0) : "v"(ab[mi]), "v"(cc[ni][1]), "v"(dd[ni][1]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(t1) : "v"(ab[mi]), "v"(cc[ni][0]), "v"(dd[ni][0]));
                } else {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(t0) : "v"(cc[ni][1]), "v"(dd[ni][1]), "v"(ab[mi]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(t1) : "v"(cc[ni][0]), "v"(dd[ni][0]), "v"(ab[mi]));
                }
                PK_O(out[mi][ni][1], ab[mi], acc[mi][ni][1], t0);
                PK_O(out[mi][ni][0], ab[mi], acc[mi][ni][0], t1);
            } else {
                f32x2 t0, t1;
                PK_T(t0, cc[ni][1], ab[mi], dd[ni][1]);
                PK_T(t1, cc[ni][0], ab[mi], dd[ni][0]);
                if constexpr (AV == 2) NOP1;
                PK_O(out[mi][ni][1], ab[mi], acc[mi][ni][1], t0);
                PK_O(out[mi][ni][0], ab[mi], acc[mi][ni][0], t1);
            }
        }
}

// BV (sibling side): 0 = idle; 1 = s_setprio 1 + 16 MFMAs (the kernel); 2 = 16 MFMAs without s_setprio; 3 = s_setprio 1 + 64 v_fma_f32;
// 4 = s_setprio 1 / 0 only; 5 = s_setprio 1 + 16 v_mfma_f32_16x16x4_f32; 6 = s_setprio 1 + 8 v_mfma_f32_32x32x16_bf16.  DELAY: s_nop 15 (16 cycles each) in front of the sibling's work.
// LDSRD: the VALU waves read their coefficients from LDS right in front of the burst, as the kernel does.
template <int AV, int BV, int DELAY, bool LDSRD>
__global__ __launch_bounds__(512) void probe(const float* seed, Report* rep, int iters) {
    __shared__ float ref_lds[64][256];                   // quiet results of the four VALU waves, [result][lane]
    __shared__ float got_lds[64][256];                   // written only when a mismatch was seen
    __shared__ float scratch[4][512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool valu_wave = wave < 4;
    const float* s = seed + (size_t)(blockIdx.x * 512 + tid) * 96;
    for (int i = tid; i < 4 * 512; i += 512) (&scratch[0][0])[i] = seed[i];
    if (valu_wave) {
        f32x2 ab[8], cc[2][2], dd[2][2], acc[8][2][2], out[8][2][2];
#pragma unroll
        for (int i = 0; i < 8; ++i) ab[i] = f32x2{s[i * 2], s[i * 2 + 1]};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) { cc[i][j] = f32x2{s[16 + i * 4 + j * 2], s[17 + i * 4 + j * 2]}; dd[i][j] = f32x2{s[24 + i * 4 + j * 2], s[25 + i * 4 + j * 2]}; }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 2; ++k) acc[i][j][k] = f32x2{s[32 + i * 8 + j * 4 + k * 2], s[33 + i * 8 + j * 4 + k * 2]};
        burst<AV>(ab, cc, dd, acc, out);                 // quiet: nobody runs MFMAs yet
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int h = 0; h < 2; ++h) ref_lds[((i * 2 + j) * 2 + k) * 2 + h][tid] = out[i][j][k][h];
        __syncthreads();
        for (int it = 0; it < iters; ++it) {
            __builtin_amdgcn_s_barrier();
            if constexpr (LDSRD) {
                f32x4 x0 = *reinterpret_cast<const f32x4*>(&scratch[wave][lane * 4]);
                f32x4 x1 = *reinterpret_cast<const f32x4*>(&scratch[wave][256 + (lane & 15) * 4]);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1));
            }
            burst<AV>(ab, cc, dd, acc, out);
            unsigned ne = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            ne |= __float_as_uint(out[i][j][k][h]) ^ __float_as_uint(ref_lds[((i * 2 + j) * 2 + k) * 2 + h][tid]);
            if (__builtin_expect(__any(ne != 0), 0)) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int k = 0; k < 2; ++k)
#pragma unroll
                            for (int h = 0; h < 2; ++h) got_lds[((i * 2 + j) * 2 + k) * 2 + h][tid] = out[i][j][k][h];
                for (int idx = 0; idx < 64; ++idx) {
                    const float gv = got_lds[idx][tid], wv = ref_lds[idx][tid];
                    if (__float_as_uint(gv) != __float_as_uint(wv)) {
                        atomicAdd(&rep->bad[lane >> 4][idx & 1], 1u);
                        atomicAdd(&rep->hist[idx], 1u);
                        const unsigned slot = atomicAdd(&rep->samples, 1u);
                        if (slot < 32) {
                            rep->sample[slot].wg = blockIdx.x; rep->sample[slot].wave = wave; rep->sample[slot].lane = lane;
                            rep->sample[slot].idx = idx; rep->sample[slot].iter = it; rep->sample[slot].got = gv; rep->sample[slot].want = wv;
                        }
                    }
                }
            }
        }
        if (lane == 0) atomicAdd(&rep->bursts, (unsigned long long)iters);
    } else {
        bf16x8 fa, fb;
#pragma unroll
        for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(s[40 + i] * 0.25f); fb[i] = (__bf16)(s[60 + i] * 0.25f); }
        f32x4 macc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) macc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        for (int it = 0; it < iters; ++it) {
            __builtin_amdgcn_s_barrier();
            if constexpr (BV != 0) {
#pragma unroll
                for (int d = 0; d < DELAY; ++d) asm volatile("s_nop 15");
                if constexpr (BV != 2) __builtin_amdgcn_s_setprio(1);
                if constexpr (BV == 1 || BV == 2) {
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int i = 0; i < 8; ++i) macc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, macc[i], 0, 0, 0);
                } else if constexpr (BV == 3) {
#pragma unroll
                    for (int k = 0; k < 8; ++k)
#pragma unroll
                        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(macc[i][k & 3]) : "v"(macc[(i + 1) & 7][0]), "v"(macc[(i + 2) & 7][1]));
                } else if constexpr (BV == 6) {
                    typedef float f32x16 __attribute__((ext_vector_type(16)));
                    f32x16 big = {};
#pragma unroll
                    for (int k = 0; k < 8; ++k) big = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, big, 0, 0, 0);
                    macc[0][0] += big[0] + big[15];
                } else if constexpr (BV == 5) {
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int i = 0; i < 8; ++i) macc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(macc[(i + 1) & 7][0], macc[(i + 2) & 7][1], macc[i], 0, 0, 0);
                }
                if constexpr (BV != 2) __builtin_amdgcn_s_setprio(0);
            }
        }
        float sum = 0.f;                                 // keep the MFMA results alive
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += macc[i][0] + macc[i][3];
        if (sum == 123.456f) rep->samples = 0xffffffffu;
    }
}

template <int AV, int BV, int DELAY, bool LDSRD>
void run(const float* seed, Report* rep, double seconds, const char* what) {
    hipMemset(rep, 0, sizeof(Report));
    const int iters = 20000;
    auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    do {
        hipLaunchKernelGGL((probe<AV, BV, DELAY, LDSRD>), dim3(256), dim3(512), 0, 0, seed, rep, iters);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(2); }
        ++launches;
    } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds);
    Report h;
    hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long total = 0;
    for (int g = 0; g < 4; ++g) total += (unsigned long long)h.bad[g][0] + h.bad[g][1];
    printf("%-78s %8.3g bursts: wrong results (16-lane group [lo,hi] half):", what, (double)h.bursts);
    for (int g = 0; g < 4; ++g) printf(" g%d[%u,%u]", g, h.bad[g][0], h.bad[g][1]);
    if (total) {
        printf("  by result index:");
        for (int i = 0; i < 64; ++i) if (h.hist[i]) printf(" %d:%u", i, h.hist[i]);
    }
    printf("\n");
    for (unsigned i = 0; i < h.samples && i < 4; ++i)
        printf("    wg %u wave %u lane %u result %u iter %u: got %.9g want %.9g\n", h.sample[i].wg, h.sample[i].wave,
               h.sample[i].lane, h.sample[i].idx, h.sample[i].iter, h.sample[i].got, h.sample[i].want);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const bool quick = argc > 2 && argv[2][0] == 'q';        // control, the hazard, the fix (tests/test_gpu_hazard_probe.py)
    const size_t n = (size_t)256 * 512 * 96;
    float* hs = (float*)malloc(n * 4);
    unsigned x = 12345u;
    for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; hs[i] = ((x >> 8) & 0xffff) / 32768.0f - 1.0f; }
    float* seed; Report* rep;
    hipMalloc(&seed, n * 4); hipMalloc(&rep, sizeof(Report));
    hipMemcpy(seed, hs, n * 4, hipMemcpyHostToDevice);
#define RUN(AV, BV, DELAY, LDSRD, what) run<AV, BV, DELAY, LDSRD>(seed, rep, seconds, what)
    if (quick) {
        RUN(0, 0, 0, false, "CONTROL op_sel:[0,1,0] | siblings idle");
        RUN(0, 1, 0, false, "HAZARD op_sel:[0,1,0] | setprio + 16 MFMA bf16");
        RUN(0, 1, 2, false, "HAZARD op_sel:[0,1,0] | 32 cycles later: setprio + 16 MFMA bf16");
        RUN(5, 1, 0, false, "FIX no op_sel | setprio + 16 MFMA bf16");
        RUN(5, 1, 1, false, "FIX no op_sel | 16 cycles later: setprio + 16 MFMA bf16");
        RUN(5, 1, 2, false, "FIX no op_sel | 32 cycles later: setprio + 16 MFMA bf16");
        RUN(5, 1, 4, false, "FIX no op_sel | 64 cycles later: setprio + 16 MFMA bf16");
        return 0;
    }
    RUN(0, 0, 0, false, "pk_fma t0,t1,o0,o1 | siblings idle");
    RUN(0, 1, 0, false, "pk_fma t0,t1,o0,o1 | setprio + 16 MFMA bf16");
    RUN(0, 2, 0, false, "pk_fma t0,t1,o0,o1 | 16 MFMA bf16, no setprio");
    RUN(0, 3, 0, false, "pk_fma t0,t1,o0,o1 | setprio + 64 v_fma");
    RUN(0, 4, 0, false, "pk_fma t0,t1,o0,o1 | setprio only");
    RUN(0, 5, 0, false, "pk_fma t0,t1,o0,o1 | setprio + 16 MFMA f32 16x16x4");
    RUN(1, 1, 0, false, "v_fma (unpacked), consumer 4 later | setprio + 16 MFMA bf16");
    RUN(2, 1, 0, false, "pk_fma t0,t1,s_nop,o0,o1 | setprio + 16 MFMA bf16");
    RUN(3, 1, 0, false, "pk_fma 32 producers then 32 consumers | setprio + 16 MFMA bf16");
    RUN(4, 1, 0, false, "compiler-generated packed FMAs | setprio + 16 MFMA bf16");
    RUN(5, 1, 0, false, "FIX: opaque splats, pk_fma without op_sel | setprio + 16 MFMA bf16");
    RUN(5, 1, 1, false, "FIX: opaque splats, pk_fma without op_sel | 16 cycles later: setprio + 16 MFMA bf16");
    RUN(5, 1, 2, false, "FIX: opaque splats, pk_fma without op_sel | 32 cycles later: setprio + 16 MFMA bf16");
    RUN(5, 1, 4, false, "FIX: opaque splats, pk_fma without op_sel | 64 cycles later: setprio + 16 MFMA bf16");
    RUN(6, 1, 0, false, "v_pk_mul_f32 op_sel:[0,1] | setprio + 16 MFMA bf16");
    RUN(7, 1, 0, false, "v_pk_add_f32 op_sel:[0,1] | setprio + 16 MFMA bf16");
    RUN(8, 1, 0, false, "v_pk_fma_f32 op_sel:[1,0,0] | setprio + 16 MFMA bf16");
    RUN(9, 1, 0, false, "v_pk_fma_f32 op_sel:[0,0,1] | setprio + 16 MFMA bf16");
    RUN(0, 6, 0, false, "pk_fma t0,t1,o0,o1 | setprio + 8 MFMA bf16 32x32x16");
    RUN(10, 1, 0, false, "v_pk_fma_f32 op_sel_hi:[1,0,1] (high half <- src1 low register) | setprio + 16 MFMA bf16");
    RUN(10, 1, 1, false, "v_pk_fma_f32 op_sel_hi:[1,0,1] | 16 cycles later: setprio + 16 MFMA bf16");
    RUN(10, 1, 2, false, "v_pk_fma_f32 op_sel_hi:[1,0,1] | 32 cycles later: setprio + 16 MFMA bf16");
    RUN(11, 1, 0, false, "v_pk_fma_f32 op_sel_hi:[1,1,0] (high half <- src2 low register) | setprio + 16 MFMA bf16");
    RUN(0, 1, 1, false, "pk_fma t0,t1,o0,o1 | 16 cycles later: setprio + 16 MFMA bf16");
    RUN(0, 1, 2, false, "pk_fma t0,t1,o0,o1 | 32 cycles later: setprio + 16 MFMA bf16");
    RUN(0, 1, 4, false, "pk_fma t0,t1,o0,o1 | 64 cycles later: setprio + 16 MFMA bf16");
    RUN(0, 1, 8, false, "pk_fma t0,t1,o0,o1 | 128 cycles later: setprio + 16 MFMA bf16");
    RUN(0, 1, 0, true, "LDS reads, pk_fma t0,t1,o0,o1 | setprio + 16 MFMA bf16");
    RUN(0, 1, 4, true, "LDS reads, pk_fma t0,t1,o0,o1 | 64 cycles later: setprio + 16 MFMA bf16");
    RUN(0, 1, 8, true, "LDS reads, pk_fma t0,t1,o0,o1 | 128 cycles later: setprio + 16 MFMA bf16");
    RUN(0, 1, 12, true, "LDS reads, pk_fma t0,t1,o0,o1 | 192 cycles later: setprio + 16 MFMA bf16");
    RUN(1, 1, 8, true, "LDS reads, v_fma (unpacked) | 128 cycles later: setprio + 16 MFMA bf16");
    return 0;
}
