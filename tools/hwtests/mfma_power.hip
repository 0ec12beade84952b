// Sustained MFMA rate under the chip's power management, from registers only (no memory traffic): 16x16x32 vs 32x32x16 bf16,
// random vs zero operands.   hipcc --offload-arch=gfx950 -O3 -o mfma_power tools/hwtests/mfma_power.hip && ./mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__global__ __launch_bounds__(512) void burn(const unsigned* seed, float* out, int iters) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned s = seed[t & 4095] * 2654435761u + t;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u;
            const float fa = seed[0] == 0 ? 0.f : ((s >> 8) & 0xffff) / 32768.f - 1.f;
            s = s * 1664525u + 1013904223u;
            const float fb = seed[0] == 0 ? 0.f : ((s >> 8) & 0xffff) / 32768.f - 1.f;
            a[i][j] = (__bf16)fa; b[i][j] = (__bf16)fb;
        }
    float r = 0.f;
    if constexpr (KIND == 0) {                       // 16 independent 16x16x32 accumulators (64 regs), 16 MFMAs per round
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
        for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][3];
    } else {                                         // 8 independent 32x32x16 accumulators (128 regs), 8 MFMAs per round = same flops
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][15];
    }
    out[t] = r;
}

int main() {
    const int blocks = 256, threads = 512, iters = 20000;           // one 8-wave workgroup per CU
    unsigned* seed; float* out;
    hipMalloc(&seed, 4096 * 4); hipMalloc(&out, blocks * threads * 4);
    unsigned h[4096];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero = 0; zero < 2; ++zero) {
        for (int i = 0; i < 4096; ++i) h[i] = zero ? 0u : (unsigned)rand() | 1u;
        hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
        for (int kind = 0; kind < 2; ++kind) {
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(burn<0>, dim3(blocks), dim3(threads), 0, 0, seed, out, iters);
                else hipLaunchKernelGGL(burn<1>, dim3(blocks), dim3(threads), 0, 0, seed, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double flops = (double)blocks * 8 /*waves*/ * iters * 16 * (2.0 * 16 * 16 * 32);
                printf("%s operands  %s  rep %d: %7.3f ms  %7.1f TFLOP/s\n", zero ? "zero  " : "random", kind ? "32x32x16" : "16x16x32", rep, ms, flops / ms / 1e9);
            }
        }
    }
    return 0;
}
