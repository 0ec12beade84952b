// Hardware question: do a wave's vector-memory operations retire from vmcnt IN ISSUE ORDER when loads and stores are mixed?
// gemm256 relies on it at the tile boundary: the next tile's DMA is issued BEFORE the epilogue's stores, and
// `s_waitcnt vmcnt(S)` (S = number of younger stores) is meant to prove that the older DMA has landed.
// Test: each wave issues one COLD load (a line never touched, far stride: HBM + TLB miss), then S stores to a hot line
// (L2 hits, fast), then `s_waitcnt vmcnt(S)` and consumes the loaded register with no further wait.  If stores could retire
// ahead of the older load, the counter would drop to S while the load is still in flight and the register would still hold
// the sentinel.   hipcc --offload-arch=gfx950 -O2 -o vmcnt_order tools/hwtests/vmcnt_order.hip && ./vmcnt_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int S, int W>
__global__ void probe(const unsigned* cold, unsigned* hot, unsigned* out, size_t stride_words, int rounds) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned bad = 0;
    for (int r = 0; r < rounds; ++r) {
        const unsigned* src = cold + ((size_t)tid * rounds + r) * stride_words;     // a fresh line per lane per round
        unsigned* dst = hot + (tid & 1023);
        unsigned val = 0xdeadbeefu;
        asm volatile("global_load_dword %0, %1, off" : "+v"(val) : "v"(src) : "memory");
#pragma unroll
        for (int i = 0; i < S; ++i) asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(tid + i) : "memory");
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(val) : "n"(W));
        unsigned seen;
        asm volatile("v_mov_b32 %0, %1" : "=v"(seen) : "v"(val));
        bad += (seen != 0x12345678u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    out[tid] = bad;
}

int main() {
    const int blocks = 512, threads = 256, rounds = 8;
    const size_t stride_words = 4096 / 4 * 3;                          // 12 KiB apart: new page / line every access
    const size_t n = (size_t)blocks * threads * rounds * stride_words;
    unsigned *cold, *hot, *out;
    if (hipMalloc(&cold, n * 4) != hipSuccess) { printf("alloc failed\n"); return 2; }
    hipMalloc(&hot, 1024 * 4); hipMalloc(&out, blocks * threads * 4);
    hipMemsetD32((hipDeviceptr_t)cold, 0x12345678u, n);
    hipDeviceSynchronize();
    unsigned* h = (unsigned*)malloc(blocks * threads * 4);
    long total = 0;
    auto run = [&](const char* name, auto kernel, bool must_pass) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, cold, hot, out, stride_words, rounds);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, out, blocks * threads * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < blocks * threads; ++i) bad += h[i];
        printf("%-34s consumed the load before it landed: %ld of %ld\n", name, bad, (long)blocks * threads * rounds);
        if (must_pass) total += bad;
        return bad;
    };
    long control = 0;
    for (int rep = 0; rep < 4; ++rep) {
        run("S=16 stores, vmcnt(16)", probe<16, 16>, true);
        run("S=32 stores, vmcnt(32)", probe<32, 32>, true);
        run("S=4 stores,  vmcnt(4)", probe<4, 4>, true);
        run("S=48 stores, vmcnt(48)", probe<48, 48>, true);
        control += run("control: S=16, vmcnt(17) (too weak)", probe<16, 17>, false);
    }
    printf("control detected %ld early reads (must be > 0 for the probe to mean anything)\n", control);
    printf("total bad %ld\n", total);
    return (total || !control) ? 1 : 0;
}
