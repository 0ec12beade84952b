// Is hipGraph replay correct under AMD_DIRECT_DISPATCH=0 (ROCm 7.2, gfx950)?  paintmind_amd's graph-replayed decode loop is bit-identical to
// its eager loop with direct dispatch (the default) and wrong in almost every replay without it (tools/dispatch_mode_stress.py); this
// program takes the library out of the picture: chains of DEPENDENT kernels captured into graphs the way engine.hip does it (a private
// non-blocking capture stream, thread-local mode, one graph per segment, replayed back to back on another stream, a parameter block
// refreshed by a kernel in front of the first graph), checked against the same chain run eagerly.
//   hipcc --offload-arch=gfx950 -O2 -o tools/hwtests/graph_dispatch_mode tools/hwtests/graph_dispatch_mode.hip
//   ./graph_dispatch_mode ; AMD_DIRECT_DISPATCH=0 ./graph_dispatch_mode
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// x[i] = x[i] * a + b[i % 7] + param[0]: every kernel depends on the previous one through x; `work` makes it last a few microseconds
__global__ void step_kernel(unsigned* x, const unsigned* param, unsigned a, unsigned salt, int n, int work) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        unsigned v = x[i];
        for (int w = 0; w < work; ++w) v = v * a + salt + param[0] + (unsigned)w;
        x[i] = v;
    }
}
__global__ void copy_kernel(unsigned* dst, const unsigned* src, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
    const int n = 1 << 16, kernels_per_graph = 150, ngraphs = 8, calls = 40;
    unsigned *x, *xref, *x0, *param, *param_host;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&xref, n * 4)); CK(hipMalloc(&x0, n * 4)); CK(hipMalloc(&param, 16));
    CK(hipHostMalloc((void**)&param_host, 16 * 64, hipHostMallocDefault));
    std::vector<unsigned> h(n), hr(n);
    for (int i = 0; i < n; ++i) h[i] = i * 2654435761u;
    CK(hipMemcpy(x0, h.data(), n * 4, hipMemcpyHostToDevice));
    hipStream_t s, cap;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    auto chain = [&](hipStream_t on, unsigned* buf, int g) {
        for (int k = 0; k < kernels_per_graph; ++k)
            hipLaunchKernelGGL(step_kernel, dim3(64 + (k % 5) * 40), dim3(256), 0, on, buf, param, 1664525u + 2 * k, (unsigned)(g * 1000 + k), n, 8 + (k % 3) * 40);
    };
    std::vector<hipGraphExec_t> execs;
    for (int g = 0; g < ngraphs; ++g) {
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
        chain(cap, x, g);
        CK(hipStreamEndCapture(cap, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        CK(hipGraphDestroy(graph));
        execs.push_back(exec);
    }
    int bad = 0;
    for (int c = 0; c < calls; ++c) {
        unsigned* slot = param_host + 4 * (c % 64);
        slot[0] = 12345u + c;
        unsigned* slot_dev; CK(hipHostGetDevicePointer((void**)&slot_dev, slot, 0));
        // eager reference on xref
        hipLaunchKernelGGL(copy_kernel, dim3(1), dim3(64), 0, s, param, slot_dev, 4);
        hipLaunchKernelGGL(copy_kernel, dim3(64), dim3(256), 0, s, xref, x0, n);
        for (int g = 0; g < ngraphs; ++g) chain(s, xref, g);
        CK(hipStreamSynchronize(s));
        // graph replay on x
        hipLaunchKernelGGL(copy_kernel, dim3(1), dim3(64), 0, s, param, slot_dev, 4);
        hipLaunchKernelGGL(copy_kernel, dim3(64), dim3(256), 0, s, x, x0, n);
        for (int g = 0; g < ngraphs; ++g) CK(hipGraphLaunch(execs[g], s));
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), x, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), xref, n * 4, hipMemcpyDeviceToHost));
        int nd = 0;
        for (int i = 0; i < n; ++i) nd += h[i] != hr[i];
        if (nd) { ++bad; if (bad <= 3) printf("call %d: %d of %d words differ\n", c, nd, n); }
    }
    const char* e = getenv("AMD_DIRECT_DISPATCH");
    printf("AMD_DIRECT_DISPATCH=%s: %d of %d graph replays (%d graphs x %d dependent kernels) differ from the eager chain\n", e ? e : "unset", bad, calls,
           ngraphs, kernels_per_graph);
    return bad ? 1 : 0;
}
