// launch_floor.hip — what does one dependent kernel boundary cost on this box, as a function of how the kernel is launched?
// build: hipcc -O3 --offload-arch=gfx950 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
struct Big { char pad[400]; float* p; int n; };
__global__ void k_big(Big b) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < b.n) b.p[i] += 1.f; }
__global__ void k_chain(const float* __restrict__ in, float* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = in[i] + 1.f;
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *a, *b; const int n = 256 * 256;
    CK(hipMalloc(&a, n * 4 * 64)); CK(hipMalloc(&b, n * 4 * 64));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    auto run = [&](const char* name, auto launch) -> int {
        for (int i = 0; i < 50; i++) launch(i);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; i++) launch(i);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %7.2f us per launch\n", name, ms * 1e3 / iters);
        return 0;
    };
    run("empty 1 WG", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); });
    run("empty 256 WG x 256", [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st); });
    run("touch 1 WG", [&](int) { hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st, a, 64); });
    run("touch 256 WG x 256", [&](int) { hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, st, a, n); });
    run("touch 256 WG x 1024", [&](int) { hipLaunchKernelGGL(k_touch, dim3(256), dim3(1024), 0, st, a, n * 4); });
    run("touch 2048 WG x 256", [&](int) { hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, st, a, n * 8); });
    run("big-args 256 WG x 256", [&](int) { Big g; g.p = a; g.n = n; hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, st, g); });
    run("chain a->b->a 256 WG x 256", [&](int i) { if (i & 1) hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, st, b, a, n);
                                                   else hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, st, a, b, n); });
    run("touch 256 WG x 256, 64 KB dyn LDS", [&](int) { hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 65536, st, a, n); });
    // the same through a captured graph of 100 launches
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 100; i++) {
            if (i & 1) hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, st, b, a, n);
            else hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, st, a, b, n);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 20; i++) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %7.2f us per launch\n", "graph of 100 chain kernels", ms * 1e3 / 2000);
    }
    // NULL stream for comparison
    {
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, 0, a, n);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, 0, a, n);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %7.2f us per launch\n", "touch 256 WG x 256 on the NULL stream", ms * 1e3 / iters);
    }
    return 0;
}
