// Micro-benchmark (manual tool): does a fork / join inside a replayed hipGraph hide launches that skip themselves?
// Models a device-decided step of a large channel: per step [pre] -> { long kernel  ||  N_EMPTY kernels that return at once } -> [post].
//   linear : everything captured on one stream (what launch_step_dyn does)
//   forked : the empty kernels captured on a second stream between an event fork and an event join
// hipcc -O3 --offload-arch=gfx950 bench_graph_fork.hip -o bench_graph_fork && ./bench_graph_fork [long_us] [n_empty] [steps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

__global__ void k_empty(const int *flag, int *sink)
{
    if (*flag) sink[blockIdx.x * blockDim.x + threadIdx.x] = 1;  // (never: the flag is 0)
}
// ~`iters` dependent FMAs per thread on a grid that fills the chip
__global__ void k_long(double *x, int iters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double a = x[i], b = 1.0000001;
    for (int k = 0; k < iters; ++k) a = fma(a, b, 1e-9);
    x[i] = a;
}
__global__ void k_small(double *x) { x[threadIdx.x] += 1.0; }

int main(int argc, char **argv)
{
    const double long_us = argc > 1 ? atof(argv[1]) : 300.0;
    const int n_empty = argc > 2 ? atoi(argv[2]) : 8, steps = argc > 3 ? atoi(argv[3]) : 40;
    const int nblk = 256 * 32, nthr = 256;
    double *x; int *flag, *sink;
    CK(hipMalloc(&x, sizeof(double) * nblk * nthr)); CK(hipMemset(x, 0, sizeof(double) * nblk * nthr));
    CK(hipMalloc(&flag, sizeof(int))); CK(hipMemset(flag, 0, sizeof(int)));
    CK(hipMalloc(&sink, sizeof(int) * 4096 * 256));
    hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t a, b, ef, ej; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    // calibrate the long kernel
    int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a, s)); hipLaunchKernelGGL(k_long, dim3(nblk), dim3(nthr), 0, s, x, iters); CK(hipEventRecord(b, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        iters = (int)(iters * long_us / (ms * 1e3) + 1);
    }
    for (int forked = 0; forked < 2; ++forked) {
        hipGraph_t g; hipGraphExec_t e;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int st = 0; st < steps; ++st) {
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, x);
            if (forked) {
                CK(hipEventRecord(ef, s)); CK(hipStreamWaitEvent(s2, ef, 0));
                for (int k = 0; k < n_empty; ++k) hipLaunchKernelGGL(k_empty, dim3(4096), dim3(256), 0, s2, (const int *)flag, sink);
                CK(hipEventRecord(ej, s2));
                hipLaunchKernelGGL(k_long, dim3(nblk), dim3(nthr), 0, s, x, iters);
                CK(hipStreamWaitEvent(s, ej, 0));
            } else {
                hipLaunchKernelGGL(k_long, dim3(nblk), dim3(nthr), 0, s, x, iters);
                for (int k = 0; k < n_empty; ++k) hipLaunchKernelGGL(k_empty, dim3(4096), dim3(256), 0, s, (const int *)flag, sink);
            }
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, x);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(e, s)); CK(hipStreamSynchronize(s));  // warm
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(a, s)); CK(hipGraphLaunch(e, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            best = ms < best ? ms : best;
        }
        printf("%s: long kernel ~%.0f us + %d self-skipping launches per step: %.1f us/step\n", forked ? "forked" : "linear", long_us, n_empty,
               1e3 * best / steps);
        CK(hipGraphExecDestroy(e)); CK(hipGraphDestroy(g));
    }
    // the long kernel alone, for reference
    {
        hipGraph_t g; hipGraphExec_t e;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int st = 0; st < steps; ++st) {
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, x);
            hipLaunchKernelGGL(k_long, dim3(nblk), dim3(nthr), 0, s, x, iters);
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, x);
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(e, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s)); CK(hipGraphLaunch(e, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("no self-skipping launches at all: %.1f us/step\n", 1e3 * ms / steps);
    }
    return 0;
}
