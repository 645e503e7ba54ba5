// Micro-benchmark (manual tool, not part of the library): what does a grid-wide barrier cost on MI355X compared
// with a kernel boundary inside a replayed hipGraph?  Decides whether fusing the four neighbour passes of a
// small channel into one persistent kernel can pay.   hipcc -O3 --offload-arch=gfx950 bench_gridsync.hip
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

__global__ void k_empty(double *p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0; }

// sense-reversing barrier on one counter: the last arriver resets the count and bumps the generation
struct Bar { unsigned count; unsigned gen; unsigned fail; unsigned pad; };
__device__ __forceinline__ void grid_barrier(Bar *b, unsigned nblocks)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = __hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        const unsigned arrived = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == nblocks - 1) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000000u) { b->fail = 1; break; }  // never hang the box
            }
        }
        __threadfence();
    }
    __syncthreads();
}

__global__ void k_manual(Bar *b, double *p, int n, int iters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (i < n) p[i] += 1.0;
        grid_barrier(b, gridDim.x);
    }
}

__global__ void k_coop(double *p, int n, int iters)
{
    cg::grid_group grid = cg::this_grid();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (i < n) p[i] += 1.0;
        grid.sync();
    }
}

int main(int argc, char **argv)
{
    const int iters = 2000;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    int occ_m = 0, occ_c = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_m, k_manual, 256, 0));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_c, k_coop, 256, 0));
    printf("CUs %d, co-resident blocks/CU manual %d coop %d\n", prop.multiProcessorCount, occ_m, occ_c);
    for (int blocks : {32, 150, 300, 600, 1024}) {
        const int n = blocks * 256;
        double *p; CK(hipMalloc(&p, n * sizeof(double))); CK(hipMemset(p, 0, n * sizeof(double)));
        Bar *bar; CK(hipMalloc(&bar, sizeof(Bar))); CK(hipMemset(bar, 0, sizeof(Bar)));
        float ms;
        // (a) kernel boundaries inside a replayed graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, s, p, n);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, a, b));
        const double us_graph = 1e3 * ms / iters;
        // (b) hand-rolled barrier (grid must be co-resident)
        double us_manual = -1, us_coop = -1;
        if (blocks <= occ_m * prop.multiProcessorCount) {
            hipLaunchKernelGGL(k_manual, dim3(blocks), dim3(256), 0, s, bar, p, n, 10); CK(hipStreamSynchronize(s));
            CK(hipEventRecord(a, s));
            hipLaunchKernelGGL(k_manual, dim3(blocks), dim3(256), 0, s, bar, p, n, iters);
            CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, a, b));
            us_manual = 1e3 * ms / iters;
            Bar hb; CK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
            if (hb.fail) { printf("manual barrier TIMED OUT at %d blocks\n", blocks); us_manual = -2; }
        }
        // (c) cooperative groups grid.sync()
        if (blocks <= occ_c * prop.multiProcessorCount) {
            int it1 = 10, it2 = iters; int nn = n;
            void *args1[] = {&p, &nn, &it1}; void *args2[] = {&p, &nn, &it2};
            CK(hipLaunchCooperativeKernel((void *)k_coop, dim3(blocks), dim3(256), args1, 0, s)); CK(hipStreamSynchronize(s));
            CK(hipEventRecord(a, s));
            CK(hipLaunchCooperativeKernel((void *)k_coop, dim3(blocks), dim3(256), args2, 0, s));
            CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, a, b));
            us_coop = 1e3 * ms / iters;
        }
        printf("blocks %5d: graph kernel boundary %.2f us, manual barrier %.2f us, cg grid.sync %.2f us\n", blocks, us_graph,
               us_manual, us_coop);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(p)); CK(hipFree(bar));
    }
    return 0;
}
