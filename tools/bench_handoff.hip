// Micro-benchmark (manual tool): neighbour-flag hand-off between workgroups inside ONE kernel versus kernel
// boundaries in a replayed graph.  Models the four dependent neighbour passes of a small channel: in every phase a
// workgroup publishes 256 x 32 bytes, then needs what the W workgroups on either side published in that phase.
//   fused  : one launch, per-(phase, workgroup) flags; data stored/loaded with agent-scope atomics (sc1), flag after a
//            drain + barrier; all workgroups must be co-resident (checked), every spin is bounded.
//   graph  : four plain kernels per step in a replayed hipGraph (plain loads / stores).
// hipcc -O3 --offload-arch=gfx950 bench_handoff.hip -o bench_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr int kB = 256, kPhases = 4, kW = 7;

__device__ __forceinline__ void st_sc1(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT));
}

// data[phase][blk*kB + t][4]
__global__ __launch_bounds__(kB) void k_fused(double *data, unsigned *flags, unsigned tag, int nb, int *fail, double *out)
{
    const int b = blockIdx.x, t = threadIdx.x;
    double acc = (double)(b + t);
    for (int ph = 0; ph < kPhases; ++ph) {
        double *mine = data + ((size_t)ph * nb * kB + (size_t)b * kB + t) * 4;
        for (int c = 0; c < 4; ++c) st_sc1(mine + c, acc + c);
        __builtin_amdgcn_s_waitcnt(0);  // drain this wave's stores
        __syncthreads();
        if (t == 0) __hip_atomic_store(&flags[ph * nb + b], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // wait for the 2W neighbours (periodic)
        if (t < 2 * kW) {
            int nbk = b + (t < kW ? -(t + 1) : (t - kW + 1));
            nbk = (nbk % nb + nb) % nb;
            unsigned spins = 0;
            while (__hip_atomic_load(&flags[ph * nb + nbk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { *fail = 1; break; }
            }
        }
        __syncthreads();
        // gather: every thread reads 8 records from neighbouring workgroups
        double s = 0.0;
        for (int j = 0; j < 8; ++j) {
            int nbk = ((b + (j - 4)) % nb + nb) % nb;
            const double *theirs = data + ((size_t)ph * nb * kB + (size_t)nbk * kB + ((t * 7 + j * 31) & (kB - 1))) * 4;
            s += ld_sc1(theirs) + ld_sc1(theirs + 1) + ld_sc1(theirs + 2) + ld_sc1(theirs + 3);
        }
        acc = acc * 0.5 + s * 1e-3;
    }
    out[(size_t)b * kB + t] = acc;
}

__global__ __launch_bounds__(kB) void k_phase(double *data, int ph, int nb, const double *in, double *out)
{
    const int b = blockIdx.x, t = threadIdx.x;
    double acc = in[(size_t)b * kB + t];
    if (ph > 0) {  // gather what the previous launch published
        double s = 0.0;
        for (int j = 0; j < 8; ++j) {
            int nbk = ((b + (j - 4)) % nb + nb) % nb;
            const double4 v = *reinterpret_cast<const double4 *>(data + ((size_t)(ph - 1) * nb * kB + (size_t)nbk * kB + ((t * 7 + j * 31) & (kB - 1))) * 4);
            s += v.x + v.y + v.z + v.w;
        }
        acc = acc * 0.5 + s * 1e-3;
    }
    *reinterpret_cast<double4 *>(data + ((size_t)ph * nb * kB + (size_t)b * kB + t) * 4) = make_double4(acc, acc + 1, acc + 2, acc + 3);
    out[(size_t)b * kB + t] = acc;
}

int main()
{
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_fused, kB, 0));
    printf("co-resident capacity %d workgroups\n", occ * prop.multiProcessorCount);
    for (int nb : {150, 600, 1000}) {
        if (nb > occ * prop.multiProcessorCount) continue;
        const size_t nd = (size_t)(kPhases + 1) * nb * kB * 4;
        double *data, *x, *y; unsigned *flags; int *fail;
        CK(hipMalloc(&data, nd * sizeof(double))); CK(hipMemset(data, 0, nd * sizeof(double)));
        CK(hipMalloc(&x, (size_t)nb * kB * sizeof(double))); CK(hipMalloc(&y, (size_t)nb * kB * sizeof(double)));
        CK(hipMemset(x, 0, (size_t)nb * kB * sizeof(double)));
        CK(hipMalloc(&flags, kPhases * nb * sizeof(unsigned))); CK(hipMemset(flags, 0, kPhases * nb * sizeof(unsigned)));
        CK(hipMalloc(&fail, sizeof(int))); CK(hipMemset(fail, 0, sizeof(int)));
        const int steps = 500;
        float ms;
        hipGraph_t g; hipGraphExec_t ge;
        // fused: one launch per step
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int it = 0; it < steps; ++it) hipLaunchKernelGGL(k_fused, dim3(nb), dim3(kB), 0, s, data, flags, (unsigned)(it + 1), nb, fail, y);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipMemsetAsync(flags, 0, kPhases * nb * sizeof(unsigned), s));
        CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, a, b));
        const double us_fused = 1e3 * ms / steps;
        int hf = 0; CK(hipMemcpy(&hf, fail, sizeof(int), hipMemcpyDeviceToHost));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        // graph: four launches per step
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int it = 0; it < steps; ++it)
            for (int ph = 0; ph < kPhases; ++ph) hipLaunchKernelGGL(k_phase, dim3(nb), dim3(kB), 0, s, data, ph, nb, (const double *)(ph ? y : x), y);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, a, b));
        const double us_graph = 1e3 * ms / steps;
        printf("workgroups %4d: fused (flags, sc1 data) %.2f us/step%s, four kernels in a graph %.2f us/step\n", nb, us_fused,
               hf ? " [SPIN LIMIT HIT]" : "", us_graph);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        CK(hipFree(data)); CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(flags)); CK(hipFree(fail));
    }
    return 0;
}
