// Cost of a grid-wide barrier on MI355X (cooperative-launch grid.sync vs a hand-rolled agent-scope atomic counter) next to an empty
// back-to-back launch: the measurement behind "no persistent set-transformer kernel" in DESIGN.md section 9.
//   hipcc --offload-arch=gfx950 -O3 -o build/grid_barrier_cost tools/grid_barrier_cost.hip && ./build/grid_barrier_cost
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;

__global__ void k_coop(int n, float* out) {
    cg::grid_group g = cg::this_grid();
    float v = threadIdx.x;
    for (int i = 0; i < n; ++i) { v = v * 1.0001f + 1.f; g.sync(); }
    if (v == 12345.f) out[0] = v;
}
// hand-rolled: monotonically increasing counter, agent-scope atomics, one thread per block spins
__global__ void k_atomic(int n, unsigned* cnt, float* out) {
    float v = threadIdx.x;
    const unsigned nb = gridDim.x;
    for (int i = 0; i < n; ++i) {
        v = v * 1.0001f + 1.f;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = nb * (unsigned)(i + 1);
            int spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
            __threadfence();
        }
        __syncthreads();
    }
    if (v == 12345.f) out[0] = v;
}
__global__ void k_empty(float* out) { if (threadIdx.x == 9999) out[0] = 1.f; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    float* out; unsigned* cnt;
    CK(hipMalloc(&out, 4)); CK(hipMalloc(&cnt, 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int dev_coop = 0; CK(hipDeviceGetAttribute(&dev_coop, hipDeviceAttributeCooperativeLaunch, 0));
    printf("cooperative launch supported: %d\n", dev_coop);
    for (int nb : {256, 512}) {
        for (int n : {1, 101}) {
            float ms;
            void* args[] = {&n, &out};
            CK(hipLaunchCooperativeKernel((void*)k_coop, dim3(nb), dim3(256), args, 0, s)); CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < 10; ++r) CK(hipLaunchCooperativeKernel((void*)k_coop, dim3(nb), dim3(256), args, 0, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("coop   blocks %d syncs %3d: %.2f us per launch\n", nb, n, ms * 100);
            CK(hipMemsetAsync(cnt, 0, 4, s));
            hipLaunchKernelGGL(k_atomic, dim3(nb), dim3(256), 0, s, n, cnt, out); CK(hipStreamSynchronize(s));
            float tot = 0;
            for (int r = 0; r < 10; ++r) {
                CK(hipMemsetAsync(cnt, 0, 4, s));
                CK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(k_atomic, dim3(nb), dim3(256), 0, s, n, cnt, out);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
            }
            printf("atomic blocks %d syncs %3d: %.2f us per launch\n", nb, n, tot * 100);
        }
    }
    {
        float ms;
        hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, out); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 100; ++r) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, out);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty kernel back to back: %.2f us per launch\n", ms * 10);
    }
    return 0;
}
