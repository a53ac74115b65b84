// Per-CU store rate of a GEMM epilogue's f16 output by the SHAPE one 16-byte-per-lane store instruction covers (round 4): the LDS-free epilogue of
// gemm_w2f8_kernel writes 16 rows x 64 bytes per instruction (what the 16x16 accumulator layout gives after one v_permlane16_swap) and is stamped at
// ~11 B/clk/CU whether 256 or 64 CUs run it.  Is that the instruction's 16 half-written 128-byte lines, i.e. would 8 rows x 128 B or 4 rows x 256 B leave faster?
// Each 512-thread block writes `tiles` 256x256 f16 tiles of a [M][N] matrix exactly as the kernel's 8 waves do (wave tile 64 rows x 128 columns, 16 KiB, 16
// instructions), nothing else; cycles per tile by s_memtime (stores issued -> next tile; the last tile's drain included through vmcnt(0)).
//   hipcc --offload-arch=gfx950 -O2 tools/store_shape_probe.hip -o build/store_shape_probe && build/store_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// SHAPE 0: 16 rows x 64 B, 1: 8 rows x 128 B, 2: 4 rows x 256 B per instruction; AUX = cache-policy bits of the buffer store (0 default, 2 = nt)
template <int SHAPE, int AUX, bool GLOBAL>
__global__ __launch_bounds__(512, 1) void store_kernel(char* C, int M, int N, int tiles_n, int ntiles, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
    const size_t row_bytes = (size_t)N * 2;
    unsigned long long t0 = 0, t1 = 0;
    const u32x4 v = {0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    int done = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, ++done) {
        const int m0 = (t / tiles_n) * 256 + wr * 64, n0 = (t % tiles_n) * 256 + wc * 128;
        char* base = C + (size_t)m0 * row_bytes + (size_t)n0 * 2;
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, AUX == 16 ? 0x00027000 : 0x00020000);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            unsigned off;
            if (SHAPE == 0) off = (unsigned)(((k >> 2) * 16 + (lane & 15)) * row_bytes) + (k & 3) * 64 + (lane >> 4) * 16;        // pass k >> 2, 64-byte quarter k & 3 of the 256-byte row
            else if (SHAPE == 1) off = (unsigned)(((k >> 1) * 8 + (lane >> 3)) * row_bytes) + (k & 1) * 128 + (lane & 7) * 16;
            else off = (unsigned)((k * 4 + (lane >> 4)) * row_bytes) + (lane & 15) * 16;
            if (GLOBAL) *(u32x4*)(base + off) = v;
            else __builtin_amdgcn_raw_buffer_store_b128(v, rc, (int)off, 0, AUX & 15);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = done; }
}

template <int SHAPE, int AUX, bool GLOBAL>
int run(const char* name, char* C, int M, int N, int grid, unsigned long long* dcyc) {
    const int tiles_n = N / 256, ntiles = (M / 256) * tiles_n;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned long long> h(2 * grid);
    float best = 1e9f; double cpt = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        store_kernel<SHAPE, AUX, GLOBAL><<<grid, 512>>>(C, M, N, tiles_n, ntiles, dcyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) {
            best = ms;
            CK(hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> per(grid);
            for (int b = 0; b < grid; ++b) per[b] = (double)h[2 * b] / (double)std::max<unsigned long long>(h[2 * b + 1], 1);
            std::sort(per.begin(), per.end()); cpt = per[grid / 2];
        }
    }
    const double bytes = (double)M * N * 2;
    printf("%-44s N %5d grid %3d: %8.1f us  %6.2f TB/s chip  %7.0f cycles per 128 KiB tile = %5.1f B/clk/CU\n", name, N, grid, best * 1e3, bytes / best / 1e9, cpt, 131072.0 / cpt);
    return 0;
}

int main() {
    const int M = 102400;
    char* C; CK(hipMalloc(&C, (size_t)M * 3072 * 2));
    unsigned long long* dcyc; CK(hipMalloc(&dcyc, 2 * 256 * 8));
    for (int N : {2304, 3072, 768}) {
        for (int grid : {256, 64}) {
            if (run<0, 0, false>("16 rows x  64 B per instruction (buffer)", C, M, N, grid, dcyc)) return 1;
            if (run<1, 0, false>(" 8 rows x 128 B per instruction (buffer)", C, M, N, grid, dcyc)) return 1;
            if (run<2, 0, false>(" 4 rows x 256 B per instruction (buffer)", C, M, N, grid, dcyc)) return 1;
            if (run<0, 2, false>("16 rows x  64 B, nt", C, M, N, grid, dcyc)) return 1;
            if (run<1, 2, false>(" 8 rows x 128 B, nt", C, M, N, grid, dcyc)) return 1;
            if (run<0, 0, true>("16 rows x  64 B, global_store", C, M, N, grid, dcyc)) return 1;
            if (run<1, 0, true>(" 8 rows x 128 B, global_store", C, M, N, grid, dcyc)) return 1;
            if (run<2, 0, true>(" 4 rows x 256 B, global_store", C, M, N, grid, dcyc)) return 1;
            if (run<0, 16, false>("16 rows x  64 B, buffer, flags 0x00027000", C, M, N, grid, dcyc)) return 1;
            if (run<1, 16, false>(" 8 rows x 128 B, buffer, flags 0x00027000", C, M, N, grid, dcyc)) return 1;
        }
    }
    return 0;
}
