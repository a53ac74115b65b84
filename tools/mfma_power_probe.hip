// Is the matrix pipe's sustained rate on MI355X a matter of the clock the power limit leaves?  Register-resident MFMA loops (no memory
// traffic in the loop), 8 waves per CU on every CU, operands random / zero, three instructions of the same FLOP count per issue slot:
//   v_mfma_f32_16x16x32_f16 (16 K FLOP, 8 operand VGPRs read), v_mfma_f32_32x32x16_f16 (32 K FLOP, 8 operand VGPRs read: half the operand
//   reads per FLOP), v_mfma_scale_f32_16x16x128_f8f6f4 (64 K FLOP at e4m3).   Prints TFLOP/s = what the clock under that load allows.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o build/mfma_power_probe && build/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i8v __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(const uint32_t* src, float* out, int iters) {
    const int t = blockIdx.x * 512 + threadIdx.x;
    union { h8 h; uint32_t u[4]; } a[4], b[4];
    union { i8v v; uint32_t u[8]; } a8[2], b8[2];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) { a[i].u[r] = src[(t * 64 + i * 4 + r) & 0xFFFFF]; b[i].u[r] = src[(t * 64 + 16 + i * 4 + r) & 0xFFFFF]; }
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 8; ++r) { a8[i].u[r] = src[(t * 64 + 32 + i * 8 + r) & 0xFFFFF] & 0x77777777u; b8[i].u[r] = src[(t * 64 + 48 + i * 8 + r) & 0xFFFFF] & 0x77777777u; }
    float sum = 0.f;
    if (MODE == 0) {
        f4 c[16];
        for (int i = 0; i < 16; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3].h, b[(i >> 2) & 3].h, c[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) sum += c[i][0] + c[i][3];
    } else if (MODE == 3 || MODE == 4 || MODE == 5) {
        f4 c[16];
        for (int i = 0; i < 16; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ia = MODE == 3 ? 0 : (MODE == 4 ? (i >> 2) : (i & 3)), ib = MODE == 3 ? 0 : (MODE == 4 ? (i & 3) : ((i + (i >> 2)) & 3));
                c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ia].h, b[ib].h, c[i], 0, 0, 0);
            }
        }
        for (int i = 0; i < 16; ++i) sum += c[i][0] + c[i][3];
    } else if (MODE == 1) {
        f16v c[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3].h, b[(i >> 1) & 3].h, c[i & 3], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) sum += c[i][0] + c[i][15];
    } else if (MODE == 6) {      // e4m3 x e5m2 (round 4: the activation image of the correction product as bf8)
        f4 c[16];
        for (int i = 0; i < 16; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[i & 1].v, b8[(i >> 1) & 1].v, c[i], 0, 1, 0, 127, 0, 127);
        }
        for (int i = 0; i < 4; ++i) sum += c[i][0] + c[i][3];
    } else {
        f4 c[16];
        for (int i = 0; i < 16; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[i & 1].v, b8[(i >> 1) & 1].v, c[i], 0, 0, 0, 127, 0, 127);
        }
        for (int i = 0; i < 4; ++i) sum += c[i][0] + c[i][3];
    }
    out[t] = sum;
}

int main() {
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<uint32_t> h(1 << 20);
    std::mt19937 rng(1);
    uint32_t* src; float* out;
    hipMalloc(&src, h.size() * 4); hipMalloc(&out, (size_t)cus * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int data = 0; data < 3; ++data) {
        // 0: random f16 bit patterns with the exponent kept small (finite, |x| < 2); 1: zeros; 2: random again (order check)
        for (auto& v : h) { const uint32_t r = rng(); v = data == 1 ? 0u : ((r & 0x83FF83FFu) | 0x38003800u); }
        hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 7; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) probe<0><<<cus, 512>>>(src, out, iters);
                else if (mode == 1) probe<1><<<cus, 512>>>(src, out, iters);
                else if (mode == 2) probe<2><<<cus, 512>>>(src, out, iters);
                else if (mode == 3) probe<3><<<cus, 512>>>(src, out, iters);
                else if (mode == 4) probe<4><<<cus, 512>>>(src, out, iters);
                else if (mode == 5) probe<5><<<cus, 512>>>(src, out, iters);
                else probe<6><<<cus, 512>>>(src, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (rep > 0 && ms < best) best = ms;       // first rep warms the clocks
            }
            const double flop_per_iter = (mode == 0 || (mode >= 3 && mode <= 5)) ? 16.0 * 16384 : (mode == 1 ? 8.0 * 32768 : 4.0 * 65536);
            static const char* names[7] = {"v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_f16", "v_mfma_scale_f32_16x16x128_f8f6f4", "16x16x32_f16, same operands every time",
                                           "16x16x32_f16, A kept over 4 / B cycles", "16x16x32_f16, A and B both change", "v_mfma_scale_f32_16x16x128 e4m3 x e5m2"};
            const double tf = flop_per_iter * iters * cus * 8 / (best * 1e-3) / 1e12;
            printf("%s operands  %-42s %8.2f ms  %7.1f TFLOP/s\n", data == 1 ? "zero  " : "random", names[mode], best, tf);
        }
    }
    return 0;
}
