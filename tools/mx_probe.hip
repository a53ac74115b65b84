// Probe of the gfx950 instructions the fp8 weight-correction product uses (tools only; prints what the hardware does so that
// gemm_w2f8.hip is written against measured semantics, not assumed ones):
//   1. v_cvt_scalef32_pk_fp8_f16: which way the scale acts, saturation, which half of the destination is written;
//   2. v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3): the A/B lane -> (row, k) map, checked with exact integer data, and how the
//      E8M0 scale bytes act.
//   3. (round 4) the same instruction with the SECOND operand in e5m2 (blgp = 1; v_cvt_scalef32_pk_bf8_f16 makes it from f16): an
//      activation image that keeps f16's whole exponent range (no saturation below 57344, no scale to choose) at 3 significant bits.
//   hipcc --offload-arch=gfx950 -O2 tools/mx_probe.hip -o build/mx_probe && build/mx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

static float e4m3_to_float(uint8_t b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 0) v = ldexpf((float)m, -9);
    else if (e == 15 && m == 7) v = NAN;
    else v = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}

__global__ void cvt_kernel(const _Float16* x, float scale, uint32_t* out_lo, uint32_t* out_hi, int ovfl) {
    const int i = threadIdx.x;
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");      // MODE.FP16_OVFL: saturate instead of NaN
    h2 v = {x[2 * i], x[2 * i + 1]};
    s2 old = {(short)0x1111, (short)0x2222};
    s2 lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, v, scale, false);
    s2 hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, v, scale, true);
    out_lo[i] = ((uint32_t)(uint16_t)lo[1] << 16) | (uint16_t)lo[0];
    out_hi[i] = ((uint32_t)(uint16_t)hi[1] << 16) | (uint16_t)hi[0];
}

static float e5m2_to_float(uint8_t b) {
    const int s = b >> 7, e = (b >> 2) & 31, m = b & 3;
    float v;
    if (e == 0) v = ldexpf((float)m, -16);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf(1.0f + m / 4.0f, e - 15);
    return s ? -v : v;
}

__global__ void cvt_bf8_kernel(const _Float16* x, float scale, uint32_t* out_lo, uint32_t* out_hi, int ovfl) {
    const int i = threadIdx.x;
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    h2 v = {x[2 * i], x[2 * i + 1]};
    s2 old = {(short)0x1111, (short)0x2222};
    s2 lo = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(old, v, scale, false);
    s2 hi = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(old, v, scale, true);
    out_lo[i] = ((uint32_t)(uint16_t)lo[1] << 16) | (uint16_t)lo[0];
    out_hi[i] = ((uint32_t)(uint16_t)hi[1] << 16) | (uint16_t)hi[0];
}

// the same product with the second operand's bytes read as e5m2 (blgp = 1)
__global__ void mfma_b_bf8_kernel(const uint8_t* a_bytes, const uint8_t* b_bytes, float* d, int scale_a, int scale_b) {
    const int l = threadIdx.x;
    i8v a, b;
    for (int r = 0; r < 8; ++r) { a[r] = ((const int*)a_bytes)[l * 8 + r]; b[r] = ((const int*)b_bytes)[l * 8 + r]; }
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 1, 0, scale_a, 0, scale_b);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

// One wave: D[16x16] = A[16x128] . B[128x16] with operands given per lane as 32 bytes
__global__ void mfma_kernel(const uint8_t* a_bytes, const uint8_t* b_bytes, float* d, int scale_a, int scale_b) {
    const int l = threadIdx.x;
    i8v a, b;
    for (int r = 0; r < 8; ++r) { a[r] = ((const int*)a_bytes)[l * 8 + r]; b[r] = ((const int*)b_bytes)[l * 8 + r]; }
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

int main() {
    // ---- 1. conversion
    const float xs[16] = {1.0f, -1.5f, 0.3f, 448.f, 500.f, 1000.f, 0.0019f, 0.001f, 3.3f, 100.f, 0.0625f, 0.015625f, 7.7f, -0.2f, 65504.f, 1e-5f};
    _Float16 hx[16];
    for (int i = 0; i < 16; ++i) hx[i] = (_Float16)xs[i];
    _Float16* dx; uint32_t *dlo, *dhi;
    hipMalloc(&dx, sizeof(hx)); hipMalloc(&dlo, 32); hipMalloc(&dhi, 32);
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    for (int pass = 0; pass < 4; ++pass) {
        const float scale = pass == 1 ? 4.0f : (pass >= 2 ? 0.25f : 1.0f);
        if (pass == 3) printf("with MODE.FP16_OVFL = 1:\n");
        cvt_kernel<<<1, 8>>>(dx, scale, dlo, dhi, pass == 3);
        uint32_t lo[8], hi[8];
        hipMemcpy(lo, dlo, 32, hipMemcpyDeviceToHost); hipMemcpy(hi, dhi, 32, hipMemcpyDeviceToHost);
        printf("cvt scale %.2f:\n", scale);
        for (int i = 0; i < 8; ++i)
            printf("  x = (%g, %g)  dst_hi=0 -> %08x  [%g, %g]   dst_hi=1 -> %08x  [%g, %g]\n", xs[2 * i], xs[2 * i + 1], lo[i],
                   e4m3_to_float(lo[i] & 0xff), e4m3_to_float((lo[i] >> 8) & 0xff), hi[i], e4m3_to_float((hi[i] >> 16) & 0xff), e4m3_to_float(hi[i] >> 24));
    }
    // ---- 2. MFMA lane map with exact integers: A[m][k] = small ints, B[k][n] asymmetric
    std::vector<float> A(16 * 128), B(128 * 16);
    auto enc = [](int v) -> uint8_t {      // e4m3 of small integers |v| <= 8 (exact)
        const int s = v < 0; int a = s ? -v : v;
        if (a == 0) return 0;
        int e = 0; while ((a >> (e + 1)) != 0) ++e;             // floor(log2 a)
        const int m = (int)lroundf((a / (float)(1 << e) - 1.0f) * 8.0f);
        return (uint8_t)((s << 7) | ((e + 7) << 3) | m);
    };
    std::vector<int> Ai(16 * 128), Bi(128 * 16);
    for (int m = 0; m < 16; ++m) for (int k = 0; k < 128; ++k) Ai[m * 128 + k] = ((m * 7 + k * 3) % 9) - 4;
    for (int k = 0; k < 128; ++k) for (int n = 0; n < 16; ++n) Bi[k * 16 + n] = ((k * 5 + n * 11 + (k >> 5)) % 7) - 3;
    std::vector<double> ref(256, 0.0);
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { double s = 0; for (int k = 0; k < 128; ++k) s += Ai[m * 128 + k] * Bi[k * 16 + n]; ref[m * 16 + n] = s; }
    // hypothesis H1: lane l = (r = l & 15, g = l >> 4) holds k = 32 g + i in byte i, for both operands
    std::vector<uint8_t> ab(64 * 32), bb(64 * 32);
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 32; ++i) {
        const int r = l & 15, g = l >> 4, k = 32 * g + i;
        ab[l * 32 + i] = enc(Ai[r * 128 + k]);
        bb[l * 32 + i] = enc(Bi[k * 16 + r]);
    }
    uint8_t *da, *db; float* dd;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dd, 1024);
    hipMemcpy(da, ab.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, bb.data(), 2048, hipMemcpyHostToDevice);
    for (int sc = 0; sc < 3; ++sc) {
        const int sa = sc == 1 ? 126 : 127, sb = sc == 2 ? 129 : 127;
        mfma_kernel<<<1, 64>>>(da, db, dd, sa, sb);
        float d[256];
        hipMemcpy(d, dd, 1024, hipMemcpyDeviceToHost);
        // D layout (16x16): col = lane & 15, row = (lane >> 4) * 4 + reg
        int bad = 0; double ratio = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const int row = (l >> 4) * 4 + r, col = l & 15;
            const double want = ref[row * 16 + col];
            if (want != 0) ratio = d[l * 4 + r] / want;
            if (fabs(d[l * 4 + r] - want) > 1e-3) ++bad;
        }
        printf("mfma scale_a=%d scale_b=%d: %d of 256 differ from the H1 product (last ratio %.4f); d[0..3] = %g %g %g %g, ref %g %g %g %g\n", sa, sb, bad, ratio,
               d[0], d[1], d[2], d[3], ref[0], ref[16], ref[32], ref[48]);
    }
    // a k-permuted pairing: the same permutation on both operands must give the same product (what the kernel relies on)
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 32; ++i) {
        const int r = l & 15, g = l >> 4, s = i >> 3, j = i & 7, k = 32 * s + 8 * g + j;
        ab[l * 32 + i] = enc(Ai[r * 128 + k]);
        bb[l * 32 + i] = enc(Bi[k * 16 + r]);
    }
    hipMemcpy(da, ab.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, bb.data(), 2048, hipMemcpyHostToDevice);
    mfma_kernel<<<1, 64>>>(da, db, dd, 127, 127);
    float d[256];
    hipMemcpy(d, dd, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (fabs(d[l * 4 + r] - ref[((l >> 4) * 4 + r) * 16 + (l & 15)]) > 1e-3) ++bad;
    printf("k-permuted pairing (k = 32 s + 8 g + j at byte 8 s + j of lane group g): %d of 256 differ\n", bad);
    // ---- 3. e5m2 second operand: conversion, then the H1 product with B encoded as e5m2
    const float ys[16] = {1.0f, -1.5f, 0.3f, 448.f, 500.f, 1000.f, 0.0019f, 0.001f, 3.3f, 100.f, 57344.f, 60000.f, 65504.f, -0.2f, 6.0e-5f, 1e-5f};
    for (int i = 0; i < 16; ++i) hx[i] = (_Float16)ys[i];
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; ++pass) {
        cvt_bf8_kernel<<<1, 8>>>(dx, 1.0f, dlo, dhi, pass);
        uint32_t lo[8], hi[8];
        hipMemcpy(lo, dlo, 32, hipMemcpyDeviceToHost); hipMemcpy(hi, dhi, 32, hipMemcpyDeviceToHost);
        printf("cvt_bf8 scale 1, MODE.FP16_OVFL = %d:\n", pass);
        for (int i = 0; i < 8; ++i)
            printf("  x = (%g, %g)  dst_hi=0 -> %08x  [%g, %g]   dst_hi=1 -> %08x  [%g, %g]\n", (float)hx[2 * i], (float)hx[2 * i + 1], lo[i],
                   e5m2_to_float(lo[i] & 0xff), e5m2_to_float((lo[i] >> 8) & 0xff), hi[i], e5m2_to_float((hi[i] >> 16) & 0xff), e5m2_to_float(hi[i] >> 24));
    }
    auto enc5 = [](int v) -> uint8_t {     // e5m2 of small integers |v| <= 4 (exact: 1, 2, 3, 4 need <= 2 mantissa bits)
        const int s = v < 0; int a = s ? -v : v;
        if (a == 0) return 0;
        int e = 0; while ((a >> (e + 1)) != 0) ++e;
        const int m = (int)lroundf((a / (float)(1 << e) - 1.0f) * 4.0f);
        return (uint8_t)((s << 7) | ((e + 15) << 2) | m);
    };
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 32; ++i) {
        const int r = l & 15, g = l >> 4, k = 32 * g + i;
        ab[l * 32 + i] = enc(Ai[r * 128 + k]);
        bb[l * 32 + i] = enc5(Bi[k * 16 + r]);
    }
    hipMemcpy(da, ab.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, bb.data(), 2048, hipMemcpyHostToDevice);
    mfma_b_bf8_kernel<<<1, 64>>>(da, db, dd, 127, 127);
    hipMemcpy(d, dd, 1024, hipMemcpyDeviceToHost);
    bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (fabs(d[l * 4 + r] - ref[((l >> 4) * 4 + r) * 16 + (l & 15)]) > 1e-3) ++bad;
    printf("e4m3 x e5m2 (blgp = 1), H1 layout: %d of 256 differ\n", bad);
    return 0;
}
