// What v_permlane16_swap does on gfx950 (the register-to-register exchange the LDS-free GEMM epilogue of round 4 relies on):
// X = value 1000 + lane, Y = value 2000 + lane; prints both after __builtin_amdgcn_permlane16_swap(X, Y, false, false).
//   hipcc --offload-arch=gfx950 -O2 tools/permlane_probe.hip -o build/permlane_probe && build/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    u2 r = __builtin_amdgcn_permlane16_swap(1000u + l, 2000u + l, false, false);
    out[l] = r[0]; out[64 + l] = r[1];
}
int main() {
    unsigned* d; unsigned h[128];
    (void)hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int r = 0; r < 4; ++r) printf("lanes %2d-%2d: first result %u..%u   second result %u..%u\n", 16 * r, 16 * r + 15, h[16 * r], h[16 * r + 15], h[64 + 16 * r], h[64 + 16 * r + 15]);
    // expected by the epilogue: first = [X row 0, Y row 0, X row 2, Y row 2], second = [X row 1, Y row 1, X row 3, Y row 3]
    bool ok = true;
    for (int l = 0; l < 64; ++l) {
        const int row = l >> 4, c = l & 15;
        const unsigned want0 = (row & 1) ? 2000u + (row - 1) * 16 + c : 1000u + row * 16 + c;
        const unsigned want1 = (row & 1) ? 2000u + row * 16 + c : 1000u + (row + 1) * 16 + c;
        ok = ok && h[l] == want0 && h[64 + l] == want1;
    }
    printf("matches the layout the epilogue assumes: %s\n", ok ? "yes" : "NO");
    return ok ? 0 : 1;
}
