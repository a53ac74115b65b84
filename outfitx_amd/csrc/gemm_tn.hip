// Weight-gradient (TN) GEMM of the training step; epilogue / split-K reduce shared through gemm_common.h.
#include "gemm_common.h"

namespace {
// ------------------------------------------------------------------------------------------------
// TN kernel (weight gradients of the training step): C[m, n] = sum_k A[k, m] * B[k, n] with BOTH operands row-major and
// the contraction index k as their ROW index — dW = dY^T X straight from the row-major activation / gradient copies,
// no transposed staging buffers.  256 x 256 x 64 tile, 8 waves (2 x 4), wave tile 128 x 64, two LDS stages, the main
// loop of gemm_big_kernel; what differs is the LDS image and the fragment reads:
//   * a k-tile of an operand is two [64 k][128 col] images in the guide's 8-row x 32-column sub-tile layout
//       off(k, ch) = 2048*(k>>3) + 512*(ch>>2) + 64*(k&7) + 16*((ch&3) ^ ((k>>2)&3))         (ch = 16-byte chunk of the row)
//     filled by LDS-DMA (1 KiB pieces = two sub-tiles; the XOR is applied to the SOURCE address);
//   * MFMA operands are gathered with ds_read_b64_tr_b16 (4 k-rows x 16 columns per 16-lane group, delivered
//     column-major): two reads per 16x16x32 operand, conflict-free on this image (each 32-lane half touches
//     8 rows x 32 B = all 64 banks once).
// Split-K over blockIdx (deterministic: fp32 slab planes + splitk_reduce) because dW tile grids are small (16-48 tiles)
// while K = live rows is deep.  The live row count may come from device memory (k_dev); rows of the last k-tile past
// it are zeroed in LDS, so operand buffers only need to be READABLE up to round_up(K, 64) rows.
struct TnArgs {
    const char* A; const char* B; float* C; float* slab; const int* k_dev;
    int M, N, K, lda, ldb, ldc, splits, tiles_m, tiles_n, nwg, group_m;
    int m_valid, n_valid, accumulate;      // store only C[:m_valid, :n_valid] (ldc may be < N); accumulate: C += result
};

template <typename T>
__global__ __launch_bounds__(512, 2) void gemm_tn_kernel(TnArgs p) {
    typedef typename OpT<T>::v8 v8;
    typedef typename OpT<T>::v4 v4;
    constexpr int IMG = 64 * 256, STAGE = 4 * IMG;       // A m-halves 0,1 | B n-halves 0,1
    extern __shared__ __attribute__((aligned(16))) char smem[];
    OFX_LDS char* lds = (OFX_LDS char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    int bid = blockIdx.x;
    {
        const int nx = 8, q = p.nwg / nx, r = p.nwg % nx, x = bid % nx, i = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int tiles = p.tiles_m * p.tiles_n;
    const int split = bid / tiles;
    int tm, tn;
    {
        const int t = bid - split * tiles;
        const int per_group = p.group_m * p.tiles_n;
        const int gidx = t / per_group, first = gidx * p.group_m;
        const int gm = min(p.group_m, p.tiles_m - first);
        const int r = t - gidx * per_group;
        tm = first + r % gm;
        tn = r / gm;
    }
    const int m0 = tm * 256, n0 = tn * 256;
    int K = p.K;
    if (p.k_dev) { const int kl = *p.k_dev; K = kl < K ? kl : K; }
    const int nkt = (K + 63) >> 6, per = (nkt + p.splits - 1) / p.splits;
    const int kt0 = split * per, kt1 = min(nkt, kt0 + per);

    // LDS-DMA pieces: wave w moves pieces w*4 .. w*4+3 of the 32 A pieces and of the 32 B pieces of a k-tile
    unsigned a_off[4], b_off[4];
    int a_dst[4], b_dst[4];
    {
        const int sub = lane >> 5, row7 = (lane & 31) >> 2, slot = lane & 3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pa = wave * 4 + q, img = pa >> 4, pc = pa & 15;
            const int rowblk = pc >> 1, chq = 2 * (pc & 1) + sub;
            const int row = 8 * rowblk + row7;
            const int col = chq * 32 + (slot ^ ((row >> 2) & 3)) * 8 + img * 128;
            a_off[q] = ((unsigned)row * p.lda + col) * 2;
            b_off[q] = ((unsigned)row * p.ldb + col) * 2;
            a_dst[q] = img * IMG + pc * 1024;
            b_dst[q] = (2 + img) * IMG + pc * 1024;
        }
    }
    const char* a_base = p.A + (size_t)m0 * 2;
    const char* b_base = p.B + (size_t)n0 * 2;
    const size_t a_kstep = (size_t)64 * p.lda * 2, b_kstep = (size_t)64 * p.ldb * 2;
    auto issue = [&](int kt, int stage) {
        OFX_LDS char* base = lds + stage * STAGE;
        const char* ak = a_base + kt * a_kstep;
        const char* bk = b_base + kt * b_kstep;
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(ak + a_off[q], base + a_dst[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(bk + b_off[q], base + b_dst[q]);
    };

    // transposed fragment reads: 16-lane group g = lane>>4 owns k = 8g .. 8g+7 of a 32-deep k-step; lane 4q+pp of the group
    // supplies the address of k-row (.. + q), columns 4pp .. 4pp+3 of the 16-column block
    const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
    const int lbase = 2048 * g + 64 * q4 + 32 * (g & 1) + 16 * (pp >> 1) + 8 * (pp & 1);
    const int a_frag = wr * IMG + lbase;
    const int b_frag = (2 + (wc >> 1)) * IMG + 1024 * (wc & 1) + lbase;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (kt0 < kt1) {
        v8 af[2][8], wf[2][4];
        issue(kt0, 0);
        issue(kt0 + 1 < kt1 ? kt0 + 1 : kt0, 1);
        for (int kt = kt0; kt < kt1; ++kt) {
            const int cur = (kt - kt0) & 1;
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            OFX_LDS char* base = lds + cur * STAGE;
            const int rem = K - kt * 64;
            if (rem < 64) {                                   // block-uniform: only the last k-tile of the last split
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                for (int idx = tid; idx < (64 - rem) * 64; idx += 512) {
                    const int r = rem + (idx >> 6), img = (idx >> 4) & 3, c = idx & 15;
                    *(OFX_LDS f32x4*)(base + img * IMG + 2048 * (r >> 3) + 512 * (c >> 2) + 64 * (r & 7) + 16 * (c & 3)) = z;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v4 lo = __builtin_bit_cast(v4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (OFX_LDS s16x4*)(base + ((b_frag ^ (32 * (j & 1))) + 8192 * ks + 512 * (j >> 1)))));
                    v4 hi = __builtin_bit_cast(v4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (OFX_LDS s16x4*)(base + ((b_frag ^ (32 * (j & 1)) ^ 16) + 8192 * ks + 512 * (j >> 1) + 256))));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { wf[ks][j][e] = lo[e]; wf[ks][j][4 + e] = hi[e]; }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v4 lo = __builtin_bit_cast(v4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (OFX_LDS s16x4*)(base + ((a_frag ^ (32 * (i & 1))) + 8192 * ks + 512 * (i >> 1)))));
                    v4 hi = __builtin_bit_cast(v4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (OFX_LDS s16x4*)(base + ((a_frag ^ (32 * (i & 1)) ^ 16) + 8192 * ks + 512 * (i >> 1) + 256))));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { af[ks][i][e] = lo[e]; af[ks][i][4 + e] = hi[e]; }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // every wave holds its fragments: the stage is free
            const int kn = kt + 2 < kt1 ? kt + 2 : kt1 - 1;    // clamped: a redundant refill of a stage nobody reads again
            const char* ak = a_base + kn * a_kstep;
            const char* bk = b_base + kn * b_kstep;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int m = 0; m < 64; ++m) {
                if (m % 5 == 0 && m / 5 < 8) {
                    const int q = m / 5;
                    if (q < 4) glds16(ak + a_off[q], base + a_dst[q]);
                    else glds16(bk + b_off[q - 4], base + b_dst[q - 4]);
                }
                const int ks = m >> 5, i = (m >> 2) & 7, j = m & 3;
                acc[i][j] = OpT<T>::mfma16(wf[ks][j], af[ks][i], acc[i][j]);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    KArgs q{};
    q.C = (char*)(p.splits > 1 ? p.slab + (size_t)split * p.M * p.N : p.C);
    q.M = p.splits > 1 ? p.M : p.m_valid; q.N = p.N; q.ldc = p.splits > 1 ? p.N : p.ldc; q.out_kind = 0;
    q.n_valid = p.splits > 1 ? p.N : p.n_valid;
    if (p.splits == 1 && p.accumulate) { q.resid = p.C; q.ldr = p.ldc; }
    OFX_LDS char* ep = lds + 2 * STAGE + wave * EPI2_BYTES_PER_WAVE;
    epilogue2<T, OFX_ACT_NONE>(q, ep, acc, m0 + wr * 128, n0 + wc * 64, lane);
}

}  // namespace

// ---- TN GEMM (wgrad): split plan + launcher
int ofx_gemm_tn_splits(int M, int N, int K) {
    // cost model (us): waves of blocks x k-tiles per split x ~1.6 us per 256x256x64 k-tile, plus the slab written once and
    // read once at ~4 TB/s.  K is the static upper bound of the live row count, so the plan is shape-only (graph-safe).
    const long tiles = (long)((M + 255) / 256) * (N / 256);
    const int nkt = (K + 63) / 64;
    int best = 1;
    double best_t = 1e30;
    for (int s = 1; s <= 16 && s <= nkt; ++s) {
        const double waves = (double)((tiles * s + 255) / 256);
        const double t = waves * ((nkt + s - 1) / s) * 1.6 + 4.0 + (s > 1 ? 2.0 * s * M * N * 4.0 / 4.0e6 + 3.0 : 0.0);
        if (t < best_t) { best_t = t; best = s; }
    }
    return best;
}
size_t ofx_gemm_tn_slab_bytes(int M, int N, int K) {
    const int s = ofx_gemm_tn_splits(M, N, K);
    return s > 1 ? (size_t)s * M * N * 4 : 0;
}
int ofx_launch_gemm_tn(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K, const int* k_dev,
                       void* slab, size_t slab_bytes, int op_dtype, hipStream_t s, int m_valid, int n_valid, int accumulate) {
    if (m_valid <= 0) m_valid = M;
    if (n_valid <= 0) n_valid = N;
    OFX_REQUIRE(M > 0 && N > 0 && K > 0, OFX_ESHAPE, "gemm_tn: empty problem M=%d N=%d K=%d", M, N, K);
    OFX_REQUIRE(M % 256 == 0 && N % 256 == 0, OFX_ESHAPE, "gemm_tn: M=%d and N=%d must be multiples of 256", M, N);
    OFX_REQUIRE(lda >= M && ldb >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc >= n_valid && ldc % 4 == 0 && n_valid % 4 == 0 && m_valid <= M && n_valid <= N, OFX_ESHAPE,
                "gemm_tn: bad leading dimension / valid extent");
    OFX_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0), OFX_EINVAL, "gemm_tn: operands must be 16-byte aligned");
    OFX_REQUIRE(op_dtype == OFX_BF16 || op_dtype == OFX_F16, OFX_EINVAL, "gemm_tn: operand dtype must be bf16 or f16");
    TnArgs t;
    t.A = (const char*)A; t.B = (const char*)B; t.C = C; t.slab = (float*)slab; t.k_dev = k_dev;
    t.M = M; t.N = N; t.K = K; t.lda = lda; t.ldb = ldb; t.ldc = ldc; t.m_valid = m_valid; t.n_valid = n_valid; t.accumulate = accumulate;
    t.splits = slab ? ofx_gemm_tn_splits(M, N, K) : 1;
    if (t.splits > 1) OFX_REQUIRE(slab_bytes >= (size_t)t.splits * M * N * 4, OFX_EWORKSPACE, "gemm_tn: split-K slab too small");
    t.tiles_m = M / 256; t.tiles_n = N / 256; t.nwg = t.tiles_m * t.tiles_n * t.splits; t.group_m = 4;
    constexpr int LDSB = 2 * 4 * 64 * 256 + 8 * EPI2_BYTES_PER_WAVE;
    static DeviceOnce attr;
    TRY(attr.run([]() -> int {
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_tn_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_tn_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        return OFX_OK;
    }));
    ProfScope prof(PROF_GEMM, s, 2.0 * M * N * K);
    if (op_dtype == OFX_F16) hipLaunchKernelGGL(gemm_tn_kernel<f16_t>, dim3(t.nwg), dim3(512), LDSB, s, t);
    else hipLaunchKernelGGL(gemm_tn_kernel<bf16_t>, dim3(t.nwg), dim3(512), LDSB, s, t);
    if (t.splits > 1) {
        KArgs k{};
        k.C = (char*)C; k.M = m_valid; k.N = N; k.ldc = ldc; k.out_kind = 0; k.act = OFX_ACT_NONE; k.splits = t.splits; k.slab = (float*)slab; k.m_slab = M;
        k.n_valid = n_valid;
        if (accumulate) { k.resid = C; k.ldr = ldc; }
        size_t tot = (size_t)M * (N / 4);
        int rg = (int)((tot + 255) / 256); if (rg > 2048) rg = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3(rg), dim3(256), 0, s, k);
    }
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
