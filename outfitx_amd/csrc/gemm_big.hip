// 256x256 (8 waves, two LDS stages) and 256x128 (4 waves, register-resident k-tile) tile kernels; see gemm.hip for the
// dispatcher and DESIGN.md §4 for the measurements that shaped them.
#include "gemm_common.h"

namespace {
// WR x WC waves, wave tile 128 x 64, NST LDS stages.  <2,4,2> = 256x256 tile, 8 waves, one block per CU;
// <2,2,1> = 256x128 tile, 4 waves, LDS is a single landing stage (the k-tile being multiplied lives in
// registers), 64 KiB per block so TWO independent blocks share a CU and overlap each other's load phases.
template <typename T, int WR, int WC, int NST, int ABL = 0>   // ABL (diagnostics, wrong results): 1 no LDS-DMA, 2 no fragment reads after k-tile 0, 3 both, 4 both + no barriers, 5 every k-tile re-reads k-slice 0 (cache-resident operands)
__global__ __launch_bounds__(64 * WR * WC, 2) void gemm_big_kernel(KArgs p) {
    typedef typename OpT<T>::v8 v8;
    constexpr int TM = WR * 128, TN = WC * 64, NW = WR * WC;
    constexpr int STAGE = (TM + TN) * BK * 2;
    constexpr int A_PER_WAVE = (TM / 8) / NW, W_PER_WAVE = (TN / 8) / NW, NLD = A_PER_WAVE + W_PER_WAVE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    OFX_LDS char* lds = (OFX_LDS char*)smem;
    const unsigned long long cstart = p.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    // One-time phase skew: the second block that lands on each CU (dispatch ids 256..511) starts `skew` x ~4 us late,
    // so the two co-resident blocks alternate main loop / epilogue instead of bursting their stores together.
    // Speed only: nothing depends on which blocks actually share a CU.
    if (p.skew > 0 && NST == 1 && blockIdx.x >= 256 && blockIdx.x < 512)
        for (int i = 0; i < p.skew; ++i) __builtin_amdgcn_s_sleep(127);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;

    int bid = blockIdx.x;
    {
        const int nx = 8, q = p.nwg / nx, r = p.nwg % nx, x = bid % nx, i = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    int tm, tn;
    {
        const int per_group = p.group_m * p.tiles_n;
        const int gidx = bid / per_group, first = gidx * p.group_m;
        const int gm = min(p.group_m, p.tiles_m - first);
        const int r = bid - gidx * per_group;
        tm = first + r % gm;
        tn = r / gm;
    }
    const int m0 = tm * TM, n0 = tn * TN;
    if (p.m_dev) {
        const int m_live = *p.m_dev;
        p.M = m_live < p.M ? m_live : p.M;
        if (m0 >= p.M) return;
    }

    // LDS-DMA: chunks of 1 KiB = 8 rows x 128 B, swizzle on the source address
    const int lrow = lane >> 3, lchk = lane & 7;
    // uniform 64-bit tile bases + 32-bit per-lane offsets (saddr form: keeps 12 address VGPRs instead of 24)
    const char* a_base = p.A + (size_t)m0 * p.lda * 2;
    const char* w_base = p.W + (size_t)n0 * p.K * 2;
    unsigned a_off[A_PER_WAVE], w_off[W_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
        const int row = (wave * A_PER_WAVE + i) * 8 + lrow;
        const int rr = m0 + row < p.M ? row : p.M - 1 - m0;          // clamp rows past M onto the last live row
        a_off[i] = ((unsigned)rr * p.lda + (lchk ^ ((row >> 1) & 7)) * 8) * 2;
    }
#pragma unroll
    for (int i = 0; i < W_PER_WAVE; ++i) {
        const int row = (wave * W_PER_WAVE + i) * 8 + lrow;
        w_off[i] = ((unsigned)row * p.K + (lchk ^ ((row >> 1) & 7)) * 8) * 2;
    }
    const int a_dst = wave * A_PER_WAVE * 1024, w_dst = TM * BK * 2 + wave * W_PER_WAVE * 1024;
    auto issue = [&](int kt, int stage) {
        if (ABL == 1 || ABL == 3 || ABL == 4) return;
        OFX_LDS char* base = lds + stage * STAGE;
        const char* ak = a_base + (size_t)(ABL == 5 ? 0 : kt) * BK * 2;
        const char* wk = w_base + (size_t)(ABL == 5 ? 0 : kt) * BK * 2;
#pragma unroll
        for (int i = 0; i < A_PER_WAVE; ++i) glds16(ak + a_off[i], base + a_dst + i * 1024);
#pragma unroll
        for (int i = 0; i < W_PER_WAVE; ++i) glds16(wk + w_off[i], base + w_dst + i * 1024);
    };

    const int fr = lane & 15, fq = lane >> 4, fsw = fr >> 1;
    const int a_frag = (wr * 128 + fr) * 128;
    const int w_frag = TM * BK * 2 + (wc * 64 + fr) * 128;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    v8 af[2][8], wf[2][4];
    const int nk = p.K / BK;
    unsigned long long c0 = 0, r0 = 0;
    if (p.dbg) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    issue(0, 0);
    if (NST == 2) issue(nk > 1 ? 1 : 0, 1);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = NST == 2 ? (kt & 1) : 0;
        // k-tile kt landed (this wave's pieces); with two stages k-tile kt+1 may stay in flight
        if (NST == 2) {
            if (NLD == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (ABL != 4) __builtin_amdgcn_s_barrier();         // ... and everybody else's
        OFX_LDS char* base = lds + cur * STAGE;
        if (ABL < 2 || ABL == 5 || kt == 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int chk = ((ks * 4 + fq) ^ fsw) * 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[ks][j] = *(OFX_LDS v8*)(base + w_frag + j * 16 * 128 + chk);
#pragma unroll
                for (int i = 0; i < 8; ++i) af[ks][i] = *(OFX_LDS v8*)(base + a_frag + i * 16 * 128 + chk);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (ABL != 4) __builtin_amdgcn_s_barrier();         // every wave holds its fragments: the stage is free
        // 64 MFMAs; the next k-tile's LDS-DMA goes out one piece per 5 MFMAs in program order, so the matrix
        // pipe keeps running while the wave issues them.  No branch in the stream: past the end the prefetch
        // is clamped to the last k-tile (a redundant fill of a stage nobody reads again).
        OFX_LDS char* nbase = lds + cur * STAGE;
        const int kn = ABL == 5 ? 0 : (kt + NST < nk ? kt + NST : nk - 1);
        const char* ak = a_base + (size_t)kn * BK * 2;
        const char* wk = w_base + (size_t)kn * BK * 2;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            if (ABL != 1 && ABL != 3 && ABL != 4 && m % 5 == 0 && m / 5 < NLD) {
                const int q = m / 5;
                if (q < A_PER_WAVE) glds16(ak + a_off[q], nbase + a_dst + q * 1024);
                else glds16(wk + w_off[q - A_PER_WAVE], nbase + w_dst + (q - A_PER_WAVE) * 1024);
            }
            const int ks = m >> 5, i = (m >> 2) & 7, j = m & 3;
            acc[i][j] = OpT<T>::mfma16(wf[ks][j], af[ks][i], acc[i][j]);
        }
        __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the clamped tail prefetches before the wave can end
    unsigned long long cloop_end = 0;
    if (p.dbg) {
        cloop_end = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            p.dbg[4 * blockIdx.x] = cloop_end - c0;
            p.dbg[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
            p.dbg[4 * blockIdx.x + 2] = c0 - cstart;
        }
    }

    OFX_LDS char* ep = lds + NST * STAGE + wave * EPI2_BYTES_PER_WAVE;   // private staging, outside the stages
    const int gm0 = m0 + wr * 128, gn0 = n0 + wc * 64;
    // LayerNorm-fold consumer: the row statistics are staged per wave in the (now idle) stage buffers - block-uniform condition
    OFX_LDS float* st = nullptr;
    if (p.row_stat && p.out_kind != 0 && ABL == 0) { __syncthreads(); st = (OFX_LDS float*)(lds + wave * 1024); }
    epilogue2_dispatch<T>(p, ep, acc, gm0, gn0, lane, st);
    if (p.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) p.dbg[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memtime() - cloop_end;
    }
}

template <typename T, int WR, int WC, int NST, int ABL = 0>
static int launch_big(KArgs& k, int M, int N, hipStream_t s) {
    constexpr int TM = WR * 128, TN = WC * 64, NW = WR * WC;
    constexpr int LDSB = NST * (TM + TN) * BK * 2 + NW * EPI2_BYTES_PER_WAVE;
    static DeviceOnce attr;
    TRY(attr.run([]() -> int {
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_big_kernel<T, WR, WC, NST, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        return OFX_OK;
    }));
    k.tiles_n = N / TN; k.tiles_m = (M + TM - 1) / TM; k.nwg = k.tiles_m * k.tiles_n;
    OFX_PLAUNCH(true, (gemm_big_kernel<T, WR, WC, NST, ABL>), dim3(k.nwg), dim3(64 * NW), LDSB, s, k);
    return OFX_OK;
}

}  // namespace

int ofx_gemm_launch_big(void* kargs, int kind, int ablate, int op_dtype, int M, int N, hipStream_t s) {
    KArgs& k = *(KArgs*)kargs;
    (void)ablate;
#ifdef OFX_DIAG      // ablation variants (wrong results, timing diagnostics): `make DIAG=1`
    if (kind == 2 && ablate == 1) return launch_big<bf16_t, 2, 4, 2, 1>(k, M, N, s);
    if (kind == 2 && ablate == 2) return launch_big<bf16_t, 2, 4, 2, 2>(k, M, N, s);
    if (kind == 2 && ablate == 3) return launch_big<bf16_t, 2, 4, 2, 3>(k, M, N, s);
    if (kind == 2 && ablate == 4) return launch_big<bf16_t, 2, 4, 2, 4>(k, M, N, s);
    if (kind == 2 && ablate == 5) return launch_big<bf16_t, 2, 4, 2, 5>(k, M, N, s);
#endif
    if (kind == 2) return op_dtype == OFX_F16 ? launch_big<f16_t, 2, 4, 2>(k, M, N, s) : launch_big<bf16_t, 2, 4, 2>(k, M, N, s);
    return op_dtype == OFX_F16 ? launch_big<f16_t, 2, 2, 1>(k, M, N, s) : launch_big<bf16_t, 2, 2, 1>(k, M, N, s);
}
