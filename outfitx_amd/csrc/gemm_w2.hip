// Dual-weight ("W2") 256x256 tile kernel: C = A . (W_hi + W_lo)^T with ONE copy of A.
//
// The weight matrix is stored split, row n = [hi(K) | lo(K)] (hi = round(w), lo = round(w - hi), both in the operand type), so
// the product keeps ~22 significant weight bits at two MFMAs per (A fragment, column tile) while the activation tile is loaded
// and its fragments are read ONCE: per 32-deep k-step and wave 8 A + 4 W_hi + 4 W_lo fragment reads feed 64 MFMAs (the
// single-product kernels read 12 fragments per 32 MFMAs), so the loop is bound by the matrix pipe, not by the LDS port.
// Used for the CLIP GEMMs whose weight rounding dominates the end-to-end error (fc2, out-proj, patch embedding: DESIGN.md §2).
//
// Structure (the ping-pong kernel's, gemm_pp.hip, at BK = 32 with three 48 KiB stages):
//   waves 0-3 (group 0, rows 0-127) and 4-7 (group 1) run the same per-step program offset by ONE barrier slot: on every SIMD
//   one wave reads its 16 fragments while the other runs its 64 MFMAs and issues its 6 LDS-DMA pieces of a later step.
//     slot:      0     1         2         3         4         5
//     group 0:  [W0]  [R0 I2]   [M0]      [R1 I3]   [M1]      [R2 I4]  ...   I(s) = issue step s into stage s % 3
//     group 1:  [W0]  [  ]      [R0 I2]   [M0]      [R1 I3]   [M1]     ...
//   Step s is read in slots 2s+1 (group 0) and 2s+2 (group 1); its stage is refilled with step s+3 in slots 2s+3 / 2s+4 (WAR: both
//   groups' reads ended before the barrier that closes slot 2s+2).  Every wave waits for its own pieces of step s (counted
//   vmcnt(6): the youngest step stays in flight) before the barrier that closes slot 2s; step s is first read in slot 2s+1 (RAW).
// LDS image of a stage: [A 256 rows | W_hi 256 rows | W_lo 256 rows] x 64 B; a wave-instruction moves 16 rows x 64 B; the 16-B
// chunk c of row r sits at slot c ^ f(r >> 2), f(g) = (-g) & 3, applied on the DMA source address and on the ds_read_b128 side:
// conflict-free for the hardware's lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS table).
#include "gemm_common.h"

namespace {

template <typename T, int ABL = 0>      // ABL (make DIAG=1; 1-3 give wrong results): 1 no LDS-DMA in the loop, 2 no fragment reads after step 0, 3 both, 5 DMA issued AFTER the fragment reads (the first shipped order)
__global__ __launch_bounds__(512, 2) void gemm_w2_kernel(KArgs p) {
    typedef typename OpT<T>::v8 v8;
    constexpr int TM = 256, TN = 256, BK2 = 32, PART = TM * BK2 * 2, STAGE = 3 * PART, NST = 3;      // 16 KiB per operand, 48 KiB per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    OFX_LDS char* lds = (OFX_LDS char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int Kh = p.K >> 1;                           // logical K; p.K = 2 Kh is the row stride of W = [hi | lo]

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

    // LDS-DMA pieces: 16 rows x 64 B; lane l -> row l >> 2, physical slot l & 3 <- logical chunk (l & 3) ^ f(row >> 2)
    const int prow = lane >> 2, pchk = (lane & 3) ^ ((4 - (lane >> 4)) & 3);
    const char* a_base = p.A + (size_t)m0 * p.lda * 2;
    const char* w_base = p.W + (size_t)n0 * p.K * 2;
    unsigned a_off[2], w_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + prow;
        const int rr = m0 + row < p.M ? row : p.M - 1 - m0;          // rows past M re-read the last live row
        a_off[i] = ((unsigned)rr * p.lda + pchk * 8) * 2;
        w_off[i] = ((unsigned)row * p.K + pchk * 8) * 2;
    }
    const int dst0 = wave * 2 * 1024;
    const unsigned lo_bytes = (unsigned)Kh * 2;
    const int nk = Kh / BK2;
#define OFX_W2_PIECE(Q, AK, WK, BASE)                                                              \
    {                                                                                              \
        if ((Q) < 2) glds16((AK) + a_off[(Q)], (BASE) + dst0 + (Q) * 1024);                        \
        else if ((Q) < 4) glds16((WK) + w_off[(Q) - 2], (BASE) + PART + dst0 + ((Q) - 2) * 1024);  \
        else glds16((WK) + lo_bytes + w_off[(Q) - 4], (BASE) + 2 * PART + dst0 + ((Q) - 4) * 1024); \
    }
    auto issue_all = [&](int step) {
        const int sc = step < nk ? step : nk - 1;
        OFX_LDS char* base = lds + (step % NST) * STAGE;
        const char* ak = a_base + (size_t)sc * BK2 * 2;
        const char* wk = w_base + (size_t)sc * BK2 * 2;
#pragma unroll
        for (int q = 0; q < 6; ++q) OFX_W2_PIECE(q, ak, wk, base)
    };

    const int fr = lane & 15, fq = lane >> 4;
    const int fchk = (fq ^ ((4 - (fr >> 2)) & 3)) * 16;
    const int a_frag = (wr * 128 + fr) * 64 + fchk;
    const int w_frag = PART + (wc * 64 + fr) * 64 + fchk;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    v8 af[8], wh[4], wl[4];

#define OFX_W2_READ(STG)                                                                                     \
    {                                                                                                        \
        OFX_LDS char* base_ = lds + (STG) * STAGE;                                                           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) wh[j] = *(OFX_LDS v8*)(base_ + w_frag + j * 16 * 64);   \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) af[i] = *(OFX_LDS v8*)(base_ + a_frag + i * 16 * 64);   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) wl[j] = *(OFX_LDS v8*)(base_ + PART + w_frag + j * 16 * 64); \
    }
    // 64 MFMAs (per A fragment: 4 hi then 4 lo): nothing else in the stream - the LDS-DMA pieces are issued from the READ slots,
    // where the wave would otherwise wait for its fragments (an LDS-DMA issue costs the issuing wave ~100 cycles; threaded through
    // the MFMAs it cost 0.3 us of every 0.8 us slot)
#define OFX_W2_MFMA()                                                                                        \
    {                                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int m = 0; m < 64; ++m) {                                                     \
            const int i = (m >> 3) & 7, j = m & 3;                                                           \
            acc[i][j] = OpT<T>::mfma16((m & 4) ? wl[j] : wh[j], af[i], acc[i][j]);                           \
        }                                                                                                    \
        __builtin_amdgcn_s_setprio(0);                                                                       \
    }

    issue_all(0); issue_all(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");            // step 0 landed (my pieces)
    __builtin_amdgcn_s_barrier();                               // ---- end of slot 0
    if (wr == 0) {
        for (int t = 0; t < nk; ++t) {
            // slot 2t+1: read step t; refill the stage of step t-1 (group 1 read it in slot 2t) with step t+2
            if (ABL == 0 || ABL == 2) issue_all(t + 2);         // before the reads: 760 vs 775 us on the fc2 shape (tools/gemm_w2_bench.py, DIAG build)
            if (ABL < 2 || ABL == 5 || t == 0) OFX_W2_READ(t % NST)
            if (ABL == 5) issue_all(t + 2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // slot 2t+2: multiply step t
            OFX_W2_MFMA()
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // my pieces of step t+1 landed (step t+2 stays in flight)
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();                           // group 1's last MFMA slot
    } else {
        __builtin_amdgcn_s_barrier();                           // slot 1: group 0 reads step 0
        for (int t = 0; t < nk; ++t) {
            // slot 2t+2: read step t; refill the stage of step t-1 (read by group 0 in slot 2t-1, by this group in slot 2t)
            if (ABL == 0 || ABL == 2) issue_all(t + 2);         // before the reads: 760 vs 775 us on the fc2 shape (tools/gemm_w2_bench.py, DIAG build)
            if (ABL < 2 || ABL == 5 || t == 0) OFX_W2_READ(t % NST)
            if (ABL == 5) issue_all(t + 2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // my pieces of step t+1 landed: group 0 reads them in slot 2t+3
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // slot 2t+3: multiply step t
            OFX_W2_MFMA()
            __builtin_amdgcn_s_barrier();
        }
    }
#undef OFX_W2_READ
#undef OFX_W2_MFMA
#undef OFX_W2_PIECE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the clamped tail fills have landed ...
    __builtin_amdgcn_s_barrier();                               // ... everybody's: the stages are dead, the epilogue staging aliases them

    OFX_LDS char* ep = lds + wave * EPI2_BYTES_PER_WAVE;
    const int gm0 = m0 + wr * 128, gn0 = n0 + wc * 64;
    OFX_LDS float* st = nullptr;
    if (p.row_stat && p.out_kind != 0) st = (OFX_LDS float*)(lds + 8 * EPI2_BYTES_PER_WAVE + wave * 1024);
    p.K = Kh;                                                   // the epilogues never read K; keep the logical value anyway
    epilogue2_dispatch<T>(p, ep, acc, gm0, gn0, lane, st);
}

template <typename T, int ABL = 0>
static int launch_w2(KArgs& k, int M, int N, hipStream_t s) {
    constexpr int LDSB = 3 * 3 * 256 * 32 * 2;          // 144 KiB
    static DeviceOnce attr;
    TRY(attr.run([]() -> int {
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_w2_kernel<T, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        return OFX_OK;
    }));
    k.tiles_n = N / 256; k.tiles_m = (M + 255) / 256; k.nwg = k.tiles_m * k.tiles_n;
    OFX_PLAUNCH(true, (gemm_w2_kernel<T, ABL>), dim3(k.nwg), dim3(512), LDSB, s, k);
    return OFX_OK;
}

}  // namespace

extern int g_gemm_ablate;
int ofx_gemm_launch_w2(void* kargs, int op_dtype, int M, int N, hipStream_t s) {
    KArgs& k = *(KArgs*)kargs;
#ifdef OFX_DIAG
    if (g_gemm_ablate == 1) return launch_w2<f16_t, 1>(k, M, N, s);
    if (g_gemm_ablate == 2) return launch_w2<f16_t, 2>(k, M, N, s);
    if (g_gemm_ablate == 3) return launch_w2<f16_t, 3>(k, M, N, s);
    if (g_gemm_ablate == 5) return launch_w2<f16_t, 5>(k, M, N, s);
#endif
    return op_dtype == OFX_F16 ? launch_w2<f16_t>(k, M, N, s) : launch_w2<bf16_t>(k, M, N, s);
}
