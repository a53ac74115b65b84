// Ping-pong variant of the 256x256 tile kernel (short-K GEMMs); dispatcher in gemm.hip.
#include "gemm_common.h"

namespace {
// ------------------------------------------------------------------------------------------------
// Ping-pong variant of the 256x256 tile (8 waves): waves 0-3 (group 0, rows 0-127) and waves 4-7 (group 1) run
// the same per-k-tile program offset by ONE barrier slot, so on every SIMD one wave reads its 24 fragments and
// issues LDS-DMA while the other wave runs its 64 MFMAs.  Slot schedule (B = block barrier, j = k-tile):
//   group 0:  [W0] B [R0] B [M0 W1] B [R1 I2] B [M1 W2] B [R2 I3] B ...
//   group 1:  [W0] B [  ] B [R0 W1] B [M0 I2] B [R1 W2] B [M1 I3] B ...
// k-tile j >= 2 is issued by both groups in slot 2j-1 (after both groups read k-tile j-2, slots 2j-3 / 2j-2: WAR),
// waited for (vmcnt(0)) at the end of slot 2j and read in slots 2j+1 / 2j+2 (RAW: every wave's wait precedes the
// barrier that opens slot 2j+1).  Two 64 KiB stages + 32 KiB private epilogue staging.
template <typename T, int ABL = 0>      // ABL (make DIAG=1, wrong results): 1 no W pieces in the loop's LDS-DMA, 2 no A pieces, 3 neither
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(KArgs p) {
    typedef typename OpT<T>::v8 v8;
    constexpr int TM = 256, TN = 256, STAGE = (TM + TN) * BK * 2;
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

    const int lrow = lane >> 3, lchk = lane & 7;
    const char* a_base = p.A + (size_t)m0 * p.lda * 2;
    const char* w_base = p.W + (size_t)n0 * p.K * 2;
    unsigned a_off[4], w_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + lrow;
        const int rr = m0 + row < p.M ? row : p.M - 1 - m0;
        a_off[i] = ((unsigned)rr * p.lda + (lchk ^ ((row >> 1) & 7)) * 8) * 2;
        w_off[i] = ((unsigned)row * p.K + (lchk ^ ((row >> 1) & 7)) * 8) * 2;
    }
    const int a_dst = wave * 4 * 1024, w_dst = TM * BK * 2 + wave * 4 * 1024;
    auto issue_all = [&](int kt, int stage) {
        OFX_LDS char* base = lds + stage * STAGE;
        const char* ak = a_base + (size_t)kt * BK * 2;
        const char* wk = w_base + (size_t)kt * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (!(ABL & 2) || kt < 2) glds16(ak + a_off[i], base + a_dst + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) if (!(ABL & 1) || kt < 2) glds16(wk + w_off[i], base + w_dst + i * 1024);
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

#define OFX_READ_FRAGS(STG)                                                                                   \
    {                                                                                                         \
        OFX_LDS char* base_ = lds + (STG) * STAGE;                                                            \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                    \
            const int chk = ((ks * 4 + fq) ^ fsw) * 16;                                                       \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) wf[ks][j] = *(OFX_LDS v8*)(base_ + w_frag + j * 16 * 128 + chk); \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) af[ks][i] = *(OFX_LDS v8*)(base_ + a_frag + i * 16 * 128 + chk); \
        }                                                                                                     \
    }

    const int nk = p.K / BK;
    issue_all(0, 0);
    issue_all(nk > 1 ? 1 : 0, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // k-tile 0 landed (my pieces)
    __builtin_amdgcn_s_barrier();                           // ---- end of slot 0
    if (wr == 0) {
        for (int t = 0; t < nk; ++t) {
            // slot 2t+1: stage (t+1)&1 held k-tile t-1, read by group 1 in slot 2t -> refill it, then read k-tile t
            if (t >= 1 && t + 1 < nk) issue_all(t + 1, (t + 1) & 1);
            OFX_READ_FRAGS(t & 1)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // slot 2t+2
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int m = 0; m < 64; ++m) {
                const int ks = m >> 5, i = (m >> 2) & 7, j = m & 3;
                acc[i][j] = OpT<T>::mfma16(wf[ks][j], af[ks][i], acc[i][j]);
            }
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // my pieces of k-tile t+1 landed
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();                       // group 1's last MFMA slot
    } else {
        __builtin_amdgcn_s_barrier();                       // slot 1: group 0 reads k-tile 0
        for (int t = 0; t < nk; ++t) {
            // slot 2t+2
            OFX_READ_FRAGS(t & 1)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // my pieces of k-tile t+1 landed (issued in slot 2t+1)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // slot 2t+3: both groups have read stage t&1 -> refill it with k-tile t+2 under the MFMAs
            // (past the end the piece addresses are clamped to the last k-tile: a redundant fill nobody reads)
            OFX_LDS char* nbase = lds + (t & 1) * STAGE;
            const int kn = t + 2 < nk ? t + 2 : nk - 1;
            const char* ak = a_base + (size_t)kn * BK * 2;
            const char* wk = w_base + (size_t)kn * BK * 2;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int m = 0; m < 64; ++m) {
                if (m % 5 == 0 && m / 5 < 8) {
                    const int q = m / 5;
                    if (q < 4) { if (!(ABL & 2)) glds16(ak + a_off[q], nbase + a_dst + q * 1024); }
                    else if (!(ABL & 1)) glds16(wk + w_off[q - 4], nbase + w_dst + (q - 4) * 1024);
                }
                const int ks = m >> 5, i = (m >> 2) & 7, j = m & 3;
                acc[i][j] = OpT<T>::mfma16(wf[ks][j], af[ks][i], acc[i][j]);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        }
    }
#undef OFX_READ_FRAGS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    OFX_LDS char* ep = lds + 2 * STAGE + wave * EPI2_BYTES_PER_WAVE;
    const int gm0 = m0 + wr * 128, gn0 = n0 + wc * 64;
    // LayerNorm-fold consumer: the row statistics are staged per wave in the (now idle) stage buffers - block-uniform condition
    OFX_LDS float* st = nullptr;
    if (p.row_stat && p.out_kind != 0) { __syncthreads(); st = (OFX_LDS float*)(lds + wave * 1024); }
    epilogue2_dispatch<T>(p, ep, acc, gm0, gn0, lane, st);
}


template <typename T, int ABL = 0>
static int launch_pp(KArgs& k, int M, int N, hipStream_t s) {
    constexpr int LDSB = 2 * (256 + 256) * BK * 2 + 8 * EPI2_BYTES_PER_WAVE;
    static DeviceOnce attr;
    TRY(attr.run([]() -> int {
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_pp_kernel<T, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        return OFX_OK;
    }));
    k.tiles_n = N / 256; k.tiles_m = (M + 255) / 256; k.nwg = k.tiles_m * k.tiles_n;
    OFX_PLAUNCH(true, (gemm_pp_kernel<T, ABL>), dim3(k.nwg), dim3(512), LDSB, s, k);
    return OFX_OK;
}

}  // namespace

extern int g_gemm_ablate;
int ofx_gemm_launch_pp(void* kargs, int op_dtype, int M, int N, hipStream_t s) {
    KArgs& k = *(KArgs*)kargs;
#ifdef OFX_DIAG
    if (g_gemm_ablate == 1) return launch_pp<f16_t, 1>(k, M, N, s);
    if (g_gemm_ablate == 2) return launch_pp<f16_t, 2>(k, M, N, s);
    if (g_gemm_ablate == 3) return launch_pp<f16_t, 3>(k, M, N, s);
#endif
    return op_dtype == OFX_F16 ? launch_pp<f16_t>(k, M, N, s) : launch_pp<bf16_t>(k, M, N, s);
}
