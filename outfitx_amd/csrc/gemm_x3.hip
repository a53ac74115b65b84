// Three-product GEMM with every operand tile loaded ONCE: C = A_hi . W_hi^T + A_lo . W_hi^T + A_hi . W_lo^T.
//
// The K-concatenated form of the same sum ([hi | lo | hi] x [hi | hi | lo], K' = 3 K on a single-product kernel) stages A_hi and
// W_hi twice: 6 operand tiles per three products.  This kernel reads the SAME buffers - activation rows [hi | lo | hi] (lda = 3 K,
// the LayerNorm / epilogue split3 output), weight rows [hi | hi | lo] (ofx_launch_pack_rows mode 2) - but stages the four distinct
// tiles of a k-step once and runs the three products from registers: 48 KiB through LDS per 32-deep k-step of a 256 x 128 tile for
// 3 x 2.1 MFLOP = 131 FLOP per staged byte, against 64 (128 x 128 tiles) / 85 (256 x 128) for the K-concatenated GEMMs - and these loops
// run at the rate their LDS fill sustains.
//
// 256 x 128 tile, 8 waves as 4 x 2 of 64 x 64 wave tiles (64 accumulator VGPRs), BK = 32, three 48 KiB stages
// [A_hi 256 rows | A_lo 256 rows | W_hi 128 rows | W_lo 128 rows] x 64 B; per k-step and wave 6 LDS-DMA pieces (2 + 2 + 1 + 1), 16
// fragment reads and 48 MFMAs.  Schedule, swizzle and persistence are gemm_w2.hip's: two wave groups one barrier slot apart, step t + 2
// issued in iteration t, counted vmcnt(6), blocks walk tiles b, b + grid, ... with the next tile's first two steps fetched under
// the epilogue; LDS-DMA through buffer resources as in gemm_w2f8.hip (rows past M read zeros).
#include "gemm_common.h"

extern int g_w2_persist, g_x3_persist;
#ifndef OFX_MFMA_WKEEP
#define OFX_MFMA_WKEEP 0
#endif
namespace {

__device__ __forceinline__ void x3_bload16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, OFX_LDS char* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (OFX_LDS void*)l, 16, (int)voff, (int)soff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t x3_rsrc(const char* base, size_t bytes = 0x7fffffff) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(bytes < 0x7fffffff ? bytes : 0x7fffffff), 0x00020000);
}

template <typename T>
__global__ __launch_bounds__(512, 2) void gemm_x3_kernel(KArgs p) {
    typedef typename OpT<T>::v8 v8;
    constexpr int TM = 256, TN = 128, BK2 = 32, PA = TM * BK2 * 2, PW = TN * BK2 * 2, STAGE = 2 * PA + 2 * PW, NST = 3;      // 16 + 16 + 8 + 8 KiB
    extern __shared__ __attribute__((aligned(16))) char smem[];
    OFX_LDS char* lds = (OFX_LDS char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;            // 4 x 2 waves of 64 x 64; waves 0-3 (rows 0-127) are ping-pong group 0
    int grp = wave >> 2;
    asm volatile("" : "+s"(grp));
    const int Kl = p.K / 3;                             // logical depth; A rows are [hi | lo | hi] (3 Kl), W rows [hi | hi | lo]
    p.K = Kl;
    if (p.m_dev) {                                      // device-side live row count: the launcher runs one block per tile then
        const int m_live = *p.m_dev;
        p.M = m_live < p.M ? m_live : p.M;
    }
    auto map_tile = [&](int vb, int& m0, int& n0) {
        int bid = vb;
        {
            const int nx = 8, q = p.nwg / nx, r = p.nwg % nx, x = bid % nx, i = bid / nx;
            bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
        }
        const int per_group = p.group_m * p.tiles_n;
        const int gidx = bid / per_group, first = gidx * p.group_m;
        const int gm = min(p.group_m, p.tiles_m - first);
        const int r = bid - gidx * per_group;
        m0 = (first + r % gm) * TM;
        n0 = (r / gm) * TN;
    };
    const int nk = Kl / BK2;
    const unsigned a_lo_off = (unsigned)Kl * 2, w_lo_off = (unsigned)Kl * 4;      // byte offsets of the lo column blocks in a row

    int vb = blockIdx.x, m0, n0;
    map_tile(vb, m0, n0);
    if (m0 >= p.M) return;
    int base = 0;                                       // (global index of the current tile's step 0) mod 3
    bool first = true;
    for (;;) {
        const bool has_next = vb + (int)gridDim.x < p.nwg;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        // pieces: 16 rows x 64 B; lane l -> row l >> 2, physical slot l & 3 <- logical chunk (l & 3) ^ f(row >> 2), f(g) = (-g) & 3
        const int prow = ln >> 2, pchk = (ln & 3) ^ ((4 - (ln >> 4)) & 3);
        unsigned a_off[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a_off[i] = ((unsigned)((wave * 2 + i) * 16 + prow) * p.lda + pchk * 8) * 2;
        const unsigned w_off = ((unsigned)(wave * 16 + prow) * (3 * Kl) + pchk * 8) * 2;
        const int fr = ln & 15, fq = ln >> 4;
        const int fchk = (fq ^ ((4 - (fr >> 2)) & 3)) * 16;
        const int a_frag = (wr * 64 + fr) * 64 + fchk;
        const int w_frag = 2 * PA + (wc * 64 + fr) * 64 + fchk;
        int m1 = m0, n1 = n0;                           // the next tile (the block's last tile re-fills its own first steps: nobody reads them)
        if (has_next) map_tile(vb + (int)gridDim.x, m1, n1);
        const size_t a_row = (size_t)p.lda * 2, w_row = (size_t)Kl * 6;
        const __amdgpu_buffer_rsrc_t r_a = x3_rsrc(p.A + (size_t)m0 * a_row, (size_t)(p.M - m0) * a_row), r_w = x3_rsrc(p.W + (size_t)n0 * w_row);
        const __amdgpu_buffer_rsrc_t r_a1 = x3_rsrc(p.A + (size_t)m1 * a_row, (size_t)(p.M - m1) * a_row), r_w1 = x3_rsrc(p.W + (size_t)n1 * w_row);
        // k-step x of the tile walk (x >= nk: step x - nk of the next tile): A_hi, A_lo (2 pieces each), W_hi, W_lo (1 each)
        auto issue_step = [&](int x) {
            OFX_LDS char* stg = lds + ((base + x) % NST) * STAGE;
            const bool nx = x >= nk;
            const __amdgpu_buffer_rsrc_t ra = nx ? r_a1 : r_a, rw = nx ? r_w1 : r_w;
            const unsigned koff = (unsigned)(nx ? x - nk : x) * BK2 * 2;
            x3_bload16(ra, a_off[0], koff, stg + wave * 2048); x3_bload16(ra, a_off[1], koff, stg + wave * 2048 + 1024);
            x3_bload16(ra, a_off[0], koff + a_lo_off, stg + PA + wave * 2048); x3_bload16(ra, a_off[1], koff + a_lo_off, stg + PA + wave * 2048 + 1024);
            x3_bload16(rw, w_off, koff, stg + 2 * PA + wave * 1024);
            x3_bload16(rw, w_off, koff + w_lo_off, stg + 2 * PA + PW + wave * 1024);
        };

        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        v8 ah[4], al[4], wh[4], wl[4];

#define OFX_X3_READ(STEP)                                                                                     \
    {                                                                                                         \
        OFX_LDS char* base_ = lds + ((base + (STEP)) % NST) * STAGE;                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { wh[j] = *(OFX_LDS v8*)(base_ + w_frag + j * 1024); wl[j] = *(OFX_LDS v8*)(base_ + PW + w_frag + j * 1024); } \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { ah[i] = *(OFX_LDS v8*)(base_ + a_frag + i * 1024); al[i] = *(OFX_LDS v8*)(base_ + PA + a_frag + i * 1024); } \
    }
    // 48 MFMAs: per (activation fragment, weight fragment) hi.hi, lo.hi, hi.lo
#define OFX_X3_MFMA()                                                                                         \
    {                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        if (OFX_MFMA_WKEEP) {       /* the instruction's first operand kept over 8 / 4 consecutive MFMAs (tools/mfma_power_probe.hip) */ \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                   \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[i][j] = OpT<T>::mfma16(wh[j], ah[i], acc[i][j]); \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[i][j] = OpT<T>::mfma16(wh[j], al[i], acc[i][j]); \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[i][j] = OpT<T>::mfma16(wl[j], ah[i], acc[i][j]); \
            }                                                                                                 \
        } else {                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                   \
                acc[i][j] = OpT<T>::mfma16(wh[j], ah[i], acc[i][j]);                                          \
                acc[i][j] = OpT<T>::mfma16(wh[j], al[i], acc[i][j]);                                          \
                acc[i][j] = OpT<T>::mfma16(wl[j], ah[i], acc[i][j]);                                          \
            }                                                                                                 \
        }                                                                                                     \
        __builtin_amdgcn_s_setprio(0);                                                                        \
    }
        if (first) {
            issue_step(0); issue_step(1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // step 0 landed (my pieces)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // steps 0 and 1 (fetched under the previous epilogue) and that epilogue's stores
        }
        __builtin_amdgcn_s_barrier();                               // ---- end of slot 0
        // group 0: slots 2t+1 (issue step t+2, read step t) and 2t+2 (multiply); group 1 one slot later (gemm_w2.hip)
        if (grp == 0) {
            int t = 0;
            do {
                issue_step(t + 2);
                OFX_X3_READ(t)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                OFX_X3_MFMA()
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // my pieces of step t+1 landed (step t+2 stays in flight)
                __builtin_amdgcn_s_barrier();
            } while (++t < nk);
            __builtin_amdgcn_s_barrier();                           // closes group 1's last MFMA slot: every read of this tile's stages is done
        } else {
            __builtin_amdgcn_s_barrier();                           // slot 1: group 0 reads step 0
            int t = 0;
            do {
                issue_step(t + 2);
                OFX_X3_READ(t)
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // my pieces of step t+1 landed: group 0 reads them in slot 2t+3
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                OFX_X3_MFMA()
                __builtin_amdgcn_s_barrier();
            } while (++t < nk);
        }
#undef OFX_X3_READ
#undef OFX_X3_MFMA
        // Epilogue staging: the stage of this tile's LAST step (8 x 4 KiB + the LayerNorm-fold statistics slots behind them); the fills of
        // steps nk, nk + 1 (the next tile's first steps, still landing) target the other two stages.
        OFX_LDS char* estage = lds + ((base + nk - 1) % NST) * STAGE;
        OFX_LDS char* ep = estage + wave * EPI2_BYTES_PER_WAVE;
        OFX_LDS float* st = nullptr;
        if (p.row_stat && p.out_kind != 0) st = (OFX_LDS float*)(estage + 8 * EPI2_BYTES_PER_WAVE + wave * 1024);
        epilogue2_dispatch<T, 4, 4, 0>(p, ep, acc, m0 + wr * 64, n0 + wc * 64, ln, st);
        if (!has_next) break;
        vb += gridDim.x; map_tile(vb, m0, n0); base = (base + nk) % NST; first = false;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the last tile's redundant fills have landed before the wave ends
}

template <typename T>
static int launch_x3(KArgs& k, int M, int N, hipStream_t s) {
    constexpr int LDSB = 3 * (2 * 256 + 2 * 128) * 32 * 2;          // 144 KiB
    static DeviceOnce attr;
    TRY(attr.run([]() -> int {
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_x3_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        return OFX_OK;
    }));
    k.tiles_n = N / 128; k.tiles_m = (M + 255) / 256; k.nwg = k.tiles_m * k.tiles_n;
    int persist = g_x3_persist == 1 ? g_w2_persist : 0;            // ofx_tune(16, 0): one block per tile (short-lived blocks: a side stream's GEMM then frees its CUs tile by tile)
    if (persist < 0) {
        static int cus[64] = {0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        int& c = cus[dev & 63];
        if (c == 0) { int v = 0; c = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256; }
        persist = c;
    }
    const int grid = (persist && !k.m_dev && k.nwg > persist) ? persist : k.nwg;
    OFX_PLAUNCH(true, (gemm_x3_kernel<T>), dim3(grid), dim3(512), LDSB, s, k);
    return OFX_OK;
}

}  // namespace

int ofx_gemm_launch_x3(void* kargs, int op_dtype, int M, int N, hipStream_t s) {
    KArgs& k = *(KArgs*)kargs;
    return op_dtype == OFX_F16 ? launch_x3<f16_t>(k, M, N, s) : launch_x3<bf16_t>(k, M, N, s);
}
