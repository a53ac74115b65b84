// Dual-weight ("W2") 256x256 tile kernel: C = A . (W_hi + W_lo)^T with ONE copy of A.
//
// The weight matrix is stored split, row n = [hi(K) | lo(K)] (hi = round(w), lo = round(w - hi), both in the operand type), so
// the product keeps ~22 significant weight bits (f16: about 18 at |w| ~ 0.02, where the lo half is subnormal with an absolute step of
// 2^-24; bf16: 16) at two MFMAs per (A fragment, column tile) while the activation tile is loaded
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
// Persistent over tiles (one block per CU; ofx_tune(11, 0) = one block per tile): steps are numbered across the block's tiles, the
// fills of steps nk and nk + 1 fetch the next tile's first two steps under the current epilogue (see the kernel body).
// LDS image of a stage: [A 256 rows | W_hi 256 rows | W_lo 256 rows] x 64 B; a wave-instruction moves 16 rows x 64 B; the 16-B
// chunk c of row r sits at slot c ^ f(r >> 2), f(g) = (-g) & 3, applied on the DMA source address and on the ds_read_b128 side:
// conflict-free for the hardware's lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS table).
#include "gemm_common.h"

extern int g_w2_persist, g_w2_trim;
namespace {

template <typename T, int ABL = 0>      // ABL (make DIAG=1; wrong results): 1 no LDS-DMA in the loop, 2 no fragment reads after step 0, 3 both
__global__ __launch_bounds__(512, 2) void gemm_w2_kernel(KArgs p) {
    typedef typename OpT<T>::v8 v8;
    constexpr int TM = 256, TN = 256, BK2 = 32, PART = TM * BK2 * 2, STAGE = 3 * PART, NST = 3;      // 16 KiB per operand, 48 KiB per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    OFX_LDS char* lds = (OFX_LDS char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int Kh = p.K >> 1;                           // logical K; 2 Kh is the row stride of W = [hi | lo]
    p.K = Kh;                                          // the epilogues never read K; keep the logical value anyway
    if (p.m_dev) {                                     // device-side live row count: the launcher runs one block per tile then
        const int m_live = *p.m_dev;
        p.M = m_live < p.M ? m_live : p.M;
    }

    // Persistent over tiles: block b runs tiles b, b + gridDim.x, ... (the launcher sizes the grid to one block per CU, or one block
    // per tile on small problems).  The k-steps are numbered across the block's tiles - step g lives in stage g % 3 - so the two fills
    // that used to be redundant at the end of a tile (steps nk, nk + 1) fetch the NEXT tile's first two steps instead: they land
    // under the epilogue, whose staging sits in the stage of the tile's last step (the one stage no fill targets), and the next main
    // loop starts without a load-latency bubble.
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
    const unsigned lo_bytes = (unsigned)Kh * 2;
    const int nk = Kh / BK2;
    const int dst0 = wave * 2 * 1024;

    int vb = blockIdx.x, m0, n0;
    map_tile(vb, m0, n0);
    if (m0 >= p.M) return;                              // only with m_dev (one block per tile)
    int base = 0;                                       // (global index of the current tile's step 0) mod 3
    bool first = true;
    for (;;) {
        const bool has_next = vb + (int)gridDim.x < p.nwg;     // its first two steps are fetched by this tile's last two fills
        // Lane constants are re-derived per tile from an opaque copy of the lane id: kept live across the epilogue they cost more
        // registers than the kernel has (spills), recomputing them costs a few dozen VALU operations per ~50 us tile.
        int ln = lane;
        asm volatile("" : "+v"(ln));
        // LDS-DMA pieces: 16 rows x 64 B; lane l -> row l >> 2, physical slot l & 3 <- logical chunk (l & 3) ^ f(row >> 2)
        const int prow = ln >> 2, pchk = (ln & 3) ^ ((4 - (ln >> 4)) & 3);
        unsigned w_off[2];                                  // tile-independent
#pragma unroll
        for (int i = 0; i < 2; ++i) w_off[i] = ((unsigned)((wave * 2 + i) * 16 + prow) * (2 * Kh) + pchk * 8) * 2;
        auto a_offset = [&](int mt, int i) {                // rows past M re-read the last live row
            const int row = (wave * 2 + i) * 16 + prow;
            const int rr = mt + row < p.M ? row : p.M - 1 - mt;
            return ((unsigned)rr * p.lda + pchk * 8) * 2;
        };
        const int fr = ln & 15, fq = ln >> 4;
        const int fchk = (fq ^ ((4 - (fr >> 2)) & 3)) * 16;
        const int a_frag = (wr * 128 + fr) * 64 + fchk;
        const int w_frag = PART + (wc * 64 + fr) * 64 + fchk;
        const char* a_base = p.A + (size_t)m0 * p.lda * 2;
        const char* w_base = p.W + (size_t)n0 * (2 * Kh) * 2;
        unsigned a_off[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a_off[i] = a_offset(m0, i);

#define OFX_W2_PIECE(Q, AK, A0, A1, WK, BASE)                                                          \
    {                                                                                                  \
        if ((Q) < 2) glds16((AK) + ((Q) ? (A1) : (A0)), (BASE) + dst0 + (Q) * 1024);                    \
        else if ((Q) < 4) glds16((WK) + w_off[(Q) - 2], (BASE) + PART + dst0 + ((Q) - 2) * 1024);       \
        else glds16((WK) + lo_bytes + w_off[(Q) - 4], (BASE) + 2 * PART + dst0 + ((Q) - 4) * 1024);     \
    }
        auto issue_cur = [&](int step) {                // a step of the current tile (step < nk)
            OFX_LDS char* sbase = lds + ((base + step) % NST) * STAGE;
            const char* ak = a_base + (size_t)step * BK2 * 2;
            const char* wk = w_base + (size_t)step * BK2 * 2;
#pragma unroll
            for (int q = 0; q < 6; ++q) OFX_W2_PIECE(q, ak, a_off[0], a_off[1], wk, sbase)
        };
        auto issue_next = [&](int j) {                  // step nk + j: the next tile's step j, in the stage of this tile's step nk + j - 3
            OFX_LDS char* sbase = lds + ((base + nk + j) % NST) * STAGE;
            if (has_next) {
                int m1, n1;
                map_tile(vb + (int)gridDim.x, m1, n1);
                const char* ak = p.A + (size_t)m1 * p.lda * 2 + (size_t)j * BK2 * 2;
                const char* wk = p.W + (size_t)n1 * (2 * Kh) * 2 + (size_t)j * BK2 * 2;
                const unsigned n0_ = a_offset(m1, 0), n1_ = a_offset(m1, 1);
#pragma unroll
                for (int q = 0; q < 6; ++q) OFX_W2_PIECE(q, ak, n0_, n1_, wk, sbase)
            } else {                                    // the block's last tile: a redundant re-fill with the last step (keeps the counted waits uniform; nobody reads it)
                const char* ak = a_base + (size_t)(nk - 1) * BK2 * 2;
                const char* wk = w_base + (size_t)(nk - 1) * BK2 * 2;
#pragma unroll
                for (int q = 0; q < 6; ++q) OFX_W2_PIECE(q, ak, a_off[0], a_off[1], wk, sbase)
            }
        };

        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        v8 af[8], wh[4], wl[4];

#define OFX_W2_READ(STEP)                                                                                    \
    {                                                                                                        \
        OFX_LDS char* base_ = lds + ((base + (STEP)) % NST) * STAGE;                                         \
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

        if (first) {
            issue_cur(0); issue_cur(1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // step 0 landed (my pieces)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // steps 0 and 1 (fetched under the previous epilogue) and that epilogue's stores
        }
        __builtin_amdgcn_s_barrier();                               // ---- end of slot 0
        // One iteration of group 0 (slots 2t+1, 2t+2) / group 1 (slots 2t+2, 2t+3): ISSUE refills the stage of step t-1 (both groups
        // have read it) with step t+2, before the reads (760 vs 775 us on the fc2 shape with the DMA after them); the last two
        // iterations of a tile are peeled so that the steady-state body carries no next-tile logic.
#define OFX_W2_ITER_G0(T_, ISSUE)                                                                                \
        {                                                                                                        \
            if (ABL == 0 || ABL == 2) { ISSUE; }                                                                 \
            if (ABL < 2 || (T_) == 0) OFX_W2_READ(T_)                                                            \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            __builtin_amdgcn_s_barrier();                                                                        \
            OFX_W2_MFMA()                                                                                        \
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    /* my pieces of step t+1 landed (step t+2 stays in flight) */ \
            __builtin_amdgcn_s_barrier();                                                                        \
        }
#define OFX_W2_ITER_G1(T_, ISSUE)                                                                                \
        {                                                                                                        \
            if (ABL == 0 || ABL == 2) { ISSUE; }                                                                 \
            if (ABL < 2 || (T_) == 0) OFX_W2_READ(T_)                                                            \
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    /* my pieces of step t+1 landed: group 0 reads them in slot 2t+3 */ \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            __builtin_amdgcn_s_barrier();                                                                        \
            OFX_W2_MFMA()                                                                                        \
            __builtin_amdgcn_s_barrier();                                                                        \
        }
        if (wr == 0) {
            for (int t = 0; t < nk - 2; ++t) OFX_W2_ITER_G0(t, issue_cur(t + 2))
            OFX_W2_ITER_G0(nk - 2, issue_next(0))
            OFX_W2_ITER_G0(nk - 1, issue_next(1))
            __builtin_amdgcn_s_barrier();                           // group 1's last MFMA slot begins: every read of this tile is done
        } else {
            __builtin_amdgcn_s_barrier();                           // slot 1: group 0 reads step 0
            for (int t = 0; t < nk - 2; ++t) OFX_W2_ITER_G1(t, issue_cur(t + 2))
            OFX_W2_ITER_G1(nk - 2, issue_next(0))
            OFX_W2_ITER_G1(nk - 1, issue_next(1))
        }
#undef OFX_W2_ITER_G0
#undef OFX_W2_ITER_G1
#undef OFX_W2_READ
#undef OFX_W2_MFMA
#undef OFX_W2_PIECE
        // Epilogue staging: the stage of this tile's LAST step - read by everybody before the barriers above, and the one stage
        // the fills of steps nk, nk + 1 (the next tile's first steps, still landing) do not target.
        OFX_LDS char* estage = lds + ((base + nk - 1) % NST) * STAGE;
        OFX_LDS char* ep = estage + wave * EPI2_BYTES_PER_WAVE;
        const int gm0 = m0 + wr * 128, gn0 = n0 + wc * 64;
        OFX_LDS float* st = nullptr;
        if (p.row_stat && p.out_kind != 0) st = (OFX_LDS float*)(estage + 8 * EPI2_BYTES_PER_WAVE + wave * 1024);
        epilogue2_dispatch<T>(p, ep, acc, gm0, gn0, ln, st);
        if (!has_next) break;
        vb += gridDim.x; map_tile(vb, m0, n0); base = (base + nk) % NST; first = false;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the last tile's redundant fills have landed before the wave ends
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
    // one block per CU, each walking tiles b, b + grid, ... (g_w2_persist: -1 = the device's CU count, 0 = one block per tile, n = n blocks)
    int persist = g_w2_persist;
    if (persist < 0) {
        static int cus[64] = {0};                      // per device ordinal; a benign race writes the same value
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        int& c = cus[dev & 63];
        if (c == 0) { int v = 0; c = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256; }
        persist = c;
    }
    int grid = (persist && !k.m_dev && k.nwg > persist) ? persist : k.nwg;
    // ofx_tune(14, 1): the smallest grid that still finishes in the same number of rounds (1,200 or 3,600 tiles take 5 / 15 rounds on 256
    // blocks and on 240 alike; the 16 CUs left alone would serve the side stream's text tower) - measured 0.4 ms per step SLOWER, off
    if (g_w2_trim && grid < k.nwg) { const int rounds = (k.nwg + grid - 1) / grid; grid = (k.nwg + rounds - 1) / rounds; }
    OFX_PLAUNCH(true, (gemm_w2_kernel<T, ABL>), dim3(grid), dim3(512), LDSB, s, k);
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
#endif
    return op_dtype == OFX_F16 ? launch_w2<f16_t>(k, M, N, s) : launch_w2<bf16_t>(k, M, N, s);
}
