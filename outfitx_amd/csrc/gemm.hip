// C[M,N] = epilogue( A[M,K] · W[N,K]^T ) on gfx950 matrix cores.
//
// Both operands are K-contiguous (activations row-major, weights in torch Linear layout), bf16 or
// f16, fp32 accumulate (v_mfma_f32_16x16x32_{bf16,f16}).  This one kernel carries every dense
// contraction of the scoring path (reference call sites: the nn.Linear / in_proj GEMMs inside
// nn.TransformerEncoderLayer built at src/models/outfit_x.py:32-45 and inside HF CLIP's
// CLIPAttention / CLIPMLP called from clip_image_encoder.py:74-76, clip_text_encoder.py:56-58).
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), wave tile 64x64 = 4x4 MFMA tiles x 2 k-steps.
//  * global -> LDS by LDS-DMA (global_load_lds_dwordx4), two stages; the LDS image is lane-linear,
//    so the bank swizzle (16-B chunk ^= (row>>1)&7) is applied to the per-lane SOURCE address and
//    again on the ds_read_b128 side (guide rule 21).
//  * operands are swapped in the MFMA (W fragment as A-operand) so a lane ends up holding 4
//    consecutive output COLUMNS of one row; the epilogue goes through LDS once and leaves as
//    whole 128/256-byte row segments with bias / activation / fp32 residual fused.
//  * 1-D grid with an XCD-aware remap: the blocks that share an A row-panel are consecutive on
//    one XCD so the panel is fetched into that XCD's L2 once.
#include "gemm_common.h"

int g_gemm_splitk = 1;     // 0 disables split-K

namespace {
// MT = MFMA row tiles per wave: 4 -> the 128x128 block tile, 2 -> 64x128 (twice the blocks for grids that leave CUs idle: the
// training step's M ~ 2k-row GEMMs with N = 1024; 48 KiB of LDS, three blocks per CU)
template <typename T, int ABL, int MT = 4>   // ABL: 0 product kernel; 1 no LDS-DMA; 2 no MFMA; 3 no LDS fragment reads (diagnostics, wrong results)
__global__ __launch_bounds__(256, 2) void gemm_128x128_kernel(KArgs p) {
    constexpr int BMT = 32 * MT, STAGE_T = (BMT + BN) * BK * 2, EPI_T = 16 * MT * EPI_STRIDE * 4;
    typedef typename OpT<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    OFX_LDS char* lds = (OFX_LDS char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap (blocks b and b+8 share an XCD)
    int bid = blockIdx.x;
    {
        const int nx = 8, q = p.nwg / nx, r = p.nwg % nx, x = bid % nx, i = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    // grouped rasterisation: consecutive blocks (= the blocks resident on one XCD at a time) cover
    // group_m row panels x several column tiles, so BOTH the A panels and the W tiles they touch fit the XCD's 4 MiB L2
    int tm, tn;
    {
        const int per_group = p.group_m * p.tiles_n;
        const int gidx = bid / per_group, first = gidx * p.group_m;
        const int gm = min(p.group_m, p.tiles_m - first);
        const int r = bid - gidx * per_group;
        tm = first + r % gm;
        tn = r / gm;
    }
    const int m0 = tm * BMT, n0 = tn * BN;
    if (p.m_dev) {                                      // block-uniform: whole tiles past the live rows leave
        const int m_live = *p.m_dev;
        p.M = m_live < p.M ? m_live : p.M;
        if (m0 >= p.M) return;
    }

    // ---- LDS-DMA source addressing: each wave moves 4 A-chunks and 4 W-chunks of 1 KiB (8 rows) per k-tile
    const int lrow = lane >> 3, lchk = lane & 7;
    const char* a_src[MT];
    const char* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = wave * 4 + i;                     // chunk 0..15 → rows c*8 .. c*8+7
        const int row = c * 8 + lrow;
        const int kch = lchk ^ ((row >> 1) & 7);        // logical 16-B chunk stored at physical slot lchk
        w_src[i] = p.W + ((size_t)(n0 + row) * p.K + kch * 8) * 2;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = (wave * MT + i) * 8 + lrow;
        const int kch = lchk ^ ((row >> 1) & 7);
        int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;
        a_src[i] = p.A + ((size_t)gm * p.lda + kch * 8) * 2;
    }
    const int a_dst = wave * MT * 1024, w_dst = BMT * BK * 2 + wave * 4 * 1024;

    auto issue = [&](int kt, int stage) {
        if (ABL == 1) return;
        OFX_LDS char* base = lds + stage * STAGE_T;
        const size_t koff = (size_t)kt * BK * 2, koff_a = (size_t)(kt % p.ka_tiles) * BK * 2;
#pragma unroll
        for (int i = 0; i < MT; ++i) glds16(a_src[i] + koff_a, base + a_dst + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(w_src[i] + koff, base + w_dst + i * 1024);
    };

    // ---- fragment read addressing (swizzled): row = tile*16 + (lane&15), logical chunk = ks*4 + (lane>>4)
    const int fr = lane & 15, fq = lane >> 4, fsw = fr >> 1;
    const int a_frag = (wm * 16 * MT + fr) * 128;
    const int w_frag = BMT * BK * 2 + (wn * 64 + fr) * 128;

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    v8 af[2][MT], wf[2][4];
    const int kt0 = p.splits > 1 ? blockIdx.y * p.kt_per_split : 0;
    const int nk = p.splits > 1 ? min(p.K / BK, kt0 + p.kt_per_split) : p.K / BK;
    // (round 4: a third stage - two k-tiles in flight - for the 64-row variant on small grids was built and measured on the batch-32 set transformer,
    //  whose 21 GEMMs are this kernel: 0.637 vs 0.638 ms per forward, i.e. nothing; the k-tile time of those launches is not prefetch depth. Removed.)
    issue(kt0, kt0 & 1);
    for (int kt = kt0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            issue(kt + 1, cur ^ 1);
            if (MT == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        OFX_LDS char* base = lds + cur * STAGE_T;
        // all 16 fragment reads of the k-tile go out first; the MFMAs then wait on counted lgkmcnt
        static_assert(true, "");
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chk = ((ks * 4 + fq) ^ fsw) * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (ABL != 3 || kt == kt0) wf[ks][j] = *(OFX_LDS v8*)(base + w_frag + j * 16 * 128 + chk);
#pragma unroll
            for (int i = 0; i < MT; ++i) if (ABL != 3 || kt == kt0) af[ks][i] = *(OFX_LDS v8*)(base + a_frag + i * 16 * 128 + chk);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (ABL == 2) { asm volatile("" :: "v"(wf[ks][j]), "v"(af[ks][i])); }
                    else acc[i][j] = OpT<T>::mfma16(wf[ks][j], af[ks][i], acc[i][j]);
                }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: acc[i][j][r] = C[m = wm*64 + i*16 + (lane&15)][n = wn*64 + j*16 + (lane>>4)*4 + r]
    OFX_LDS float* ep = (OFX_LDS float*)(lds + wave * EPI_T);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *(OFX_LDS f32x4*)(ep + (i * 16 + fr) * EPI_STRIDE + j * 16 + fq * 4) = acc[i][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (p.splits > 1) {                                  // raw partial sums; bias / activation / residual happen in splitk_reduce
        KArgs q = p;
        q.C = (char*)(p.slab + (size_t)blockIdx.y * p.m_slab * p.N); q.ldc = p.N; q.out_kind = 0; q.bias = nullptr; q.resid = nullptr; q.aux_out = nullptr; q.drop.thresh = 0;
        epilogue<T, OFX_ACT_NONE>(q, ep, m0 + wm * 16 * MT, n0 + wn * 64, lane, 16 * MT);
        return;
    }
    switch (p.act) {
        case OFX_ACT_QUICK_GELU: epilogue<T, OFX_ACT_QUICK_GELU>(p, ep, m0 + wm * 16 * MT, n0 + wn * 64, lane, 16 * MT); break;
        case OFX_ACT_GELU: epilogue<T, OFX_ACT_GELU>(p, ep, m0 + wm * 16 * MT, n0 + wn * 64, lane, 16 * MT); break;
        case OFX_ACT_MISH: epilogue<T, OFX_ACT_MISH>(p, ep, m0 + wm * 16 * MT, n0 + wn * 64, lane, 16 * MT); break;
        case OFX_ACT_MISH_GRAD: epilogue<T, OFX_ACT_MISH_GRAD>(p, ep, m0 + wm * 16 * MT, n0 + wn * 64, lane, 16 * MT); break;
        default: epilogue<T, OFX_ACT_NONE>(p, ep, m0 + wm * 16 * MT, n0 + wn * 64, lane, 16 * MT); break;
    }
}


}  // namespace

// Plan for problems the big-tile kernels cannot fill (small M): tile height (128 rows, 2 blocks per CU = 512 slots; or 64 rows, 3 per
// CU = 768 slots) x deterministic split-K over blockIdx.y.  Cost model (us), measured on these kernels: ~1.06 / ~0.65 us per k-tile
// and round of co-resident 128- / 64-row blocks; a split adds the fp32 slab written once and read once (~5 TB/s) and the reduce
// launch.  M = 2304 x N = 1024 (144 blocks) is faster UNSPLIT (17 vs 26 us); M = 288 (K = 3072) wants the split.
// g_gemm_splitk: 0 = never split, 1 = plan, >= 2 = forced (experiments: low byte = splits, bit 8 = 64-row tiles).
struct SmallPlan { int tile64, splits; double cost; };
static SmallPlan plan_small(int M, int N, int K, bool allow_split, bool allow64) {
    const int nk = K / BK;
    SmallPlan best{0, 1, 1e30};
    if (g_gemm_splitk >= 2) {
        SmallPlan f{(g_gemm_splitk >> 8) & 1, (g_gemm_splitk & 0xff) > 16 ? 16 : (g_gemm_splitk & 0xff), 0.0};      // forced plans: at most 16 slices, as the planner's own
        if (f.tile64 && !allow64) f.tile64 = 0;
        if (!allow_split || f.splits > nk) f.splits = 1;
        const int kps = (nk + f.splits - 1) / f.splits;
        f.splits = (nk + kps - 1) / kps;                       // every split owns at least one k-tile
        return f;
    }
    for (int t64 = 0; t64 <= (allow64 ? 1 : 0); ++t64) {
        const long blocks = (long)((M + (t64 ? 63 : 127)) / (t64 ? 64 : 128)) * (N / 128);
        const long slots = t64 ? 768 : 512;
        const double per = t64 ? 0.65 : 1.06;
        for (int sp = 1; sp <= 16; ++sp) {
            if (sp > 1 && !(allow_split && g_gemm_splitk != 0 && blocks < 256 && nk >= 16 && sp <= nk / 4)) break;
            const int kps = (nk + sp - 1) / sp;
            if ((nk + kps - 1) / kps != sp) continue;
            double t = (double)((blocks * sp + slots - 1) / slots) * kps * per + (t64 ? 1.0 : 0.0);
            if (sp > 1) t += 2.0 * sp * M * N * 4.0 / 5.0e6 + 3.0;
            if (t < best.cost) best = SmallPlan{t64, sp, t};
        }
    }
    return best;
}
int ofx_gemm_splitk_plan(int M, int N, int K) { return plan_small(M, N, K, true, M > 64).splits; }
// Slab size for ANY plan the launcher may pick for this shape: it plans with or without 64-row tiles depending on the kernel knobs
// and the grid (can64 in ofx_launch_gemm), so the slab covers the larger slice count of the two tile heights (forced plans: <= 16).
size_t ofx_gemm_splitk_bytes(int M, int N, int K) {
    const int a = plan_small(M, N, K, true, true).splits, b = plan_small(M, N, K, true, false).splits;
    const int sp = a > b ? a : b;
    return sp > 1 ? (size_t)sp * M * N * 4 : 0;
}


int g_w2f8_skew = 0;      // ofx_tune(19, v): start skew of gemm_w2f8_kernel's blocks by XCD (experiment; KArgs::skew)
int g_epi_direct = 1;     // ofx_tune(18, v): 1 (default) gemm_w2f8_kernel's operand-type outputs leave straight from the accumulator layout (epilogue_direct), 0 = through LDS
int g_gemm_group_m = 0;   // 0 = adaptive
int g_gemm_ablate = 0;    // diagnostics only (tools/gemm_bench.py)
unsigned long long* g_gemm_dbg = nullptr;   // diagnostics only
int g_gemm_pref = 2;      // short-K big GEMMs: 0 -> 256x128 kernel, 1 -> 256x256, 2 -> 256x256 ping-pong
int g_w2_fp8 = 1;         // split-weight GEMMs that carry an fp8 copy of their lo halves run the fp8 correction product (gemm_w2f8.hip); 0 = the f16 one, ofx_tune(12, v)
int ofx_w2f8_act_is_bf8();
int g_w2_fp8_ashift = -99; // activations enter the fp8 product as fp8(a 2^shift), ofx_tune(13, v); -99 = the build's default: 0 for the e5m2 image (f16's exponent
                          // range, nothing to choose), 2 for the e4m3 build (-DOFX_F8_ABF8=0: keeps |a| >= 2^-8 out of the subnormal step and saturates at 112)
int g_x3_persist = 1;     // ofx_tune(16, v): 1 (default) gemm_x3_kernel launches one block per CU walking its tiles (when it has more tiles than CUs), 0 = one block per tile
int g_x3_kernel = 1;      // three-product GEMMs (k_mult == 3: A rows [hi | lo | hi], W rows [hi | hi | lo]): 1 = the operand-tiles-loaded-once 256x128 kernel
                          // (gemm_x3.hip) from 192 tiles on, 2 = always, 0 = the K-concatenated single-product kernels; ofx_tune(15, v)
int g_w2_trim = 0;        // 1: persistent split-weight GEMMs shrink their grid to the smallest one with the same round count (measured: +0.4 ms per step), ofx_tune(14, v)
int g_w2_persist = -1;    // dual-weight kernel: persistent grid size (blocks walk tiles b, b + grid, ...): -1 = one block per CU of the device, 0 = one block per tile, ofx_tune(11, v)
int g_gemm_skew = 0;      // start skew of the second co-resident block (x 8128 cycles), 256x128 kernel only
int g_gemm_kernel = 0;    // 0 auto, 1 force 128x128, 2 force 256x256 (8 waves, 2 stages), 3 force 256x128 (4 waves, register-resident k-tile), 4 force 256x256 ping-pong, 6 force the dual-weight 256x256 kernel for split weights

int ofx_launch_gemm(const GemmArgs& g, int op_dtype, hipStream_t s) {
    OFX_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, OFX_ESHAPE, "gemm: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
    OFX_REQUIRE(g.N % BN == 0, OFX_ESHAPE, "gemm: N=%d must be a multiple of %d (pad the weight at pack time)", g.N, BN);
    OFX_REQUIRE(g.K % BK == 0, OFX_ESHAPE, "gemm: K=%d must be a multiple of %d", g.K, BK);
    OFX_REQUIRE(g.a_wrap == 0 || (g.a_wrap > 0 && g.a_wrap % BK == 0 && g.K % g.a_wrap == 0), OFX_ESHAPE, "gemm: a_wrap=%d must be a multiple of %d dividing K=%d", g.a_wrap, BK, g.K);
    const int ka = g.a_wrap ? g.a_wrap : g.K;
    OFX_REQUIRE(g.lda >= ka && g.lda % 8 == 0, OFX_ESHAPE, "gemm: lda=%d must be >= K and a multiple of 8", g.lda);
    OFX_REQUIRE(g.ldc % (g.out_kind == 0 ? 4 : 8) == 0 && g.ldc >= (g.out_kind == 2 ? 3 * g.N : g.N), OFX_ESHAPE, "gemm: bad ldc=%d", g.ldc);
    OFX_REQUIRE(!g.resid || (g.ldr % 4 == 0 && g.ldr >= g.N), OFX_ESHAPE, "gemm: bad ldr=%d", g.ldr);
    OFX_REQUIRE(((uintptr_t)g.A % 16 == 0) && ((uintptr_t)g.W % 16 == 0) && ((uintptr_t)g.C % 16 == 0), OFX_EINVAL,
                "gemm: operands must be 16-byte aligned");
    OFX_REQUIRE(op_dtype == OFX_BF16 || op_dtype == OFX_F16, OFX_EINVAL, "gemm: operand dtype must be bf16 or f16");
#ifndef OFX_DIAG
    OFX_REQUIRE(g_gemm_ablate == 0, OFX_ESTATE, "gemm: the ablation kernels are only built with `make DIAG=1`");
#endif
    if (g.defer_splits) *g.defer_splits = 1;
    if (g.ln_done) *g.ln_done = false;
    KArgs k;
    k.A = (const char*)g.A; k.W = (const char*)g.W; k.C = (char*)g.C; k.bias = g.bias; k.resid = g.resid; k.aux_out = g.aux_out; k.m_dev = g.m_dev; k.dbg = g_gemm_dbg; k.skew = g_gemm_skew; k.splits = 1; k.slab = nullptr; k.m_slab = g.M; k.kt_per_split = 0;
    k.ka_tiles = ka / BK;
    k.M = g.M; k.N = g.N; k.K = g.K; k.lda = g.lda; k.ldc = g.ldc; k.ldr = g.ldr; k.act = g.act; k.out_kind = g.out_kind; k.drop = g.drop; k.n_valid = g.N;
    k.xb_out = (char*)g.xb_out; k.stat_part = g.stat_part; k.row_stat = g.row_stat; k.col_sum = g.col_sum; k.stat_ld = g.stat_ld > 0 ? g.stat_ld : 1; k.xlo = (char*)g.xlo;
    OFX_REQUIRE(!g.xlo || (g.xb_out && g.stat_part), OFX_EINVAL, "gemm: xlo needs the LayerNorm-fold producer outputs");
    OFX_REQUIRE(!(g.xb_out || g.stat_part) || ((g.out_kind == 0 || (g.xlo && g.out_kind == 1 && g.C == g.xb_out && g.ldc == g.N)) && g.N % 64 == 0), OFX_EINVAL,
                "gemm: LayerNorm-fold producer outputs need an fp32 output (or, with xlo, C == xb_out in the operand type)");
    OFX_REQUIRE(!g.row_stat || g.col_sum, OFX_EINVAL, "gemm: row_stat needs col_sum");
    OFX_REQUIRE(!(g.row_stat || g.xb_out || g.stat_part) || (!g.aux_out && !g.drop.thresh && g.act != OFX_ACT_MISH && g.act != OFX_ACT_MISH_GRAD), OFX_EINVAL,
                "gemm: LayerNorm folding does not combine with the training epilogue features");
    OFX_REQUIRE(!g.row_stat || !g.resid, OFX_EINVAL, "gemm: a LayerNorm-fold consumer takes no residual");
    OFX_REQUIRE(!(g.xb_out || g.stat_part) || g.act == OFX_ACT_NONE, OFX_EINVAL, "gemm: a LayerNorm-fold producer has no activation");
    static DeviceOnce attr_set;
    TRY(attr_set.run([]() -> int {
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_128x128_kernel<bf16_t, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_128x128_kernel<f16_t, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_128x128_kernel<bf16_t, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM64_LDS_BYTES));
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_128x128_kernel<f16_t, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM64_LDS_BYTES));
#ifdef OFX_DIAG
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_128x128_kernel<bf16_t, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_128x128_kernel<bf16_t, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_128x128_kernel<bf16_t, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
#endif
        return OFX_OK;
    }));
    // big tiles when they still fill the chip, else the 128^2 kernel
    int kind = g_gemm_kernel == 6 ? 0 : g_gemm_kernel;      // 6 only forces the kernel of split-weight GEMMs; every other GEMM keeps the automatic choice
    if (g.a_wrap) {        // split weights: the dual-weight 256x256 kernel when its grid fills the chip, else the 128x128 kernel with a wrapping A index
        const long t2 = (long)((g.M + 255) / 256) * (g.N / 256);
        kind = (g.K == 2 * g.a_wrap && g.a_wrap % 32 == 0 && g.a_wrap >= 64 && g.N % 256 == 0 && (t2 >= 256 || g_gemm_kernel == 6) && g_gemm_kernel != 1) ? 6 : 1;
        // the correction product on the fp8 matrix instruction (ofx_tune(12, 0) keeps the f16 one)
        if (kind == 6 && g.W8 && g.w8_scale && g_w2_fp8 && op_dtype == OFX_F16 && g.a_wrap % 128 == 0) kind = 8;
    } else if (kind == 0) {   // measured crossover points (tools/gemm_bench.py, profiles/r01_gemm_variants.txt)
        const long t2 = (long)((g.M + 255) / 256) * (g.N / 256), t3 = (long)((g.M + 255) / 256) * (g.N / 128);
        if (g.N % 256 == 0 && t2 >= 1024 && (g.K > 1024 || g_gemm_pref >= 1)) kind = g_gemm_pref == 2 && g.K <= 1024 ? 4 : 2;   // 256x256, one block per CU
        else if (g.N % 128 == 0 && t3 >= 512) kind = 3;                    // short K / mid-size M: 256x128, two blocks per CU
        else kind = 1;
    }
    if (g.k_mult == 3 && !g.a_wrap && g_x3_kernel && (g_gemm_kernel == 0 || g_gemm_kernel == 6) && g.K % 96 == 0 && g.N % 128 == 0 && g.lda >= g.K &&
        (g_x3_kernel >= 2 || (long)((g.M + 255) / 256) * (g.N / 128) >= 192))
        kind = 9;                  // the three products from ONE copy of each operand tile
    if ((kind == 2 || kind == 4) && g.N % 256) kind = 1;
    if (kind == 5 && g.N % 128) kind = 1;
    // record label: logical shape (K without the split-weight / three-product concatenation), the K multiplier and
    // kernel kind (1 128x128 [+ split-K / 64-row variants], 2 256x256, 3 256x128, 4 256x256 ping-pong, 6 dual-weight 256x256, 8 the same with the fp8 correction product, 9 three-product 256x128 with the operand tiles loaded once)
    if (g_ofx_prof_on) {
        const int km = g.a_wrap ? g.K / g.a_wrap : (g.k_mult > 0 ? g.k_mult : 1);
        // algorithmic HBM bytes of this launch: A once (its k index wraps over a_wrap columns for split weights), the weight rows as
        // stored, and per output element what the configured epilogue moves: fp32 4 (+4 residual read, +2 operand copy), operand
        // type 2 (in-place (hi, lo) stream: 4 read + 4 written), [hi | lo | hi] 6, +4 pre-activation tape copy
        const double ab = kind == 9 ? 2.0 * g.M * (g.K / 3) * 2 : 2.0 * g.M * (g.a_wrap ? g.a_wrap : g.K), wb = kind == 8 ? 3.0 * g.N * g.a_wrap : (kind == 9 ? 2.0 * g.N * (g.K / 3) * 2 : 2.0 * g.N * g.K);      // kind 8 reads the hi rows (2 B) + the fp8 lo rows (1 B)
        double ob = g.out_kind == 0 ? 4.0 : (g.out_kind == 2 ? 6.0 : 2.0);
        if (g.xlo) ob = 8.0;
        else { if (g.resid) ob += 4.0; if (g.xb_out) ob += 2.0; }
        if (g.aux_out) ob += 4.0;
        ofx_prof_set_tag(g.M, g.N, g.K / km, kind, km, ab + wb + ob * g.M * g.N);
    }
    // executed FLOPs in f16-rate equivalents: the fp8 correction product of kind 8 runs at twice the f16 rate (1.5 products, not 2)
    ProfScope prof(PROF_GEMM, s, 2.0 * g.M * g.N * g.K * (kind == 8 ? 0.75 : 1.0), true);      // events ride on the launches (OFX_PLAUNCH)
    if (kind == 2 || kind == 3 || kind == 4 || kind == 6 || kind == 8 || kind == 9) {
        k.group_m = g_gemm_group_m > 0 ? g_gemm_group_m : (kind == 3 ? 4 : 8);
        int rc;
        if (kind == 8) {
            const int ash = g_w2_fp8_ashift == -99 ? (ofx_w2f8_act_is_bf8() ? 0 : 2) : g_w2_fp8_ashift;
            k.W8 = (const char*)g.W8; k.w8_scale = (const char*)g.w8_scale; k.a8_scale = ldexpf(1.0f, -ash); k.a8_e8m0 = 127 - ash; rc = ofx_gemm_launch_w2f8(&k, g.M, g.N, s);
        }
        else if (kind == 9) rc = ofx_gemm_launch_x3(&k, op_dtype, g.M, g.N, s);
        else if (kind == 6) rc = ofx_gemm_launch_w2(&k, op_dtype, g.M, g.N, s);
        else if (kind == 4) rc = ofx_gemm_launch_pp(&k, op_dtype, g.M, g.N, s);
        else rc = ofx_gemm_launch_big(&k, kind, g_gemm_ablate, op_dtype, g.M, g.N, s);
        if (rc != OFX_OK) return rc;
    } else {
        k.tiles_n = g.N / BN; k.tiles_m = (g.M + BM - 1) / BM; k.nwg = k.tiles_m * k.tiles_n;
        // row panels per L2 group: (group_m + 64/group_m) panels of 128 x K operands should fit ~3 MiB of the XCD's L2
        int gm = g_gemm_group_m;
        if (gm <= 0) { gm = (int)((3u << 20) / ((size_t)BM * g.K * 2) / 2); gm = gm < 1 ? 1 : (gm > 8 ? 8 : gm); }
        k.group_m = gm;
        const bool can_split = g.slab && !g.xb_out && !g.stat_part && !g.row_stat;
        const bool can64 = (kind == 5) || (kind == 1 && g_gemm_kernel == 0 && g.M > 64 && (long)k.tiles_m * k.tiles_n <= 384);
        SmallPlan pl = plan_small(g.M, g.N, g.K, can_split, can64);
        if (kind == 5) pl.tile64 = 1;
        const int splits = pl.splits;
        k.splits = splits; k.slab = (float*)g.slab; k.m_slab = g.M;
        k.kt_per_split = splits > 1 ? (g.K / BK + splits - 1) / splits : 0;
        if (splits > 1) OFX_REQUIRE(g.slab_bytes >= (size_t)splits * g.M * g.N * 4, OFX_EWORKSPACE, "gemm: split-K slab too small");
        // second pass of a split-K plan: left to the consumer (defer_splits), fused with the following LayerNorm (ln_gamma), or the plain reduce
        auto second_pass = [&]() -> int {
            if (g.defer_splits) { *g.defer_splits = splits; return OFX_OK; }
            if (g.ln_gamma && g.out_kind == 0 && g.act == OFX_ACT_NONE && !g.aux_out && !g.drop.thresh && (g.N == 512 || g.N == 768 || g.N == 1024)) {
                SplitKLnArgs a{(const float*)g.slab, (size_t)g.M * g.N, splits, g.bias, g.resid, g.ldr, (float*)g.C, g.ldc,
                               g.ln_gamma, g.ln_beta, g.ln_out, g.ln_ld, g.ln_kind, g.ln_eps, g.M, g.N, g.m_dev};
                TRY(ofx_launch_splitk_reduce_ln(a, op_dtype, s, true));
                if (g.ln_done) *g.ln_done = true;
                return OFX_OK;
            }
            size_t tot = (size_t)g.M * (g.N / 4);
            int rg = (int)((tot + 255) / 256); if (rg > 2048) rg = 2048;
            if (op_dtype == OFX_F16) OFX_PLAUNCH(true, splitk_reduce_kernel<f16_t>, dim3(rg), dim3(256), 0, s, k);
            else OFX_PLAUNCH(true, splitk_reduce_kernel<bf16_t>, dim3(rg), dim3(256), 0, s, k);
            return OFX_OK;
        };
        if (pl.tile64) {        // 64-row tiles, three blocks per CU: grids that would leave CUs idle with 128-row tiles (with or without split-K)
            k.tiles_m = (g.M + 63) / 64; k.nwg = k.tiles_m * k.tiles_n; k.group_m = 2 * gm;
            const dim3 grid64(k.nwg, splits);
            const bool one64 = splits <= 1 || g.defer_splits;      // the launch that carries the profile record's stop event
            if (op_dtype == OFX_F16) OFX_PLAUNCH(one64, (gemm_128x128_kernel<f16_t, 0, 2>), grid64, dim3(256), GEMM64_LDS_BYTES, s, k);
            else OFX_PLAUNCH(one64, (gemm_128x128_kernel<bf16_t, 0, 2>), grid64, dim3(256), GEMM64_LDS_BYTES, s, k);
            if (splits > 1) TRY(second_pass());
            OFX_LAUNCH_CHECK();
            return OFX_OK;
        }
        const dim3 grid(k.nwg, splits > 1 ? splits : 1);
        const bool one = splits <= 1 || g.defer_splits;
        if (op_dtype == OFX_F16) OFX_PLAUNCH(one, (gemm_128x128_kernel<f16_t, 0>), grid, dim3(256), GEMM_LDS_BYTES, s, k);
#ifdef OFX_DIAG
        else if (g_gemm_ablate == 1) OFX_PLAUNCH(one, (gemm_128x128_kernel<bf16_t, 1>), grid, dim3(256), GEMM_LDS_BYTES, s, k);
        else if (g_gemm_ablate == 2) OFX_PLAUNCH(one, (gemm_128x128_kernel<bf16_t, 2>), grid, dim3(256), GEMM_LDS_BYTES, s, k);
        else if (g_gemm_ablate == 3) OFX_PLAUNCH(one, (gemm_128x128_kernel<bf16_t, 3>), grid, dim3(256), GEMM_LDS_BYTES, s, k);
#endif
        else OFX_PLAUNCH(one, (gemm_128x128_kernel<bf16_t, 0>), grid, dim3(256), GEMM_LDS_BYTES, s, k);
        if (splits > 1) TRY(second_pass());
    }
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
