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
#include <type_traits>

#include "ofx_common.h"

int g_gemm_splitk = 1;     // 0 disables split-K

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;        // 32 KiB
constexpr int EPI_STRIDE = 68;                         // floats per staged output row (64 + 4 pad)
constexpr int EPI_BYTES_PER_WAVE = 64 * EPI_STRIDE * 4;
constexpr int GEMM_LDS_BYTES = 4 * EPI_BYTES_PER_WAVE > 2 * STAGE_BYTES ? 4 * EPI_BYTES_PER_WAVE : 2 * STAGE_BYTES;

struct KArgs {
    const char* A;
    const char* W;
    char* C;
    const float* bias;
    const float* resid;
    float* aux_out;     // optional fp32 [M, N] pre-activation copy
    const int* m_dev;   // optional device-side row count (pad-free varlen sets); M is then the upper bound
    unsigned long long* dbg;   // diagnostics only (tools/gemm_bench.py --clock): per block {shader cycles, 100 MHz ticks} of the main loop
    int M, N, K, lda, ldc, ldr, act, out_kind, tiles_n, tiles_m, nwg, group_m, skew;
    int m_slab;                 // rows per slab plane (the host-side M, never the clamped live count)
    int splits, kt_per_split;   // 128x128 kernel only: blockIdx.y owns k-tiles [y*kt_per_split, ...) and writes a raw fp32 slab
    float* slab;                // [splits, M, N] partial sums when splits > 1
    DropArgs drop;
    char* xb_out; float* stat_part; const float* row_stat; const float* col_sum;   // LayerNorm folding (GemmArgs)
};

__device__ __forceinline__ void glds16(const char* g, OFX_LDS char* l) {
    __builtin_amdgcn_global_load_lds((const OFX_GLB void*)g, (OFX_LDS void*)l, 16, 0, 0);
}


// One wave drains its 64x64 fp32 sub-tile from LDS as whole row segments: 16 lanes x 16 B per row.
template <typename T, int ACT>
__device__ __forceinline__ void epilogue(const KArgs& p, OFX_LDS float* ep, int gm0, int gn0, int lane) {
    typedef typename OpT<T>::v4 v4;
    const int col = (lane & 15) * 4;
    const int gn = gn0 + col;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *(const f32x4*)(p.bias + gn);
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int row = it * 4 + (lane >> 4);
        const int gm = gm0 + row;
        f32x4 v = *(OFX_LDS f32x4*)(ep + row * EPI_STRIDE + col);
        if (gm < p.M) {
            if (p.row_stat) {
                const float mu = p.row_stat[2 * (size_t)gm], rs = p.row_stat[2 * (size_t)gm + 1];
                v = (v - *(const f32x4*)(p.col_sum + gn) * mu) * rs + bias4;
            } else v += bias4;
            if (p.aux_out) *(f32x4*)(p.aux_out + (size_t)gm * p.N + gn) = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (ACT == OFX_ACT_QUICK_GELU) v[e] = act_quick_gelu(v[e]);
                else if (ACT == OFX_ACT_GELU) v[e] = act_gelu(v[e]);
                else if (ACT == OFX_ACT_MISH) v[e] = act_mish(v[e]);
            }
            if (p.drop.thresh) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= drop_mul(p.drop, gm, gn + e);
            }
            if (p.resid) {
                const f32x4 rr = *(const f32x4*)(p.resid + (size_t)gm * p.ldr + gn);
                if (ACT == OFX_ACT_MISH_GRAD) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= act_mish_grad(rr[e]);
                } else v += rr;
            }
            if (p.out_kind == 0) {
                *(f32x4*)(p.C + ((size_t)gm * p.ldc + gn) * 4) = v;
                if (p.xb_out) {
                    v4 hb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hb[e] = (T)v[e];
                    *(v4*)((T*)p.xb_out + (size_t)gm * p.N + gn) = hb;
                }
                if (p.stat_part) {          // gm is uniform over the 16 lanes that share this row
                    const float ssum = row16_sum_to_lane15((v[0] + v[1]) + (v[2] + v[3]));
                    const float ssq = row16_sum_to_lane15((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
                    if ((lane & 15) == 15) *(f32x2*)(p.stat_part + ((size_t)gm * (p.N >> 6) + (gn0 >> 6)) * 2) = f32x2{ssum, ssq};
                }
            } else {
                v4 hi;
#pragma unroll
                for (int e = 0; e < 4; ++e) hi[e] = (T)v[e];
                T* crow = (T*)p.C + (size_t)gm * p.ldc + gn;
                *(v4*)crow = hi;
                if (p.out_kind == 2) {
                    v4 lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) lo[e] = (T)(v[e] - (float)hi[e]);
                    *(v4*)(crow + p.N) = lo;
                    *(v4*)(crow + 2 * p.N) = hi;
                }
            }
        }
    }
}

template <typename T, int ABL>   // ABL: 0 product kernel; 1 no LDS-DMA; 2 no MFMA; 3 no LDS fragment reads (diagnostics, wrong results)
__global__ __launch_bounds__(256, 2) void gemm_128x128_kernel(KArgs p) {
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
    const int m0 = tm * BM, n0 = tn * BN;
    if (p.m_dev) {                                      // block-uniform: whole tiles past the live rows leave
        const int m_live = *p.m_dev;
        p.M = m_live < p.M ? m_live : p.M;
        if (m0 >= p.M) return;
    }

    // ---- LDS-DMA source addressing: each wave moves 4 A-chunks and 4 W-chunks of 1 KiB (8 rows) per k-tile
    const int lrow = lane >> 3, lchk = lane & 7;
    const char* a_src[4];
    const char* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = wave * 4 + i;                     // chunk 0..15 → rows c*8 .. c*8+7
        const int row = c * 8 + lrow;
        const int kch = lchk ^ ((row >> 1) & 7);        // logical 16-B chunk stored at physical slot lchk
        int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;
        a_src[i] = p.A + ((size_t)gm * p.lda + kch * 8) * 2;
        w_src[i] = p.W + ((size_t)(n0 + row) * p.K + kch * 8) * 2;
    }
    const int a_dst = wave * 4 * 1024, w_dst = BM * BK * 2 + wave * 4 * 1024;

    auto issue = [&](int kt, int stage) {
        if (ABL == 1) return;
        OFX_LDS char* base = lds + stage * STAGE_BYTES;
        const size_t koff = (size_t)kt * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(a_src[i] + koff, base + a_dst + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(w_src[i] + koff, base + w_dst + i * 1024);
    };

    // ---- fragment read addressing (swizzled): row = tile*16 + (lane&15), logical chunk = ks*4 + (lane>>4)
    const int fr = lane & 15, fq = lane >> 4, fsw = fr >> 1;
    const int a_frag = (wm * 64 + fr) * 128;
    const int w_frag = BM * BK * 2 + (wn * 64 + fr) * 128;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    v8 af[2][4], wf[2][4];
    const int kt0 = p.splits > 1 ? blockIdx.y * p.kt_per_split : 0;
    const int nk = p.splits > 1 ? min(p.K / BK, kt0 + p.kt_per_split) : p.K / BK;
    issue(kt0, kt0 & 1);
    for (int kt = kt0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            issue(kt + 1, cur ^ 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        OFX_LDS char* base = lds + cur * STAGE_BYTES;
        // all 16 fragment reads of the k-tile go out first; the MFMAs then wait on counted lgkmcnt
        static_assert(true, "");
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chk = ((ks * 4 + fq) ^ fsw) * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (ABL != 3 || kt == kt0) wf[ks][j] = *(OFX_LDS v8*)(base + w_frag + j * 16 * 128 + chk);
#pragma unroll
            for (int i = 0; i < 4; ++i) if (ABL != 3 || kt == kt0) af[ks][i] = *(OFX_LDS v8*)(base + a_frag + i * 16 * 128 + chk);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
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
    OFX_LDS float* ep = (OFX_LDS float*)(lds + wave * EPI_BYTES_PER_WAVE);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *(OFX_LDS f32x4*)(ep + (i * 16 + fr) * EPI_STRIDE + j * 16 + fq * 4) = acc[i][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (p.splits > 1) {                                  // raw partial sums; bias / activation / residual happen in splitk_reduce
        KArgs q = p;
        q.C = (char*)(p.slab + (size_t)blockIdx.y * p.m_slab * p.N); q.ldc = p.N; q.out_kind = 0; q.bias = nullptr; q.resid = nullptr; q.aux_out = nullptr; q.drop.thresh = 0;
        epilogue<T, OFX_ACT_NONE>(q, ep, m0 + wm * 64, n0 + wn * 64, lane);
        return;
    }
    switch (p.act) {
        case OFX_ACT_QUICK_GELU: epilogue<T, OFX_ACT_QUICK_GELU>(p, ep, m0 + wm * 64, n0 + wn * 64, lane); break;
        case OFX_ACT_GELU: epilogue<T, OFX_ACT_GELU>(p, ep, m0 + wm * 64, n0 + wn * 64, lane); break;
        case OFX_ACT_MISH: epilogue<T, OFX_ACT_MISH>(p, ep, m0 + wm * 64, n0 + wn * 64, lane); break;
        case OFX_ACT_MISH_GRAD: epilogue<T, OFX_ACT_MISH_GRAD>(p, ep, m0 + wm * 64, n0 + wn * 64, lane); break;
        default: epilogue<T, OFX_ACT_NONE>(p, ep, m0 + wm * 64, n0 + wn * 64, lane); break;
    }
}


// ================================================================================================
// v2: 256x256x64 tile, 512 threads = 8 waves (2 x 4), wave tile 128x64 = 8x4 MFMA tiles x 2 k-steps.
// Half the LDS-fill bytes and two thirds of the LDS fragment reads per FLOP of the 128^2 kernel
// (the ablation in DESIGN.md §4 shows the fill path, not the MFMA pipe, bounds that kernel).
// Per k-tile: [vmcnt -> barrier -> 24 ds_read_b128 into registers -> barrier] frees the stage at
// once, so the LDS-DMA of k-tile t+2 is issued before the 64 MFMAs of k-tile t and two k-tiles
// (128 KiB per CU) stay in flight.  LDS: 2 stages x 64 KiB + 32 KiB epilogue staging = 160 KiB.
constexpr int EPI2_BYTES_PER_WAVE = 16 * 64 * 4;       // 16 rows x 64 fp32, XOR-swizzled, no padding

// Epilogue of the 128x64 wave tile: 8 passes of 16 rows through the wave's private LDS staging (XOR-swizzled
// 16-B chunks), leaving as whole 128/256-byte row segments with 16-byte stores (the store tail is issue-bound:
// guide T21).  fp32 output: 16 lanes x 4 columns per row, the fp32 residual of pass i+2 requested while pass i is
// written out.  bf16/f16 output: 8 lanes x 8 columns per row -> one dwordx4 store per lane instead of two dwordx2.
template <typename T, int ACT>
__device__ __forceinline__ float act_apply(float v) {
    if (ACT == OFX_ACT_QUICK_GELU) return act_quick_gelu(v);
    if (ACT == OFX_ACT_GELU) return act_gelu(v);
    if (ACT == OFX_ACT_MISH) return act_mish(v);
    return v;
}

// FOLD (LayerNorm folding, compile-time so the common path keeps its registers): 0 none, 1 producer (fp32 output + operand copy
// + per-segment statistics), 2 consumer (row statistics + column sums applied to the accumulator).
template <typename T, int ACT, int FOLD = 0>
__device__ __forceinline__ void epilogue2(const KArgs& p, OFX_LDS char* ep, f32x4 (&acc)[8][4], int gm0, int gn0, int lane) {
    typedef typename OpT<T>::v8 v8;
    const int fr = lane & 15, fq = lane >> 4;
    if (p.out_kind == 0) {
        constexpr int DEPTH = 2;
        const int chunk = lane & 15, rsub = lane >> 4;
        const int gn = gn0 + chunk * 4;
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bias4 = *(const f32x4*)(p.bias + gn);
        const bool has_res = p.resid != nullptr;
        f32x4 res[DEPTH + 1][4];
        auto fetch = [&](int pass, f32x4 (&dst)[4]) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int gm = gm0 + pass * 16 + it * 4 + rsub;
                dst[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (has_res && gm < p.M) dst[it] = *(const f32x4*)(p.resid + (size_t)gm * p.ldr + gn);
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) fetch(d, res[d]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i + DEPTH < 8) fetch(i + DEPTH, res[(i + DEPTH) % (DEPTH + 1)]);
#pragma unroll
            for (int j = 0; j < 4; ++j) *(OFX_LDS f32x4*)(ep + fr * 256 + (((j * 4 + fq) ^ (fr & 7)) << 4)) = acc[i][j];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 4 + rsub;
                const int gm = gm0 + i * 16 + row;
                f32x4 v = *(OFX_LDS f32x4*)(ep + row * 256 + ((chunk ^ (row & 7)) << 4));
                if (gm < p.M) {
                    if (FOLD == 2) {
                        const float mu = p.row_stat[2 * (size_t)gm], rs = p.row_stat[2 * (size_t)gm + 1];
                        v = (v - *(const f32x4*)(p.col_sum + gn) * mu) * rs + bias4;
                    } else v += bias4;
                    if (FOLD == 0 && p.aux_out) *(f32x4*)(p.aux_out + (size_t)gm * p.N + gn) = v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_apply<T, ACT>(v[e]);
                    if (FOLD == 0 && p.drop.thresh) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= drop_mul(p.drop, gm, gn + e);
                    }
                    if (ACT == OFX_ACT_MISH_GRAD) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= act_mish_grad(res[i % (DEPTH + 1)][it][e]);
                    } else v += res[i % (DEPTH + 1)][it];
                    *(f32x4*)(p.C + ((size_t)gm * p.ldc + gn) * 4) = v;
                    if (FOLD == 1 && p.xb_out) {
                        typename OpT<T>::v4 hb;
#pragma unroll
                        for (int e = 0; e < 4; ++e) hb[e] = (T)v[e];
                        *(typename OpT<T>::v4*)((T*)p.xb_out + (size_t)gm * p.N + gn) = hb;
                    }
                    if (FOLD == 1 && p.stat_part) {      // gm is uniform over the 16 lanes (same rsub) that share this row
                        const float ssum = row16_sum_to_lane15((v[0] + v[1]) + (v[2] + v[3]));
                        const float ssq = row16_sum_to_lane15((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
                        if (chunk == 15) *(f32x2*)(p.stat_part + ((size_t)gm * (p.N >> 6) + (gn0 >> 6)) * 2) = f32x2{ssum, ssq};
                    }
                }
            }
        }
    } else {
        const int c8 = lane & 7, rsub = lane >> 3;          // 8 columns per lane, 8 rows per wave-instruction
        const int gn = gn0 + c8 * 8;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (p.bias) { b0 = *(const f32x4*)(p.bias + gn); b1 = *(const f32x4*)(p.bias + gn + 4); }
        f32x4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = cs0;
        if (FOLD == 2) { cs0 = *(const f32x4*)(p.col_sum + gn); cs1 = *(const f32x4*)(p.col_sum + gn + 4); }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *(OFX_LDS f32x4*)(ep + fr * 256 + (((j * 4 + fq) ^ (fr & 7)) << 4)) = acc[i][j];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 8 + rsub;
                const int gm = gm0 + i * 16 + row;
                f32x4 v0 = *(OFX_LDS f32x4*)(ep + row * 256 + (((2 * c8) ^ (row & 7)) << 4));
                f32x4 v1 = *(OFX_LDS f32x4*)(ep + row * 256 + (((2 * c8 + 1) ^ (row & 7)) << 4));
                if (gm < p.M) {
                    if (FOLD == 2) {
                        const float mu = p.row_stat[2 * (size_t)gm], rs = p.row_stat[2 * (size_t)gm + 1];
                        v0 = (v0 - cs0 * mu) * rs + b0; v1 = (v1 - cs1 * mu) * rs + b1;
                    } else { v0 += b0; v1 += b1; }
                    if (FOLD == 0 && p.aux_out) { *(f32x4*)(p.aux_out + (size_t)gm * p.N + gn) = v0; *(f32x4*)(p.aux_out + (size_t)gm * p.N + gn + 4) = v1; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] = act_apply<T, ACT>(v0[e]); v1[e] = act_apply<T, ACT>(v1[e]); }
                    if (FOLD == 0 && p.drop.thresh) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] *= drop_mul(p.drop, gm, gn + e); v1[e] *= drop_mul(p.drop, gm, gn + 4 + e); }
                    }
                    if (p.resid) {
                        const f32x4 r0 = *(const f32x4*)(p.resid + (size_t)gm * p.ldr + gn), r1 = *(const f32x4*)(p.resid + (size_t)gm * p.ldr + gn + 4);
                        if (ACT == OFX_ACT_MISH_GRAD) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v0[e] *= act_mish_grad(r0[e]); v1[e] *= act_mish_grad(r1[e]); }
                        } else { v0 += r0; v1 += r1; }
                    }
                    v8 hi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { hi[e] = (T)v0[e]; hi[4 + e] = (T)v1[e]; }
                    T* crow = (T*)p.C + (size_t)gm * p.ldc + gn;
                    *(v8*)crow = hi;
                    if (p.out_kind == 2) {
                        v8 lo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { lo[e] = (T)(v0[e] - (float)hi[e]); lo[4 + e] = (T)(v1[e] - (float)hi[4 + e]); }
                        *(v8*)(crow + p.N) = lo;
                        *(v8*)(crow + 2 * p.N) = hi;
                    }
                }
            }
        }
    }
}

template <typename T>
__device__ __forceinline__ void epilogue2_dispatch(const KArgs& p, OFX_LDS char* ep, f32x4 (&acc)[8][4], int gm0, int gn0, int lane) {
    if (p.row_stat) {                                  // LayerNorm-fold consumer: towers only (no residual, no dropout, no tape)
        switch (p.act) {
            case OFX_ACT_QUICK_GELU: epilogue2<T, OFX_ACT_QUICK_GELU, 2>(p, ep, acc, gm0, gn0, lane); break;
            case OFX_ACT_GELU: epilogue2<T, OFX_ACT_GELU, 2>(p, ep, acc, gm0, gn0, lane); break;
            default: epilogue2<T, OFX_ACT_NONE, 2>(p, ep, acc, gm0, gn0, lane); break;
        }
        return;
    }
    if (p.xb_out || p.stat_part) { epilogue2<T, OFX_ACT_NONE, 1>(p, ep, acc, gm0, gn0, lane); return; }
    switch (p.act) {
        case OFX_ACT_QUICK_GELU: epilogue2<T, OFX_ACT_QUICK_GELU>(p, ep, acc, gm0, gn0, lane); break;
        case OFX_ACT_GELU: epilogue2<T, OFX_ACT_GELU>(p, ep, acc, gm0, gn0, lane); break;
        case OFX_ACT_MISH: epilogue2<T, OFX_ACT_MISH>(p, ep, acc, gm0, gn0, lane); break;
        case OFX_ACT_MISH_GRAD: epilogue2<T, OFX_ACT_MISH_GRAD>(p, ep, acc, gm0, gn0, lane); break;
        default: epilogue2<T, OFX_ACT_NONE>(p, ep, acc, gm0, gn0, lane); break;
    }
}

// Split-K second pass: out = epilogue( sum_s slab[s] ) in a fixed order (deterministic), 4 columns per thread.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(KArgs p) {
    typedef typename OpT<T>::v4 v4;
    const int M = p.m_dev ? min(*p.m_dev, p.M) : p.M;
    const int n4 = p.N / 4;
    const size_t total = (size_t)M * n4, plane = (size_t)p.m_slab * p.N;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int gm = (int)(i / n4), gn = (int)(i % n4) * 4;
        const float* sp = p.slab + (size_t)gm * p.N + gn;
        f32x4 v = *(const f32x4*)sp;
        for (int s = 1; s < p.splits; ++s) v += *(const f32x4*)(sp + s * plane);
        if (p.bias) v += *(const f32x4*)(p.bias + gn);
        if (p.aux_out) *(f32x4*)(p.aux_out + (size_t)gm * p.N + gn) = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
        if (p.drop.thresh) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= drop_mul(p.drop, gm, gn + e);
        }
        if (p.resid) {
            const f32x4 rr = *(const f32x4*)(p.resid + (size_t)gm * p.ldr + gn);
            if (p.act == OFX_ACT_MISH_GRAD) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= act_mish_grad(rr[e]);
            } else v += rr;
        }
        if (p.out_kind == 0) {
            *(f32x4*)(p.C + ((size_t)gm * p.ldc + gn) * 4) = v;
        } else {
            v4 hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) hi[e] = (T)v[e];
            T* crow = (T*)p.C + (size_t)gm * p.ldc + gn;
            *(v4*)crow = hi;
            if (p.out_kind == 2) {
                v4 lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) lo[e] = (T)(v[e] - (float)hi[e]);
                *(v4*)(crow + p.N) = lo;
                *(v4*)(crow + 2 * p.N) = hi;
            }
        }
    }
}

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
    epilogue2_dispatch<T>(p, ep, acc, gm0, gn0, lane);
    if (p.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) p.dbg[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memtime() - cloop_end;
    }
}

// ------------------------------------------------------------------------------------------------
// Ping-pong variant of the 256x256 tile (8 waves): waves 0-3 (group 0, rows 0-127) and waves 4-7 (group 1) run
// the same per-k-tile program offset by ONE barrier slot, so on every SIMD one wave reads its 24 fragments and
// issues LDS-DMA while the other wave runs its 64 MFMAs.  Slot schedule (B = block barrier, j = k-tile):
//   group 0:  [W0] B [R0] B [M0 W1] B [R1 I2] B [M1 W2] B [R2 I3] B ...
//   group 1:  [W0] B [  ] B [R0 W1] B [M0 I2] B [R1 W2] B [M1 I3] B ...
// k-tile j >= 2 is issued by both groups in slot 2j-1 (after both groups read k-tile j-2, slots 2j-3 / 2j-2: WAR),
// waited for (vmcnt(0)) at the end of slot 2j and read in slots 2j+1 / 2j+2 (RAW: every wave's wait precedes the
// barrier that opens slot 2j+1).  Two 64 KiB stages + 32 KiB private epilogue staging.
template <typename T>
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
        for (int i = 0; i < 4; ++i) glds16(ak + a_off[i], base + a_dst + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(wk + w_off[i], base + w_dst + i * 1024);
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
                    if (q < 4) glds16(ak + a_off[q], nbase + a_dst + q * 1024);
                    else glds16(wk + w_off[q - 4], nbase + w_dst + (q - 4) * 1024);
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
    epilogue2_dispatch<T>(p, ep, acc, gm0, gn0, lane);
}


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
    q.M = p.M; q.N = p.N; q.ldc = p.splits > 1 ? p.N : p.ldc; q.out_kind = 0;
    OFX_LDS char* ep = lds + 2 * STAGE + wave * EPI2_BYTES_PER_WAVE;
    epilogue2<T, OFX_ACT_NONE>(q, ep, acc, m0 + wr * 128, n0 + wc * 64, lane);
}

template <typename T>
static int launch_pp(KArgs& k, int M, int N, hipStream_t s) {
    constexpr int LDSB = 2 * (256 + 256) * BK * 2 + 8 * EPI2_BYTES_PER_WAVE;
    static bool attr = false;
    if (!attr) {
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_pp_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        attr = true;
    }
    k.tiles_n = N / 256; k.tiles_m = (M + 255) / 256; k.nwg = k.tiles_m * k.tiles_n;
    hipLaunchKernelGGL(gemm_pp_kernel<T>, dim3(k.nwg), dim3(512), LDSB, s, k);
    return OFX_OK;
}

template <typename T, int WR, int WC, int NST, int ABL = 0>
static int launch_big(KArgs& k, int M, int N, hipStream_t s) {
    constexpr int TM = WR * 128, TN = WC * 64, NW = WR * WC;
    constexpr int LDSB = NST * (TM + TN) * BK * 2 + NW * EPI2_BYTES_PER_WAVE;
    static bool attr = false;
    if (!attr) {
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_big_kernel<T, WR, WC, NST, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        attr = true;
    }
    k.tiles_n = N / TN; k.tiles_m = (M + TM - 1) / TM; k.nwg = k.tiles_m * k.tiles_n;
    hipLaunchKernelGGL((gemm_big_kernel<T, WR, WC, NST, ABL>), dim3(k.nwg), dim3(64 * NW), LDSB, s, k);
    return OFX_OK;
}

}  // namespace

// Split-K plan for the 128x128 kernel: used when the tile grid leaves most of the chip idle and K is deep.
// Returns the number of K splits (1 = none).  Shared by the launcher and the workspace sizing (api.hip).
int ofx_gemm_splitk_plan(int M, int N, int K) {
    if (g_gemm_splitk == 0 || N % 128 || K % 64) return 1;
    const long blocks = (long)((M + 127) / 128) * (N / 128);
    const int nk = K / 64;
    if (blocks >= 256 || nk < 16) return 1;
    int s = (int)((512 + blocks - 1) / blocks);
    if (s > 8) s = 8;
    if (s > nk / 4) s = nk / 4;                       // at least 4 k-tiles per split
    if (s < 2) return 1;
    const int kps = (nk + s - 1) / s;
    return (nk + kps - 1) / kps;                      // every split owns at least one k-tile
}
size_t ofx_gemm_splitk_bytes(int M, int N, int K) {
    const int s = ofx_gemm_splitk_plan(M, N, K);
    return s > 1 ? (size_t)s * M * N * 4 : 0;
}


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
                       void* slab, size_t slab_bytes, int op_dtype, hipStream_t s) {
    OFX_REQUIRE(M > 0 && N > 0 && K > 0, OFX_ESHAPE, "gemm_tn: empty problem M=%d N=%d K=%d", M, N, K);
    OFX_REQUIRE(M % 256 == 0 && N % 256 == 0, OFX_ESHAPE, "gemm_tn: M=%d and N=%d must be multiples of 256", M, N);
    OFX_REQUIRE(lda >= M && ldb >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc >= N && ldc % 4 == 0, OFX_ESHAPE, "gemm_tn: bad leading dimension");
    OFX_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0), OFX_EINVAL, "gemm_tn: operands must be 16-byte aligned");
    OFX_REQUIRE(op_dtype == OFX_BF16 || op_dtype == OFX_F16, OFX_EINVAL, "gemm_tn: operand dtype must be bf16 or f16");
    TnArgs t;
    t.A = (const char*)A; t.B = (const char*)B; t.C = C; t.slab = (float*)slab; t.k_dev = k_dev;
    t.M = M; t.N = N; t.K = K; t.lda = lda; t.ldb = ldb; t.ldc = ldc;
    t.splits = slab ? ofx_gemm_tn_splits(M, N, K) : 1;
    if (t.splits > 1) OFX_REQUIRE(slab_bytes >= (size_t)t.splits * M * N * 4, OFX_EWORKSPACE, "gemm_tn: split-K slab too small");
    t.tiles_m = M / 256; t.tiles_n = N / 256; t.nwg = t.tiles_m * t.tiles_n * t.splits; t.group_m = 4;
    constexpr int LDSB = 2 * 4 * 64 * 256 + 8 * EPI2_BYTES_PER_WAVE;
    static bool attr = false;
    if (!attr) {
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_tn_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_tn_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        attr = true;
    }
    ProfScope prof(PROF_GEMM, s, 2.0 * M * N * K);
    if (op_dtype == OFX_F16) hipLaunchKernelGGL(gemm_tn_kernel<f16_t>, dim3(t.nwg), dim3(512), LDSB, s, t);
    else hipLaunchKernelGGL(gemm_tn_kernel<bf16_t>, dim3(t.nwg), dim3(512), LDSB, s, t);
    if (t.splits > 1) {
        KArgs k{};
        k.C = (char*)C; k.M = M; k.N = N; k.ldc = ldc; k.out_kind = 0; k.act = OFX_ACT_NONE; k.splits = t.splits; k.slab = (float*)slab; k.m_slab = M;
        size_t tot = (size_t)M * (N / 4);
        int rg = (int)((tot + 255) / 256); if (rg > 2048) rg = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3(rg), dim3(256), 0, s, k);
    }
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}

int g_gemm_group_m = 0;   // 0 = adaptive
int g_gemm_ablate = 0;    // diagnostics only (tools/gemm_bench.py)
unsigned long long* g_gemm_dbg = nullptr;   // diagnostics only
int g_gemm_pref = 2;      // short-K big GEMMs: 0 -> 256x128 kernel, 1 -> 256x256, 2 -> 256x256 ping-pong
int g_gemm_skew = 0;      // start skew of the second co-resident block (x 8128 cycles), 256x128 kernel only
int g_gemm_kernel = 0;    // 0 auto, 1 force 128x128, 2 force 256x256 (8 waves, 2 stages), 3 force 256x128 (4 waves, register-resident k-tile), 4 force 256x256 ping-pong

int ofx_launch_gemm(const GemmArgs& g, int op_dtype, hipStream_t s) {
    OFX_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, OFX_ESHAPE, "gemm: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
    OFX_REQUIRE(g.N % BN == 0, OFX_ESHAPE, "gemm: N=%d must be a multiple of %d (pad the weight at pack time)", g.N, BN);
    OFX_REQUIRE(g.K % BK == 0, OFX_ESHAPE, "gemm: K=%d must be a multiple of %d", g.K, BK);
    OFX_REQUIRE(g.lda >= g.K && g.lda % 8 == 0, OFX_ESHAPE, "gemm: lda=%d must be >= K and a multiple of 8", g.lda);
    OFX_REQUIRE(g.ldc % (g.out_kind == 0 ? 4 : 8) == 0 && g.ldc >= (g.out_kind == 2 ? 3 * g.N : g.N), OFX_ESHAPE, "gemm: bad ldc=%d", g.ldc);
    OFX_REQUIRE(!g.resid || (g.ldr % 4 == 0 && g.ldr >= g.N), OFX_ESHAPE, "gemm: bad ldr=%d", g.ldr);
    OFX_REQUIRE(((uintptr_t)g.A % 16 == 0) && ((uintptr_t)g.W % 16 == 0) && ((uintptr_t)g.C % 16 == 0), OFX_EINVAL,
                "gemm: operands must be 16-byte aligned");
    OFX_REQUIRE(op_dtype == OFX_BF16 || op_dtype == OFX_F16, OFX_EINVAL, "gemm: operand dtype must be bf16 or f16");
#ifndef OFX_DIAG
    OFX_REQUIRE(g_gemm_ablate == 0, OFX_ESTATE, "gemm: the ablation kernels are only built with `make DIAG=1`");
#endif
    KArgs k;
    k.A = (const char*)g.A; k.W = (const char*)g.W; k.C = (char*)g.C; k.bias = g.bias; k.resid = g.resid; k.aux_out = g.aux_out; k.m_dev = g.m_dev; k.dbg = g_gemm_dbg; k.skew = g_gemm_skew; k.splits = 1; k.slab = nullptr; k.m_slab = g.M; k.kt_per_split = 0;
    k.M = g.M; k.N = g.N; k.K = g.K; k.lda = g.lda; k.ldc = g.ldc; k.ldr = g.ldr; k.act = g.act; k.out_kind = g.out_kind; k.drop = g.drop;
    k.xb_out = (char*)g.xb_out; k.stat_part = g.stat_part; k.row_stat = g.row_stat; k.col_sum = g.col_sum;
    OFX_REQUIRE(!(g.xb_out || g.stat_part) || (g.out_kind == 0 && g.N % 64 == 0), OFX_EINVAL, "gemm: LayerNorm-fold producer outputs need an fp32 output");
    OFX_REQUIRE(!g.row_stat || g.col_sum, OFX_EINVAL, "gemm: row_stat needs col_sum");
    OFX_REQUIRE(!(g.row_stat || g.xb_out || g.stat_part) || (!g.aux_out && !g.drop.thresh && g.act != OFX_ACT_MISH && g.act != OFX_ACT_MISH_GRAD), OFX_EINVAL,
                "gemm: LayerNorm folding does not combine with the training epilogue features");
    OFX_REQUIRE(!g.row_stat || !g.resid, OFX_EINVAL, "gemm: a LayerNorm-fold consumer takes no residual");
    OFX_REQUIRE(!(g.xb_out || g.stat_part) || g.act == OFX_ACT_NONE, OFX_EINVAL, "gemm: a LayerNorm-fold producer has no activation");
    static bool attr_set = false;
    if (!attr_set) {
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_128x128_kernel<bf16_t, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_128x128_kernel<f16_t, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
#ifdef OFX_DIAG
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_128x128_kernel<bf16_t, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_128x128_kernel<bf16_t, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_128x128_kernel<bf16_t, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
#endif
        attr_set = true;
    }
    // big tiles when they still fill the chip, else the 128^2 kernel
    int kind = g_gemm_kernel;
    if (kind == 0) {   // measured crossover points (tools/gemm_bench.py, profiles/r01_gemm_variants.txt)
        const long t2 = (long)((g.M + 255) / 256) * (g.N / 256), t3 = (long)((g.M + 255) / 256) * (g.N / 128);
        if (g.N % 256 == 0 && t2 >= 1024 && (g.K > 1024 || g_gemm_pref >= 1)) kind = g_gemm_pref == 2 && g.K <= 1024 ? 4 : 2;   // 256x256, one block per CU
        else if (g.N % 128 == 0 && t3 >= 512) kind = 3;                    // short K / mid-size M: 256x128, two blocks per CU
        else kind = 1;
    }
    if ((kind == 2 || kind == 4) && g.N % 256) kind = 1;
    ProfScope prof(PROF_GEMM, s, 2.0 * g.M * g.N * g.K);
    if (kind == 2 || kind == 3 || kind == 4) {
        k.group_m = g_gemm_group_m > 0 ? g_gemm_group_m : (kind == 3 ? 4 : 8);
        int rc;
        if (kind == 4) rc = op_dtype == OFX_F16 ? launch_pp<f16_t>(k, g.M, g.N, s) : launch_pp<bf16_t>(k, g.M, g.N, s);
#ifdef OFX_DIAG      // ablation variants (wrong results, timing diagnostics): `make DIAG=1`; they double the build time of this file
        else if (kind == 2 && g_gemm_ablate == 1) rc = launch_big<bf16_t, 2, 4, 2, 1>(k, g.M, g.N, s);
        else if (kind == 2 && g_gemm_ablate == 2) rc = launch_big<bf16_t, 2, 4, 2, 2>(k, g.M, g.N, s);
        else if (kind == 2 && g_gemm_ablate == 3) rc = launch_big<bf16_t, 2, 4, 2, 3>(k, g.M, g.N, s);
        else if (kind == 2 && g_gemm_ablate == 4) rc = launch_big<bf16_t, 2, 4, 2, 4>(k, g.M, g.N, s);
        else if (kind == 2 && g_gemm_ablate == 5) rc = launch_big<bf16_t, 2, 4, 2, 5>(k, g.M, g.N, s);
#endif
        else if (kind == 2) rc = op_dtype == OFX_F16 ? launch_big<f16_t, 2, 4, 2>(k, g.M, g.N, s) : launch_big<bf16_t, 2, 4, 2>(k, g.M, g.N, s);
        else rc = op_dtype == OFX_F16 ? launch_big<f16_t, 2, 2, 1>(k, g.M, g.N, s) : launch_big<bf16_t, 2, 2, 1>(k, g.M, g.N, s);
        if (rc != OFX_OK) return rc;
    } else {
        k.tiles_n = g.N / BN; k.tiles_m = (g.M + BM - 1) / BM; k.nwg = k.tiles_m * k.tiles_n;
        // row panels per L2 group: (group_m + 64/group_m) panels of 128 x K operands should fit ~3 MiB of the XCD's L2
        int gm = g_gemm_group_m;
        if (gm <= 0) { gm = (int)((3u << 20) / ((size_t)BM * g.K * 2) / 2); gm = gm < 1 ? 1 : (gm > 8 ? 8 : gm); }
        k.group_m = gm;
        const int splits = (g.slab && !g.xb_out && !g.stat_part && !g.row_stat) ? ofx_gemm_splitk_plan(g.M, g.N, g.K) : 1;
        k.splits = splits; k.slab = (float*)g.slab; k.m_slab = g.M;
        k.kt_per_split = splits > 1 ? (g.K / BK + splits - 1) / splits : 0;
        if (splits > 1) OFX_REQUIRE(g.slab_bytes >= (size_t)splits * g.M * g.N * 4, OFX_EWORKSPACE, "gemm: split-K slab too small");
        const dim3 grid(k.nwg, splits > 1 ? splits : 1);
        if (op_dtype == OFX_F16) hipLaunchKernelGGL((gemm_128x128_kernel<f16_t, 0>), grid, dim3(256), GEMM_LDS_BYTES, s, k);
#ifdef OFX_DIAG
        else if (g_gemm_ablate == 1) hipLaunchKernelGGL((gemm_128x128_kernel<bf16_t, 1>), grid, dim3(256), GEMM_LDS_BYTES, s, k);
        else if (g_gemm_ablate == 2) hipLaunchKernelGGL((gemm_128x128_kernel<bf16_t, 2>), grid, dim3(256), GEMM_LDS_BYTES, s, k);
        else if (g_gemm_ablate == 3) hipLaunchKernelGGL((gemm_128x128_kernel<bf16_t, 3>), grid, dim3(256), GEMM_LDS_BYTES, s, k);
#endif
        else hipLaunchKernelGGL((gemm_128x128_kernel<bf16_t, 0>), grid, dim3(256), GEMM_LDS_BYTES, s, k);
        if (splits > 1) {
            size_t tot = (size_t)g.M * (g.N / 4);
            int rg = (int)((tot + 255) / 256); if (rg > 2048) rg = 2048;
            if (op_dtype == OFX_F16) hipLaunchKernelGGL(splitk_reduce_kernel<f16_t>, dim3(rg), dim3(256), 0, s, k);
            else hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3(rg), dim3(256), 0, s, k);
        }
    }
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
