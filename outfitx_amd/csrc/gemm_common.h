// Shared pieces of the GEMM translation units (gemm.hip: 128x128 kernel + dispatcher; gemm_big.hip: 256x256 / 256x128 tiles;
// gemm_pp.hip: ping-pong 256x256; gemm_tn.hip: weight-gradient TN kernel): kernel arguments, the LDS-DMA helper and the fused
// epilogues.  Everything here has internal linkage (anonymous namespace / templates); the files are split only so that they
// compile in parallel.
#pragma once
#include <type_traits>

#include "ofx_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;        // 32 KiB
constexpr int EPI_STRIDE = 68;                         // floats per staged output row (64 + 4 pad)
constexpr int EPI_BYTES_PER_WAVE = 64 * EPI_STRIDE * 4;
constexpr int GEMM_LDS_BYTES = 4 * EPI_BYTES_PER_WAVE > 2 * STAGE_BYTES ? 4 * EPI_BYTES_PER_WAVE : 2 * STAGE_BYTES;
constexpr int GEMM64_LDS_BYTES = 2 * (64 + BN) * BK * 2;      // 64x128 variant: 48 KiB (stages) > 4 x 8.5 KiB (epilogue staging)

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
    char* xb_out; float* stat_part; const float* row_stat; const float* col_sum; int stat_ld; char* xlo;   // LayerNorm folding (GemmArgs)
    int n_valid;                // columns >= n_valid are computed but not stored (fp32 outputs of the TN kernel; = N elsewhere)
    int ka_tiles;               // A's k-tile index wraps modulo ka_tiles (K-concatenated weights against ONE copy of A: GemmArgs::a_wrap); = K / BK otherwise
    // fp8 weight-correction product (gemm_w2f8.hip): e4m3 copy of the weights' lo half, [N, K] bytes, k-permuted inside 128-blocks,
    // and its per-row E8M0 scale bytes laid out 8 per (128-column block, lane & 15); a8_scale / a8_e8m0: activations are converted as
    // fp8(a / a8_scale) in registers and enter the product with the scale byte a8_e8m0 (= 127 + log2 a8_scale)
    const char* W8 = nullptr; const char* w8_scale = nullptr; float a8_scale = 0.25f; int a8_e8m0 = 125;
    int epi_direct = 0;         // operand-type outputs without residual / tape leave straight from the accumulator layout (epilogue_direct), no LDS round trip
};

__device__ __forceinline__ void glds16(const char* g, OFX_LDS char* l) {
    __builtin_amdgcn_global_load_lds((const OFX_GLB void*)g, (OFX_LDS void*)l, 16, 0, 0);
}


// One wave drains its 64x64 fp32 sub-tile from LDS as whole row segments: 16 lanes x 16 B per row.
template <typename T, int ACT>
__device__ __forceinline__ void epilogue(const KArgs& p, OFX_LDS float* ep, int gm0, int gn0, int lane, int nrows = 64) {
    typedef typename OpT<T>::v4 v4;
    const int col = (lane & 15) * 4;
    const int gn = gn0 + col;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *(const f32x4*)(p.bias + gn);
#pragma unroll 4
    for (int it = 0; it < nrows / 4; ++it) {
        const int row = it * 4 + (lane >> 4);
        const int gm = gm0 + row;
        f32x4 v = *(OFX_LDS f32x4*)(ep + row * EPI_STRIDE + col);
        if (gm < p.M) {
            if (p.row_stat) {
                const float mu = p.row_stat[2 * (size_t)gm * p.stat_ld], rs = p.row_stat[2 * (size_t)gm * p.stat_ld + 1];
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
            if (p.xlo) {            // LayerNorm-fold producer on a (hi, lo) residual stream (see epilogue2, FOLD 3): in-place update
                T* hp = (T*)p.xb_out + (size_t)gm * p.N + gn;
                T* lp = (T*)p.xlo + (size_t)gm * p.N + gn;
                const v4 h0 = *(const v4*)hp, l0 = *(const v4*)lp;
                v4 hb, lb;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += (float)h0[e] + (float)l0[e]; hb[e] = (T)v[e]; lb[e] = (T)(v[e] - (float)hb[e]); }
                *(v4*)hp = hb; *(v4*)lp = lb;
                const float ssum = row16_sum_to_lane15((v[0] + v[1]) + (v[2] + v[3]));
                const float ssq = row16_sum_to_lane15((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
                if ((lane & 15) == 15) *(f32x2*)(p.stat_part + ((size_t)gm * (p.N >> 6) + (gn0 >> 6)) * 2) = f32x2{ssum, ssq};
            } else if (p.out_kind == 0) {
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
// + per-segment statistics), 2 consumer (row statistics + column sums applied to the accumulator), 3 producer whose residual
// stream is the operand-type pair (xb_out = hi, xlo = lo = x - hi), read and rewritten in place - no fp32 copy of the stream.
// NP x 16 rows by 64 columns of the wave's accumulators: acc[i][J0 + j], i < NP passes, j < 4 column fragments (the 128x64 wave tile
// is NP = 8, JW = 4, J0 = 0; a 64x128 wave tile drains as two halves NP = 4, JW = 8, J0 = 0 / 4).
template <typename T, int ACT, int FOLD = 0, int RAMP = 0, int NP = 8, int JW = 4, int J0 = 0>
__device__ __forceinline__ void epilogue2(const KArgs& p, OFX_LDS char* ep, f32x4 (&acc)[NP][JW], int gm0, int gn0, int lane,
                                          OFX_LDS float* st = nullptr) {
    typedef typename OpT<T>::v8 v8;
    const int fr = lane & 15, fq = lane >> 4;
    if (FOLD != 3 && p.out_kind == 0) {
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
                if (has_res && gm < p.M && gn < p.n_valid) dst[it] = *(const f32x4*)(p.resid + (size_t)gm * p.ldr + gn);
            }
        };
        // RAMP: the residual prefetch deepens as accumulator registers retire (16 per pass): passes fetched before pass i is
        // processed = min(8, 3 + 2 i), so by pass 3 every residual row of the tile is in flight; else a fixed DEPTH ahead
        f32x4 resr[RAMP ? NP : 1][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (RAMP) fetch(d, resr[RAMP ? d : 0]);
            else fetch(d, res[d]);
        }
        if (RAMP) fetch(2, resr[RAMP ? 2 : 0]);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (RAMP) {
                if (i >= 1 && 2 * i + 1 < NP) fetch(2 * i + 1, resr[RAMP ? (2 * i + 1) % NP : 0]);
                if (i >= 1 && 2 * i + 2 < NP) fetch(2 * i + 2, resr[RAMP ? (2 * i + 2) % NP : 0]);
            } else if (i + DEPTH < NP) fetch(i + DEPTH, res[(i + DEPTH) % (DEPTH + 1)]);
#pragma unroll
            for (int j = 0; j < 4; ++j) *(OFX_LDS f32x4*)(ep + fr * 256 + (((j * 4 + fq) ^ (fr & 7)) << 4)) = acc[i][J0 + j];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 4 + rsub;
                const int gm = gm0 + i * 16 + row;
                f32x4 v = *(OFX_LDS f32x4*)(ep + row * 256 + ((chunk ^ (row & 7)) << 4));
                if (gm < p.M && gn < p.n_valid) {
                    if (FOLD == 2) {
                        const float mu = p.row_stat[2 * (size_t)gm * p.stat_ld], rs = p.row_stat[2 * (size_t)gm * p.stat_ld + 1];
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
                        for (int e = 0; e < 4; ++e) v[e] *= act_mish_grad((RAMP ? resr[RAMP ? i : 0][it] : res[i % (DEPTH + 1)][it])[e]);
                    } else v += (RAMP ? resr[RAMP ? i : 0][it] : res[i % (DEPTH + 1)][it]);
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
        // FOLD 2 with a staging slot: the wave's 128 (mean, rstd) pairs go through LDS - one 16-byte global load per lane (rows
        // 2 lane, 2 lane + 1) instead of an 8-byte global load per row, lane and pass (the epilogues are bound by their memory
        // instructions)
        const bool st_lds = FOLD == 2 && st != nullptr;
        if (st_lds) {
            const int ra = min(gm0 + 2 * lane, p.M - 1), rb = min(gm0 + 2 * lane + 1, p.M - 1);
            f32x4 pr;
            if (p.stat_ld == 1 && rb == ra + 1) pr = *(const f32x4*)(p.row_stat + 2 * (size_t)ra);
            else {
                const f32x2 a2 = *(const f32x2*)(p.row_stat + 2 * (size_t)ra * p.stat_ld), b2 = *(const f32x2*)(p.row_stat + 2 * (size_t)rb * p.stat_ld);
                pr = f32x4{a2[0], a2[1], b2[0], b2[1]};
            }
            *(OFX_LDS f32x4*)(st + 4 * lane) = pr;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // FOLD 3: the residual stream is the operand-type pair (hi at xb_out == C, lo at xlo), 16 bytes of each per lane and row,
        // fetched two passes ahead and rewritten in place (every element is owned by exactly one lane of one block)
#ifndef OFX_EP3_DEPTH
#define OFX_EP3_DEPTH 2
#endif
        constexpr int D3 = OFX_EP3_DEPTH < NP ? OFX_EP3_DEPTH : NP;        // passes of the (hi, lo) stream in flight ahead of the one being rewritten
        v8 rh[FOLD == 3 ? D3 + 1 : 1][2], rl[FOLD == 3 ? D3 + 1 : 1][2];
        auto fetch_hl = [&](int pass, v8 (&dh)[2], v8 (&dl)[2]) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int gm = min(gm0 + pass * 16 + it * 8 + rsub, p.M - 1);
                dh[it] = *(const v8*)((const T*)p.xb_out + (size_t)gm * p.N + gn);
                dl[it] = *(const v8*)((const T*)p.xlo + (size_t)gm * p.N + gn);
            }
        };
        if (FOLD == 3) {
#pragma unroll
            for (int d = 0; d < D3; ++d) fetch_hl(d, rh[FOLD == 3 ? d : 0], rl[FOLD == 3 ? d : 0]);
        }
        // FOLD 3 statistics: the 8 lanes of a row all hold its (sum, sum of squares) after row8_sum; lane c8 keeps those of pass-half
        // (2 i + it) & 7 == c8, so after the passes every lane owns NP / 4 rows and the wave writes them with NP / 4 store instructions of 64
        // lanes instead of 2 NP of 8 lanes (the epilogue is bound by its vector-memory instruction count)
        constexpr int KS = FOLD == 3 ? (NP + 3) / 4 : 1;
        float keep_s[KS], keep_q[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k) { keep_s[k] = 0.f; keep_q[k] = 0.f; }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (FOLD == 3 && i + D3 < NP) fetch_hl(i + D3, rh[FOLD == 3 ? (i + D3) % (D3 + 1) : 0], rl[FOLD == 3 ? (i + D3) % (D3 + 1) : 0]);
#ifndef OFX_EP3_NOLDS
#define OFX_EP3_NOLDS 0          // experiment build only (WRONG results): the (hi, lo) epilogue without its LDS transposition - same loads, stores and arithmetic
#endif
            if (!(OFX_EP3_NOLDS && FOLD == 3)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) *(OFX_LDS f32x4*)(ep + fr * 256 + (((j * 4 + fq) ^ (fr & 7)) << 4)) = acc[i][J0 + j];
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 8 + rsub;
                const int gm = gm0 + i * 16 + row;
                f32x4 v0, v1;
                if (OFX_EP3_NOLDS && FOLD == 3) { v0 = acc[i][J0 + 2 * it]; v1 = acc[i][J0 + 2 * it + 1]; }
                else {
                    v0 = *(OFX_LDS f32x4*)(ep + row * 256 + (((2 * c8) ^ (row & 7)) << 4));
                    v1 = *(OFX_LDS f32x4*)(ep + row * 256 + (((2 * c8 + 1) ^ (row & 7)) << 4));
                }
                if (gm < p.M) {
                    if (FOLD == 2) {
                        float mu, rs;
                        if (st_lds) { const f32x2 ms = *(OFX_LDS f32x2*)(st + 2 * (i * 16 + row)); mu = ms[0]; rs = ms[1]; }
                        else { mu = p.row_stat[2 * (size_t)gm * p.stat_ld]; rs = p.row_stat[2 * (size_t)gm * p.stat_ld + 1]; }
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
                    if (FOLD == 3) {
                        const v8 h = rh[FOLD == 3 ? i % (D3 + 1) : 0][it], l = rl[FOLD == 3 ? i % (D3 + 1) : 0][it];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] += (float)h[e] + (float)l[e]; v1[e] += (float)h[4 + e] + (float)l[4 + e]; }
                    }
                    v8 hi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { hi[e] = (T)v0[e]; hi[4 + e] = (T)v1[e]; }
                    T* crow = (T*)p.C + (size_t)gm * p.ldc + gn;
                    *(v8*)crow = hi;
                    if (FOLD == 3) {
                        v8 lo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { lo[e] = (T)(v0[e] - (float)hi[e]); lo[4 + e] = (T)(v1[e] - (float)hi[4 + e]); }
                        *(v8*)((T*)p.xlo + (size_t)gm * p.N + gn) = lo;
                        // per-64-column (sum, sum of squares) of the fp32 values: the 8 lanes of this row
                        const float ssum = row8_sum(((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3])));
                        const float ssq = row8_sum(((v0[0] * v0[0] + v0[1] * v0[1]) + (v0[2] * v0[2] + v0[3] * v0[3])) +
                                                   ((v1[0] * v1[0] + v1[1] * v1[1]) + (v1[2] * v1[2] + v1[3] * v1[3])));
                        if (c8 == ((2 * i + it) & 7)) { keep_s[FOLD == 3 ? (2 * i + it) >> 3 : 0] = ssum; keep_q[FOLD == 3 ? (2 * i + it) >> 3 : 0] = ssq; }
                    }
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
        if (FOLD == 3) {
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const int c = k * 8 + c8;                       // the pass-half this lane kept in slot k
                const int gm = gm0 + (c >> 1) * 16 + (c & 1) * 8 + rsub;
                if (c < 2 * NP && gm < p.M) *(f32x2*)(p.stat_part + ((size_t)gm * (p.N >> 6) + (gn0 >> 6)) * 2) = f32x2{keep_s[k], keep_q[k]};
            }
        }
    }
}

// Operand-type (f16 / bf16) output STRAIGHT from the accumulator layout - no LDS round trip (round 4).  epilogue2 above sends every 16-row pass through
// the wave's LDS staging (4 KiB of fp32 written with ds_write_b128 at ~79 B/clk/CU, read back as row segments): 256 KiB in + 256 KiB out per 256x256
// tile and a write -> read -> convert -> store chain per pass; stamped at 10.4-11.8k cycles per tile for a plain f16 output (profiles/r04_w2f8_slots.txt),
// a sixth of a K = 768 tile.  Here a lane keeps what the MFMA left it - row i 16 + (lane & 15), columns j 16 + (lane >> 4) 4 + 0..3 of column block j -,
// applies the LayerNorm-fold terms / bias / activation in that layout, rounds to the operand type, and ONE v_permlane16_swap per register pair trades halves
// with the lane 16 away between two neighbouring column blocks: lanes with (lane >> 4) even end up with 8 consecutive columns of block A, the odd ones with
// 8 consecutive columns of block B (tools/permlane_probe.hip pins the instruction's row mapping).  A store instruction then covers 16 rows x 64 contiguous
// bytes; the two column-block pairs of a 64-column half complete every 128-byte line back to back.  Stores go through a buffer resource sized to the M
// valid rows: rows past M fall outside it and are dropped by the hardware (no per-row predicate, no clamping).  FOLD: 0 bias (+ activation), 2 LayerNorm-fold
// consumer ((acc - colsum mean) rstd + bias', row statistics through the wave's 1 KiB LDS slot `st` as in epilogue2).
// WIDE (ofx_tune(18, 2)): the two registers of a 64-column half trade their upper / lower eight rows through DPP row_ror:8 (lane fr <-> fr ^ 8, 8 moves per pair of
// stores), so that a store instruction covers 8 rows x 128 contiguous bytes - whole cache lines - and leaves as a plain global store predicated on row < M.
// tools/store_shape_probe.hip: with a quarter of the chip storing, 16 rows x 64 B per instruction leave at 33 GB/s per CU whatever the instruction, 8 rows x 128 B
// at > 80 GB/s (HBM-bound at 64 CUs); with all 256 CUs storing at once every shape is HBM-bound (5.1-6.9 TB/s).
template <typename T, int ACT, int FOLD, int NP, int JW, bool WIDE = false>
__device__ __forceinline__ void epilogue_direct(const KArgs& p, f32x4 (&acc)[NP][JW], int gm0, int gn0, int lane, OFX_LDS float* st) {
    static_assert(JW % 4 == 0 && (FOLD == 0 || FOLD == 2), "whole 64-column halves; bias or LayerNorm-fold consumer");
    typedef T t2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int fr = lane & 15, fq = lane >> 4;
    if (FOLD == 2) {            // the wave's (mean, rstd) pairs: one 16-byte load per lane (rows 2 lane, 2 lane + 1) -> LDS slot
        const int ra = min(gm0 + 2 * lane, p.M - 1), rb = min(gm0 + 2 * lane + 1, p.M - 1);
        f32x4 pr;
        if (p.stat_ld == 1 && rb == ra + 1) pr = *(const f32x4*)(p.row_stat + 2 * (size_t)ra);
        else {
            const f32x2 a2 = *(const f32x2*)(p.row_stat + 2 * (size_t)ra * p.stat_ld), b2 = *(const f32x2*)(p.row_stat + 2 * (size_t)rb * p.stat_ld);
            pr = f32x4{a2[0], a2[1], b2[0], b2[1]};
        }
        *(OFX_LDS f32x4*)(st + 4 * lane) = pr;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const size_t cbytes = (size_t)p.M * p.ldc * 2;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)(cbytes < 0x7fffffff ? cbytes : 0x7fffffff), 0x00020000);
    // this lane's bytes of pass 0, column-block pair 0: row gm0 + fr, column gn0 + (fq & 1) 16 + (fq >> 1) 8
    const unsigned vo = ((unsigned)(gm0 + fr) * (unsigned)p.ldc + (unsigned)(gn0 + (fq & 1) * 16 + (fq >> 1) * 8)) * 2u;
    const unsigned pass_bytes = 16u * (unsigned)p.ldc * 2u;
    // WIDE: row gm0 + (fr & 7) (+ 8 for the second store), bytes (fr >> 3) 64 + (fq & 1) 32 + (fq >> 1) 16 of the 128-byte half
    const unsigned wo = ((unsigned)(gm0 + (fr & 7)) * (unsigned)p.ldc + (unsigned)(gn0 + (fr >> 3) * 32 + (fq & 1) * 16 + (fq >> 1) * 8)) * 2u;
#pragma unroll
    for (int h = 0; h < JW / 4; ++h) {          // 64-column halves: the per-column constants of one half (32 registers) at a time
        f32x4 bb[4], cs[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const int gn = gn0 + (h * 4 + jb) * 16 + fq * 4;
            bb[jb] = p.bias ? *(const f32x4*)(p.bias + gn) : f32x4{0.f, 0.f, 0.f, 0.f};
            cs[jb] = FOLD == 2 ? *(const f32x4*)(p.col_sum + gn) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float mu = 0.f, rs = 1.f;
            if (FOLD == 2) { const f32x2 ms = *(OFX_LDS f32x2*)(st + 2 * (i * 16 + fr)); mu = ms[0]; rs = ms[1]; }
            u32x4 d[2];
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                f32x4 va = acc[i][h * 4 + 2 * jp], vb = acc[i][h * 4 + 2 * jp + 1];
                if (FOLD == 2) { va = (va - cs[2 * jp] * mu) * rs + bb[2 * jp]; vb = (vb - cs[2 * jp + 1] * mu) * rs + bb[2 * jp + 1]; }
                else { va += bb[2 * jp]; vb += bb[2 * jp + 1]; }
#pragma unroll
                for (int e = 0; e < 4; ++e) { va[e] = act_apply<T, ACT>(va[e]); vb[e] = act_apply<T, ACT>(vb[e]); }
                const unsigned a_lo = __builtin_bit_cast(unsigned, t2{(T)va[0], (T)va[1]}), a_hi = __builtin_bit_cast(unsigned, t2{(T)va[2], (T)va[3]});
                const unsigned b_lo = __builtin_bit_cast(unsigned, t2{(T)vb[0], (T)vb[1]}), b_hi = __builtin_bit_cast(unsigned, t2{(T)vb[2], (T)vb[3]});
                // first result = [A rows 0 | B rows 0 | A rows 2 | B rows 2] of the 16-lane rows, second = [A 1 | B 1 | A 3 | B 3]: a lane with fq even now holds A's
                // columns fq 4 .. fq 4 + 7 (its own four, then its neighbour's), a lane with fq odd B's columns (fq - 1) 4 .. + 7
                const u32x2 s0 = __builtin_amdgcn_permlane16_swap(a_lo, b_lo, false, false), s1 = __builtin_amdgcn_permlane16_swap(a_hi, b_hi, false, false);
                // (the pass offset rides in the per-lane offset, NOT in the instruction's scalar offset: with a non-zero soffset on this resource the second and
                //  fourth dword of rows 12-15 of every pass but the first came out wrong - tools/dbg_direct.py; everything in voffset is correct)
                if (!WIDE) __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, rc, (int)(vo + (unsigned)i * pass_bytes + (unsigned)(h * 128 + jp * 64)), 0, 0);
                else d[jp] = u32x4{s0[0], s1[0], s0[1], s1[1]};
            }
            if (WIDE) {
                // d[0] = bytes [0, 64) of this lane's row in the half, d[1] = bytes [64, 128).  x: rows 0-7 whole (lanes fr >= 8 take row fr - 8's d[1]),
                // y: rows 8-15 whole (lanes fr < 8 take row fr + 8's d[0]); row_ror:8 = 0x128, bank mask = the four-lane banks written
                u32x4 x = d[0], y = d[1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[e] = (unsigned)__builtin_amdgcn_update_dpp((int)x[e], (int)d[1][e], 0x128, 0xf, 0xc, false);
                    y[e] = (unsigned)__builtin_amdgcn_update_dpp((int)y[e], (int)d[0][e], 0x128, 0xf, 0x3, false);
                }
                const unsigned o = wo + (unsigned)i * pass_bytes + (unsigned)(h * 128);
                if (gm0 + i * 16 + (fr & 7) < p.M) *(u32x4*)((char*)p.C + (size_t)o) = x;
                if (gm0 + i * 16 + 8 + (fr & 7) < p.M) *(u32x4*)((char*)p.C + (size_t)(o + 8u * (unsigned)p.ldc * 2u)) = y;
            }
        }
    }
}

template <typename T, int NP, int JW>
__device__ __forceinline__ bool epilogue_direct_dispatch(const KArgs& p, f32x4 (&acc)[NP][JW], int gm0, int gn0, int lane, OFX_LDS float* st) {
    if (!p.epi_direct || p.out_kind != 1 || p.resid || p.xb_out || p.stat_part || p.aux_out || p.drop.thresh || p.n_valid != p.N) return false;
    if ((size_t)p.M * p.ldc * 2 >= 0x7fffffff) return false;
    const bool wide = p.epi_direct == 2;
#define OFX_EPD(ACT_, FOLD_) { if (wide) epilogue_direct<T, ACT_, FOLD_, NP, JW, true>(p, acc, gm0, gn0, lane, st); else epilogue_direct<T, ACT_, FOLD_, NP, JW, false>(p, acc, gm0, gn0, lane, st); return true; }
    if (p.row_stat) {
        if (!st) return false;
        switch (p.act) {
            case OFX_ACT_QUICK_GELU: OFX_EPD(OFX_ACT_QUICK_GELU, 2)
            case OFX_ACT_GELU: OFX_EPD(OFX_ACT_GELU, 2)
            case OFX_ACT_NONE: OFX_EPD(OFX_ACT_NONE, 2)
            default: return false;
        }
    }
    switch (p.act) {
        case OFX_ACT_QUICK_GELU: OFX_EPD(OFX_ACT_QUICK_GELU, 0)
        case OFX_ACT_GELU: OFX_EPD(OFX_ACT_GELU, 0)
        case OFX_ACT_NONE: OFX_EPD(OFX_ACT_NONE, 0)
        default: return false;
    }
#undef OFX_EPD
}

template <typename T, int NP = 8, int JW = 4, int J0 = 0>
__device__ __forceinline__ void epilogue2_dispatch(const KArgs& p, OFX_LDS char* ep, f32x4 (&acc)[NP][JW], int gm0, int gn0, int lane,
                                                   OFX_LDS float* st = nullptr) {
    if (p.row_stat) {                                  // LayerNorm-fold consumer: towers only (no residual, no dropout, no tape)
        switch (p.act) {
            case OFX_ACT_QUICK_GELU: epilogue2<T, OFX_ACT_QUICK_GELU, 2, 0, NP, JW, J0>(p, ep, acc, gm0, gn0, lane, st); break;
            case OFX_ACT_GELU: epilogue2<T, OFX_ACT_GELU, 2, 0, NP, JW, J0>(p, ep, acc, gm0, gn0, lane, st); break;
            default: epilogue2<T, OFX_ACT_NONE, 2, 0, NP, JW, J0>(p, ep, acc, gm0, gn0, lane, st); break;
        }
        return;
    }
    if (p.xb_out || p.stat_part) {
        if (p.xlo) epilogue2<T, OFX_ACT_NONE, 3, 1, NP, JW, J0>(p, ep, acc, gm0, gn0, lane);
        else epilogue2<T, OFX_ACT_NONE, 1, 1, NP, JW, J0>(p, ep, acc, gm0, gn0, lane);
        return;
    }
    switch (p.act) {
        case OFX_ACT_QUICK_GELU: epilogue2<T, OFX_ACT_QUICK_GELU, 0, 0, NP, JW, J0>(p, ep, acc, gm0, gn0, lane); break;
        case OFX_ACT_GELU: epilogue2<T, OFX_ACT_GELU, 0, 0, NP, JW, J0>(p, ep, acc, gm0, gn0, lane); break;
        case OFX_ACT_MISH: epilogue2<T, OFX_ACT_MISH, 0, 0, NP, JW, J0>(p, ep, acc, gm0, gn0, lane); break;
        case OFX_ACT_MISH_GRAD: epilogue2<T, OFX_ACT_MISH_GRAD, 0, 0, NP, JW, J0>(p, ep, acc, gm0, gn0, lane); break;
        default:      // fp32 residual outputs (out-proj / fc2): ramped residual prefetch, +1.2 % on those GEMMs (tools/gemm_bench.py)
            if (p.resid && p.out_kind == 0) epilogue2<T, OFX_ACT_NONE, 0, 1, NP, JW, J0>(p, ep, acc, gm0, gn0, lane);
            else epilogue2<T, OFX_ACT_NONE, 0, 0, NP, JW, J0>(p, ep, acc, gm0, gn0, lane);
            break;
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
        if (gn >= p.n_valid) continue;
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

}  // namespace

// launchers implemented by the other translation units (KArgs travels as an opaque pointer: each unit sees the same definition)
int ofx_gemm_launch_big(void* kargs, int kind, int ablate, int op_dtype, int M, int N, hipStream_t s);
int ofx_gemm_launch_pp(void* kargs, int op_dtype, int M, int N, hipStream_t s);
int ofx_gemm_launch_w2(void* kargs, int op_dtype, int M, int N, hipStream_t s);
int ofx_gemm_launch_w2f8(void* kargs, int M, int N, hipStream_t s);
int ofx_gemm_launch_x3(void* kargs, int op_dtype, int M, int N, hipStream_t s);
