// Fused QKV projection + scaled-dot-product attention of a CLIP ViT layer (the kernel BASELINE.json's north star names).
//
// Replaces, per layer, the pair [QKV GEMM -> q|k|v in HBM -> attention kernel] (HF CLIPAttention reached from
// src/models/encoders/image_encoders/clip_image_encoder.py:74-76: q/k/v_proj, softmax(q k^T / 8) v; out_proj stays a GEMM).
// q | k | v never touch HBM: 2 x 472 MB per ViT layer at 2048 images.
//
// One block = G images (G = 256 / S rounded down = 5 for S = 50: 250 of the 256 tile rows live) x ONE head:
//   1. [256 x 192] = X_blk [256 x W] . W_h^T, W_h = the head's 64 q, 64 k and 64 v weight rows (K = W = 768, 12 k-tiles of 64):
//      the ping-pong main loop of gemm_pp.hip (LDS-DMA into two 56 KiB stages, XOR-swizzled 128-byte rows, the two wave groups
//      offset by one barrier slot), 8 waves x (128 x 48) accumulators;
//   2. epilogue INTO LDS: (LayerNorm-fold row statistics, column sums,) bias, rounding to the operand type; Q and K as 144-byte
//      rows, V as 160-byte rows (conflict-free for ds_read_b128 / ds_read_b64_tr_b16) over the now idle stages: "LDS-staged K/V";
//   3. waves 0 .. G-1 each run one image's attention exactly as attention_mfma_kernel does (S^T = K Q^T on the matrix core, a
//      query's scores in one lane quad: wavefront softmax; P re-used in place as the operand of O^T = V^T P^T; V by transposed
//      LDS reads) with every fragment read from LDS, and store the image's [S x 64] output slice.
// Block order (bijective XCD remap, then units of 6 image groups x half the heads): the blocks resident on an XCD share six 384 KB
// activation tiles and six heads' weights (4.1 MB, the size of that XCD's L2).
#include "gemm_common.h"

namespace {

struct FusedK {
    const char* X;          // [n_img * S, ldx] operand type: LayerNorm output, or the raw stream copy when row_stat (LayerNorm folding)
    const char* Wqkv;       // [3 Wm, Wm] operand type, rows q | k | v
    char* out;              // [n_img * S, ldo] operand type: attention output (heads concatenated)
    const float* bias;      // [3 Wm]
    const float* row_stat;  // optional [rows, 2] (mean, rstd): LayerNorm-fold consumer
    const float* col_sum;   // [3 Wm] with row_stat
    int n_img, S, Wm, heads, ldx, ldo, G, nwg;
    float scale;
};

constexpr int FQ_TM = 256, FQ_TN = 192, FQ_STAGE = (FQ_TM + FQ_TN) * BK * 2;      // 57344
constexpr int FQ_QK_ROW = 144, FQ_V_ROW = 160;
constexpr int FQ_Q_OFF = 0, FQ_K_OFF = FQ_TM * FQ_QK_ROW, FQ_V_OFF = 2 * FQ_TM * FQ_QK_ROW;      // 0, 36864, 73728; V ends at 114688 = 2 stages
constexpr int FQ_LDS = 2 * FQ_STAGE + 16 * FQ_V_ROW;                               // + 16 zeroed V rows behind the last image (keys S .. 63 of image G-1)
// Dual-weight variant (W2 = true: Wqkv rows are [hi(Wm) | lo(Wm)], the split weights of DESIGN.md section 2): the main loop of
// gemm_w2.hip - BK = 32, three stages [A 256 rows | W_hi 192 rows | W_lo 192 rows] x 64 B, the two wave groups offset by one barrier
// slot, 48 MFMAs (8 A fragments x 3 column tiles x {hi, lo}) per wave and step against 14 fragment reads - then the same epilogue
// and attention.  The q | k | v image (117,248 B incl. the V pad rows) lies inside the three stages (122,880 B).
constexpr int FQ2_BK = 32, FQ2_A = FQ_TM * FQ2_BK * 2, FQ2_W = FQ_TN * FQ2_BK * 2, FQ2_STAGE = FQ2_A + 2 * FQ2_W, FQ2_NST = 3;   // 16384, 12288, 40960
constexpr int FQ2_LDS = FQ2_NST * FQ2_STAGE;
static_assert(FQ2_LDS >= FQ_LDS, "q | k | v image must fit the dual-weight stages");

template <typename T, bool W2 = false>
__global__ __launch_bounds__(512, 2) void fused_qkv_attn_kernel(FusedK p) {
    typedef typename OpT<T>::v8 v8;
    typedef typename OpT<T>::v4 v4;
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
    // L2 working set (4 MiB per XCD, 32 resident blocks): a unit of 36 consecutive blocks = 6 image groups x HALF the heads
    // (6 activation tiles 2.3 MB + 6 heads' weights 1.8 MB), the next unit the same groups x the other half - the activation
    // tiles are fetched once, the weight halves once per 6 groups (head-fastest over all 12 heads needed all 3.5 MB of weights
    // plus the tiles at once and re-fetched the weights per group: 985 MB of fabric traffic per launch against 318 MB algorithmic)
    const int hh = p.heads % 2 == 0 ? p.heads / 2 : p.heads, halves = p.heads / hh, per_unit = 6 * hh;
    const int unit = bid / per_unit, in_unit = bid % per_unit;
    const int grp = (unit / halves) * 6 + in_unit / hh, head = (unit % halves) * hh + in_unit % hh;
    if (grp * p.G >= p.n_img) return;                 // ragged last group batch (block-uniform, before any barrier)
    const int M = p.n_img * p.S;
    const int m0 = grp * p.G * p.S;
    const int n_live_img = min(p.G, p.n_img - grp * p.G);

    f32x4 acc[8][3];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;

    if constexpr (W2) {
    // ------------------------------------------------------------------ dual-weight main loop (see gemm_w2.hip for the slot schedule)
    const int prow = lane >> 2, pchk = (lane & 3) ^ ((4 - (lane >> 4)) & 3);
    const char* a_base = p.X + (size_t)m0 * p.ldx * 2;
    const char* w_base = p.Wqkv + (size_t)head * 64 * (2 * p.Wm) * 2;          // row stride 2 Wm elements: [hi | lo]
    unsigned a_off[2], w_off[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + prow;
        const int rr = m0 + row < M ? row : M - 1 - m0;
        a_off[i] = ((unsigned)rr * p.ldx + pchk * 8) * 2;
    }
    int w_dst[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = wave * 3 + i;                                            // 24 weight pieces: 0..11 hi, 12..23 lo (16 tile rows each)
        const int row = (q % 12) * 16 + prow;                                  // tile row 0..191 = which * 64 + r
        const unsigned grow = (unsigned)(row >> 6) * p.Wm + (row & 63);
        w_off[i] = (grow * (2 * p.Wm) + (q >= 12 ? p.Wm : 0) + pchk * 8) * 2;
        w_dst[i] = FQ2_A + (q >= 12 ? FQ2_W : 0) + (q % 12) * 1024;
    }
    const int a_dst = wave * 2 * 1024;
    const int nk = p.Wm / FQ2_BK;
    auto issue_all = [&](int step) {
        const int sc = step < nk ? step : nk - 1;
        OFX_LDS char* base = lds + (step % FQ2_NST) * FQ2_STAGE;
        const char* ak = a_base + (size_t)sc * FQ2_BK * 2;
        const char* wk = w_base + (size_t)sc * FQ2_BK * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(ak + a_off[i], base + a_dst + i * 1024);
#pragma unroll
        for (int i = 0; i < 3; ++i) glds16(wk + w_off[i], base + w_dst[i]);
    };
    const int fchk = (fq ^ ((4 - (fr >> 2)) & 3)) * 16;
    const int a_frag = (wr * 128 + fr) * 64 + fchk;
    const int w_frag = FQ2_A + (wc * 48 + fr) * 64 + fchk;
    v8 af[8], wh[3], wl[3];
#define FQ2_READ(STG)                                                                                            \
    {                                                                                                            \
        OFX_LDS char* base_ = lds + (STG) * FQ2_STAGE;                                                           \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) wh[j] = *(OFX_LDS v8*)(base_ + w_frag + j * 16 * 64);       \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) af[i] = *(OFX_LDS v8*)(base_ + a_frag + i * 16 * 64);       \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) wl[j] = *(OFX_LDS v8*)(base_ + FQ2_W + w_frag + j * 16 * 64); \
    }
#define FQ2_MFMA()                                                                                               \
    {                                                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                                           \
        _Pragma("unroll") for (int m = 0; m < 48; ++m) {                                                         \
            const int i = m / 6, j = m % 3;                                                                      \
            acc[i][j] = OpT<T>::mfma16((m % 6) >= 3 ? wl[j] : wh[j], af[i], acc[i][j]);                          \
        }                                                                                                        \
        __builtin_amdgcn_s_setprio(0);                                                                           \
    }
    issue_all(0); issue_all(1);
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");            // step 0 landed (my pieces)
    __builtin_amdgcn_s_barrier();
    if (wr == 0) {
        for (int t = 0; t < nk; ++t) {
            issue_all(t + 2);
            FQ2_READ(t % FQ2_NST)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            FQ2_MFMA()
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");    // my pieces of step t+1 landed (step t+2 stays in flight)
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();
    } else {
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nk; ++t) {
            issue_all(t + 2);
            FQ2_READ(t % FQ2_NST)
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            FQ2_MFMA()
            __builtin_amdgcn_s_barrier();
        }
    }
#undef FQ2_READ
#undef FQ2_MFMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // every stage read and every (clamped) fill is done: the stages become q | k | v
    // the V pad rows lie inside stage 2 here: zero them now (the barrier before the attention publishes them)
    if (tid < 16 * FQ_V_ROW / 16) *(OFX_LDS f32x4*)(lds + 2 * FQ_STAGE + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
    // ------------------------------------------------------------------ single-product main loop (ping-pong, BK = 64, two stages)
    // zero the V pad rows once (never touched by the stages)
    if (tid < 16 * FQ_V_ROW / 16) *(OFX_LDS f32x4*)(lds + 2 * FQ_STAGE + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    const int lrow = lane >> 3, lchk = lane & 7;
    const char* a_base = p.X + (size_t)m0 * p.ldx * 2;
    const char* w_base = p.Wqkv + (size_t)head * 64 * p.Wm * 2;
    unsigned a_off[4], w_off[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + lrow;
        const int rr = m0 + row < M ? row : M - 1 - m0;
        a_off[i] = ((unsigned)rr * p.ldx + (lchk ^ ((row >> 1) & 7)) * 8) * 2;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int row = (wave * 3 + i) * 8 + lrow;                         // tile row 0..191 = which * 64 + r
        const unsigned grow = (unsigned)(row >> 6) * p.Wm + (row & 63);    // + head * 64 (in w_base)
        w_off[i] = (grow * p.Wm + (lchk ^ ((row >> 1) & 7)) * 8) * 2;
    }
    const int a_dst = wave * 4 * 1024, w_dst = FQ_TM * BK * 2 + wave * 3 * 1024;
    auto issue_all = [&](int kt, int stage) {
        OFX_LDS char* base = lds + stage * FQ_STAGE;
        const char* ak = a_base + (size_t)kt * BK * 2;
        const char* wk = w_base + (size_t)kt * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(ak + a_off[i], base + a_dst + i * 1024);
#pragma unroll
        for (int i = 0; i < 3; ++i) glds16(wk + w_off[i], base + w_dst + i * 1024);
    };

    const int fsw = fr >> 1;
    const int a_frag = (wr * 128 + fr) * 128;
    const int w_frag = FQ_TM * BK * 2 + (wc * 48 + fr) * 128;
    v8 af[2][8], wf[2][3];

#define FQ_READ_FRAGS(STG)                                                                                    \
    {                                                                                                         \
        OFX_LDS char* base_ = lds + (STG) * FQ_STAGE;                                                         \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                    \
            const int chk = ((ks * 4 + fq) ^ fsw) * 16;                                                       \
            _Pragma("unroll") for (int j = 0; j < 3; ++j) wf[ks][j] = *(OFX_LDS v8*)(base_ + w_frag + j * 16 * 128 + chk); \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) af[ks][i] = *(OFX_LDS v8*)(base_ + a_frag + i * 16 * 128 + chk); \
        }                                                                                                     \
    }

    const int nk = p.Wm / BK;
    issue_all(0, 0);
    issue_all(nk > 1 ? 1 : 0, 1);
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");        // k-tile 0 landed (my pieces)
    __builtin_amdgcn_s_barrier();                           // ---- end of slot 0
    if (wr == 0) {
        for (int t = 0; t < nk; ++t) {
            if (t >= 1 && t + 1 < nk) issue_all(t + 1, (t + 1) & 1);
            FQ_READ_FRAGS(t & 1)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int m = 0; m < 48; ++m) {
                const int ks = m / 24, i = (m % 24) / 3, j = m % 3;
                acc[i][j] = OpT<T>::mfma16(wf[ks][j], af[ks][i], acc[i][j]);
            }
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();
    } else {
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nk; ++t) {
            FQ_READ_FRAGS(t & 1)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            OFX_LDS char* nbase = lds + (t & 1) * FQ_STAGE;
            const int kn = t + 2 < nk ? t + 2 : nk - 1;
            const char* ak = a_base + (size_t)kn * BK * 2;
            const char* wk = w_base + (size_t)kn * BK * 2;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int m = 0; m < 48; ++m) {
                if (m % 6 == 0 && m / 6 < 7) {
                    const int q = m / 6;
                    if (q < 4) glds16(ak + a_off[q], nbase + a_dst + q * 1024);
                    else glds16(wk + w_off[q - 4], nbase + w_dst + (q - 4) * 1024);
                }
                const int ks = m / 24, i = (m % 24) / 3, j = m % 3;
                acc[i][j] = OpT<T>::mfma16(wf[ks][j], af[ks][i], acc[i][j]);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        }
    }
#undef FQ_READ_FRAGS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // every stage read and every (clamped) fill is done: the stages become q | k | v
    }

    // ---- epilogue into LDS: acc[i][j][r] = C[row wr*128 + i*16 + fr][col wc*48 + j*16 + fq*4 + r]
    {
        float mu[8], rs[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mu[i] = 0.f; rs[i] = 1.f;
            if (p.row_stat) {
                const int gm = min(m0 + wr * 128 + i * 16 + fr, M - 1);
                const f32x2 ms = *(const f32x2*)(p.row_stat + 2 * (size_t)gm);
                mu[i] = ms[0]; rs[i] = ms[1];
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = wc * 48 + j * 16 + fq * 4, which = c >> 6, d = c & 63;
            const int gn = which * p.Wm + head * 64 + d;
            const f32x4 b4 = *(const f32x4*)(p.bias + gn);
            f32x4 cs4 = {0.f, 0.f, 0.f, 0.f};
            if (p.row_stat) cs4 = *(const f32x4*)(p.col_sum + gn);
            OFX_LDS char* reg = lds + (which == 0 ? FQ_Q_OFF : which == 1 ? FQ_K_OFF : FQ_V_OFF);
            const int stride = which == 2 ? FQ_V_ROW : FQ_QK_ROW;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = wr * 128 + i * 16 + fr;
                const f32x4 v = (acc[i][j] - cs4 * mu[i]) * rs[i] + b4;
                v4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (T)v[e];
                *(OFX_LDS v4*)(reg + row * stride + d * 2) = o;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- attention: wave w < n_live_img owns image w of the group (rows w*S ..), head `head`
    if (wave >= n_live_img) return;
    constexpr int NT = 4, KS = 2;
    const int S = p.S, r0 = wave * S;
    const int r16 = lane & 15, q4 = lane >> 4;
    OFX_LDS char* ql = lds + FQ_Q_OFF + r0 * FQ_QK_ROW;
    OFX_LDS char* kl = lds + FQ_K_OFF + r0 * FQ_QK_ROW;
    OFX_LDS char* vl = lds + FQ_V_OFF + r0 * FQ_V_ROW;
    v8 kf[NT][2], qf[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int row = 16 * t + r16;
        row = row < S ? row : S - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[t][ks] = *(OFX_LDS v8*)(kl + row * FQ_QK_ROW + (q4 * 8 + ks * 32) * 2);
            qf[t][ks] = *(OFX_LDS v8*)(ql + row * FQ_QK_ROW + (q4 * 8 + ks * 32) * 2);
        }
    }
    float neg[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) neg[t][r] = (16 * t + 4 * q4 + r >= S) ? -INFINITY : 0.f;
    f32x4 st[NT][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            f32x4 c = {neg[t][0], neg[t][1], neg[t][2], neg[t][3]};
            c = OpT<T>::mfma16(kf[t][0], qf[u][0], c);
            st[t][u] = OpT<T>::mfma16(kf[t][1], qf[u][1], c);
        }
    const float sc = p.scale * 1.4426950408889634f;
    v8 pf[NT][KS];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, st[t][u][r]);
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (m == -INFINITY) m = 0.f;
        const float mb = -m * sc;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[t][u][r], sc, mb));
                st[t][u][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[u][ks][j] = (T)(st[2 * ks + (j >> 2)][u][j & 3] * inv);
    }
    f32x4 ot[4][NT];
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) {
        v8 vf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int key0 = 32 * ks + 16 * h2 + 4 * q4;
                OFX_LDS s16x4* ap = (OFX_LDS s16x4*)(vl + (key0 + (r16 >> 2)) * FQ_V_ROW + (16 * (r16 & 3) + 4 * nd) * 2);
                const v4 trv = __builtin_bit_cast(v4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(ap));
#pragma unroll
                for (int e = 0; e < 4; ++e) vf[ks][4 * h2 + e] = trv[e];
            }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) c = OpT<T>::mfma16(vf[ks], pf[u][ks], c);
            ot[nd][u] = c;
        }
    }
    const int row_first = m0 + r0;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int query = 16 * u + r16;
        if (query < S) {
            T* op = (T*)p.out + (size_t)(row_first + query) * p.ldo + head * 64 + 16 * q4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                v8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (T)ot[2 * h + (e >> 2)][u][e & 3];
                *(v8*)(op + 8 * h) = o;
            }
        }
    }
}

}  // namespace

// X [n_img * S, ldx], Wqkv [3 Wm, Wm] (q | k | v rows; w2: [3 Wm, 2 Wm], row = [hi | lo]), out [n_img * S, ldo]; non-causal, no key mask (the ViT's attention)
int ofx_launch_fused_qkv_attn(const void* X, const void* Wqkv, const float* bias, const float* row_stat, const float* col_sum, void* out,
                              int n_img, int S, int Wm, int heads, int ldx, int ldo, float scale, int op_dtype, hipStream_t s, bool w2) {
    OFX_REQUIRE(X && Wqkv && bias && out && n_img > 0, OFX_EINVAL, "fused_qkv_attn: NULL argument");
    OFX_REQUIRE(S >= 33 && S <= 64 && Wm == heads * 64 && Wm % BK == 0 && Wm / BK >= 2, OFX_ESHAPE, "fused_qkv_attn: S=%d must be in [33,64], width %d = heads * 64", S, Wm);
    // the last image of a block reads keys up to tile row (G - 1) S + 63; only 16 zeroed V pad rows sit behind row 255
    OFX_REQUIRE((FQ_TM / S - 1) * S + 64 - FQ_TM <= 16, OFX_ESHAPE, "fused_qkv_attn: S=%d would read %d key rows past the tile's 16 pad rows (supported: 33, 34, 37..41, 43..64)", S,
                (FQ_TM / S - 1) * S + 64 - FQ_TM);
    OFX_REQUIRE(ldx >= Wm && ldx % 8 == 0 && ldo >= Wm && ldo % 8 == 0, OFX_ESHAPE, "fused_qkv_attn: bad strides");
    OFX_REQUIRE(!row_stat || col_sum, OFX_EINVAL, "fused_qkv_attn: row_stat needs col_sum");
    FusedK k;
    k.X = (const char*)X; k.Wqkv = (const char*)Wqkv; k.out = (char*)out; k.bias = bias; k.row_stat = row_stat; k.col_sum = col_sum;
    k.n_img = n_img; k.S = S; k.Wm = Wm; k.heads = heads; k.ldx = ldx; k.ldo = ldo; k.G = FQ_TM / S; k.scale = scale;
    const int groups = (n_img + k.G - 1) / k.G;
    k.nwg = ((groups + 5) / 6) * 6 * heads;            // whole units of 6 groups (blocks of the ragged tail exit at once)
    static DeviceOnce attr;
    TRY(attr.run([]() -> int {
        OFX_HIP(hipFuncSetAttribute((const void*)fused_qkv_attn_kernel<bf16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FQ_LDS));
        OFX_HIP(hipFuncSetAttribute((const void*)fused_qkv_attn_kernel<f16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FQ_LDS));
        OFX_HIP(hipFuncSetAttribute((const void*)fused_qkv_attn_kernel<bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FQ2_LDS));
        OFX_HIP(hipFuncSetAttribute((const void*)fused_qkv_attn_kernel<f16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FQ2_LDS));
        return OFX_OK;
    }));
    // algorithmic bytes: X read once, the (split) weight rows once, the attention output [rows, Wm] written once in the operand type
    if (g_ofx_prof_on) ofx_prof_set_tag(n_img * S, 3 * Wm, Wm, 7, w2 ? 2 : 1, 2.0 * n_img * S * Wm + 2.0 * (w2 ? 2 : 1) * 3.0 * Wm * Wm + 2.0 * n_img * S * Wm);
    ProfScope prof(PROF_GEMM, s, (w2 ? 2.0 : 1.0) * 2.0 * n_img * S * 3.0 * Wm * Wm + 4.0 * n_img * S * S * Wm, true);
    if (w2) {
        if (op_dtype == OFX_F16) OFX_PLAUNCH(true, (fused_qkv_attn_kernel<f16_t, true>), dim3(k.nwg), dim3(512), FQ2_LDS, s, k);
        else OFX_PLAUNCH(true, (fused_qkv_attn_kernel<bf16_t, true>), dim3(k.nwg), dim3(512), FQ2_LDS, s, k);
    } else {
        if (op_dtype == OFX_F16) OFX_PLAUNCH(true, (fused_qkv_attn_kernel<f16_t, false>), dim3(k.nwg), dim3(512), FQ_LDS, s, k);
        else OFX_PLAUNCH(true, (fused_qkv_attn_kernel<bf16_t, false>), dim3(k.nwg), dim3(512), FQ_LDS, s, k);
    }
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
