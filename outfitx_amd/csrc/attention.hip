// Attention kernels of the scoring path (head_dim 64 everywhere).
//
// (1) attention_mfma: one wavefront per (sequence, head), whole sequence (S <= 64) in one tile —
//     CLIP ViT-B/32 (S = 50, 12 heads) and CLIP text (S <= 64, causal ∧ key-padding, 8 heads);
//     replaces HF CLIPAttention's softmax(QK^T/8 + mask)V (reached from clip_image_encoder.py:74-76,
//     clip_text_encoder.py:56-58).  S^T = K·Q^T on v_mfma_f32_16x16x32 so a query's scores sit in
//     one lane quad (wavefront softmax: in-lane + 2 shuffles); the S^T accumulators are re-used
//     in place as the P operand of O^T = V^T·P^T (k-slot permutation, guide §3), V goes through
//     LDS once and is read back transposed with ds_read_b64_tr_b16 (160-B rows: conflict-free).
// (2) set_attention: fp32 VALU attention over an outfit's 1 + n items (S <= 32, 16 heads);
//     replaces nn.MultiheadAttention's SDPA inside nn.TransformerEncoderLayer
//     (src/models/outfit_x.py:137-140,165-168) with -inf on padded keys realised by pad-free
//     compaction.  < 0.3 % of the path's FLOPs, kept exact.
#include "ofx_common.h"

namespace {

struct AttnK {
    const char* qkv;
    char* out;
    const int64_t* key_mask;
    int nseq, S, n_head, ld, ldo, k_off, v_off, mask_ld, causal;
    float scale;
    const int* cu;          // optional varlen mode (outfit sets): sequence b = rows cu[b] .. cu[b+1], at most 16 * NT of them
    int only_row0;          // varlen mode: compute and store query row 0 only (pruned last layer)
    DropArgs drop;          // varlen mode: attention-probability dropout (row = seq * n_head + head, col = query * 32 + key)
    int split3_w;           // > 0: output rows are [hi(W) | lo(W) | hi(W)] (the A operand of a three-product K-concatenated GEMM), W = split3_w
};

constexpr int V_ROW = 160;                 // bytes per V row in LDS (64 x 2 B + 32 pad): tr-read conflict-free

template <typename T, int NT>   // NT = ceil(S / 16) in {1, 2, 4}: number of 16-row query / key tiles
__global__ __launch_bounds__(256, 3) void attention_mfma_kernel(AttnK a) {      // 3 blocks per CU: <= 168 registers per lane
    typedef typename OpT<T>::v8 v8;
    typedef typename OpT<T>::v4 v4;
    constexpr int KS = (NT + 1) / 2, VROWS = 32 * KS, VT = VROWS * V_ROW;
    __shared__ __attribute__((aligned(16))) char smem[4 * VT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int npairs = a.nseq * a.n_head;
    const int pair_raw = blockIdx.x * 4 + wave;
    const bool live = pair_raw < npairs;           // wave-uniform; dead waves recompute the last pair and store nothing
    const int pair = live ? pair_raw : npairs - 1;
    const int seq = pair / a.n_head, head = pair % a.n_head;
    const int row_first = a.cu ? a.cu[seq] : seq * a.S;
    const int S = a.cu ? min(a.cu[seq + 1] - row_first, 16 * NT) : a.S;
    const T* base = (const T*)a.qkv + (size_t)row_first * a.ld + head * 64;
    const int r16 = lane & 15, q4 = lane >> 4;

    // ---- V -> LDS, zero rows beyond S (0 * garbage must stay 0)
    OFX_LDS char* vl = (OFX_LDS char*)smem + wave * VT;
#pragma unroll
    for (int it = 0; it < VROWS / 8; ++it) {
        const int key = it * 8 + (lane >> 3);
        v8 val;
#pragma unroll
        for (int e = 0; e < 8; ++e) val[e] = (T)0.0f;
        if (key < S) val = *(const v8*)(base + (size_t)key * a.ld + a.v_off + (lane & 7) * 8);
        *(OFX_LDS v8*)(vl + key * V_ROW + (lane & 7) * 16) = val;
    }

    // ---- K (A operand) and Q (B operand) fragments straight from global, rows clamped
    v8 kf[NT][2], qf[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int row = 16 * t + r16;
        row = row < S ? row : S - 1;
        const T* rp = base + (size_t)row * a.ld + q4 * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[t][ks] = *(const v8*)(rp + a.k_off + ks * 32);
            qf[t][ks] = *(const v8*)(rp + ks * 32);
        }
    }

    // ---- additive key mask for this lane's 16 keys (key = 16t + 4q4 + r), independent of the query: 0 or -inf
    float neg[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * t + 4 * q4 + r;
            bool dead = key >= S;
            if (!dead && a.key_mask) dead = a.key_mask[(size_t)seq * a.mask_ld + key] == 0;
            neg[t][r] = dead ? -INFINITY : 0.f;
        }

    // only_row0: query row 0 alone is wanted (pruned last layers: the CLS row / the set's prefix row) -> only query tile 0 is computed
    const int nu = a.only_row0 ? 1 : NT;
    // ---- S^T[key][query]: st[t][u][r] = score(query 16u + r16, key 16t + 4q4 + r)
    f32x4 st[NT][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            if (u >= nu) continue;
            f32x4 c = {neg[t][0], neg[t][1], neg[t][2], neg[t][3]};      // the mask rides in as the accumulator: -inf + finite = -inf
            c = OpT<T>::mfma16(kf[t][0], qf[u][0], c);
            st[t][u] = OpT<T>::mfma16(kf[t][1], qf[u][1], c);
        }
    if (a.causal) {                  // wave-uniform: the ViT never enters
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * t + 4 * q4 + r > 16 * u + r16) st[t][u][r] = -INFINITY;
    }

    // ---- wavefront softmax per query column on the raw scores: p = exp2(s * c - max(s) * c), c = scale * log2 e > 0
    // (one max, then one fma + v_exp_f32 + add per element); P written back normalised
    const float sc = a.scale * 1.4426950408889634f;
    v8 pf[NT][KS];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        if (u >= nu) continue;
        const int query = 16 * u + r16;
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
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[t][u][r], sc, mb));      // arguments <= 0: no range fix-up needed
                st[t][u][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = __builtin_amdgcn_rcpf(sum);
        if (a.drop.thresh) {            // dropout on the probabilities (training): same (row, col) counters as the fp32 set kernel
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[t][u][r] *= drop_mul(a.drop, pair, query * 32 + 16 * t + 4 * q4 + r);
        }
        // P fragment of k-step ks: element j <-> key 16(2ks + (j>>2)) + 4q4 + (j&3)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[u][ks][j] = 2 * ks + (j >> 2) < NT ? (T)(st[(2 * ks + (j >> 2)) % NT][u][j & 3] * inv) : (T)0.0f;
    }

    // ---- O^T[d][query] = V^T · P^T; V fragments by transposed LDS reads (EXEC is all ones here)
    f32x4 ot[4][NT];
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) {
        v8 vf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int key0 = 32 * ks + 16 * h2 + 4 * q4;
                // d <-> MFMA row permutation: row m = r16 of tile nd carries d = 16 (m >> 2) + 4 nd + (m & 3), so after the MFMA a
                // lane's 16 outputs of one query are 16 CONSECUTIVE columns (one 32-byte run per lane, 128 B per lane quad)
                OFX_LDS s16x4* ap = (OFX_LDS s16x4*)(vl + (key0 + (r16 >> 2)) * V_ROW + (16 * (r16 & 3) + 4 * nd) * 2);
                const s16x4 tr = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ap);
                const v4 trv = __builtin_bit_cast(v4, tr);
#pragma unroll
                for (int e = 0; e < 4; ++e) vf[ks][4 * h2 + e] = trv[e];
            }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            if (u >= nu) continue;
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) c = OpT<T>::mfma16(vf[ks], pf[u][ks], c);
            ot[nd][u] = c;
        }
    }

    // ---- store: ot[nd][u][r] = O[query 16u + r16][d = 16 q4 + 4 nd + r]: two 16-byte stores per lane and query
    if (live) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            if (u >= nu) continue;
            const int query = 16 * u + r16;
            if (query < (a.only_row0 ? 1 : S)) {
                T* op = (T*)a.out + (size_t)(row_first + query) * a.ldo + head * 64 + 16 * q4;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (T)ot[2 * h + (e >> 2)][u][e & 3];
                    *(v8*)(op + 8 * h) = o;
                    if (a.split3_w) {                                   // wave-uniform
                        v8 lo;
#pragma unroll
                        for (int e = 0; e < 8; ++e) lo[e] = (T)(ot[2 * h + (e >> 2)][u][e & 3] - (float)o[e]);
                        *(v8*)(op + a.split3_w + 8 * h) = lo;
                        *(v8*)(op + 2 * a.split3_w + 8 * h) = o;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct SetK {
    const void* qkv;       // fp32, or the operand type when qkv_op
    char* out;
    const int* cu;
    int nseq, n_head, D, ldo, out_kind, only_row0;
    float scale;
    DropArgs drop;
    // fixed-length mode (three-product CLIP text tower): cu == nullptr, sequence b = rows b * fixed_S .. + fixed_S, with HF's
    // causal AND key-padding mask (key_mask [nseq, mask_ld] int64, 0 = ignored)
    int fixed_S, causal, mask_ld;
    const int64_t* key_mask;
    // fp32 q | k | v still in split-K slabs: element = sum_s qkv[s * plane + ...] + bias[col] (the GEMM's second pass, done here)
    int splits; size_t plane; const float* bias;
};

template <typename T, int SMAX, typename TI = float>      // TI: element type of qkv (float: scoring path; T: training tape)
__global__ __launch_bounds__(64) void set_attention_kernel(SetK a) {
    constexpr int STR = 68;                         // floats per staged row: 16-B slots of consecutive rows differ
    __shared__ __attribute__((aligned(16))) float qs[SMAX * STR];
    __shared__ __attribute__((aligned(16))) float ks[SMAX * STR];
    __shared__ __attribute__((aligned(16))) float sc[SMAX * (SMAX + 4)];
    const int lane = threadIdx.x;
    const int b = blockIdx.x / a.n_head, h = blockIdx.x % a.n_head;
    const int r0 = a.cu ? a.cu[b] : b * a.fixed_S;
    int S = a.cu ? a.cu[b + 1] - r0 : a.fixed_S;
    S = S < SMAX ? S : SMAX;
    const int nq = a.only_row0 ? 1 : S;
    const int D = a.D;
    float vreg[SMAX];
    if (a.splits > 1) {
        // q | k | v still in the QKV GEMM's split-K slabs: sum them here in the reduce kernel's order (slab 0 + 1 + ... + bias), 16 lanes x
        // float4 per 64-column row slice, four rows per pass; v goes through LDS to reach its lane = column layout
        constexpr int VS = SMAX <= 32 ? SMAX : 1;        // the launcher keeps the slab path to sets of <= 32 rows
        __shared__ __attribute__((aligned(16))) float vs[VS * STR];
        const int c4 = (lane & 15) * 4, rsub = lane >> 4;
        const float* bp = a.bias + h * 64 + c4;
        const f32x4 bq = *(const f32x4*)bp, bk = *(const f32x4*)(bp + D), bv = *(const f32x4*)(bp + 2 * D);
        for (int j0 = 0; j0 < S; j0 += 4) {
            const int j = j0 + rsub;
            if (j < S) {
                const float* rp = (const float*)a.qkv + (size_t)(r0 + j) * 3 * D + h * 64 + c4;
                f32x4 q = {0.f, 0.f, 0.f, 0.f}, kk = *(const f32x4*)(rp + D), vv = *(const f32x4*)(rp + 2 * D);
                if (j < nq) q = *(const f32x4*)rp;
                for (int sl = 1; sl < a.splits; ++sl) {
                    const float* sp = rp + sl * a.plane;
                    if (j < nq) q += *(const f32x4*)sp;
                    kk += *(const f32x4*)(sp + D); vv += *(const f32x4*)(sp + 2 * D);
                }
                if (j < nq) *(f32x4*)(qs + j * STR + c4) = q + bq;
                *(f32x4*)(ks + j * STR + c4) = kk + bk;
                *(f32x4*)(vs + (j % VS) * STR + c4) = vv + bv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SMAX; ++j) vreg[j] = j < S ? vs[(j % VS) * STR + lane] : 0.f;
    } else if (sizeof(TI) == 4) {
        // fp32 q | k | v: q and k rows as 16-byte loads (16 lanes per 64-column slice, four rows per instruction) straight into their LDS
        // rows; v stays one dword per lane and row (its lane = column layout is what the P.V loop wants).  The kernel is bound by its
        // vector-memory instruction count, not by bytes (S = 8: 12 loads + 3 stores per wave instead of 24 + 24)
        const int c4 = (lane & 15) * 4, rsub = lane >> 4;
        for (int j0 = 0; j0 < S; j0 += 4) {
            const int j = j0 + rsub;
            if (j < S) {
                const float* rp = (const float*)a.qkv + (size_t)(r0 + j) * 3 * D + h * 64 + c4;
                if (j < nq) *(f32x4*)(qs + j * STR + c4) = *(const f32x4*)rp;
                *(f32x4*)(ks + j * STR + c4) = *(const f32x4*)(rp + D);
            }
        }
#pragma unroll
        for (int j = 0; j < SMAX; ++j) vreg[j] = j < S ? ((const float*)a.qkv)[(size_t)(r0 + j) * 3 * D + 2 * D + h * 64 + lane] : 0.f;
    } else {
#pragma unroll
        for (int j = 0; j < SMAX; ++j) {
            vreg[j] = 0.f;
            if (j < S) {
                const TI* rp = (const TI*)a.qkv + (size_t)(r0 + j) * 3 * D + h * 64 + lane;
                if (j < nq) qs[j * STR + lane] = (float)rp[0];
                ks[j * STR + lane] = (float)rp[D];
                vreg[j] = (float)rp[2 * D];
            }
        }
    }
    __syncthreads();
    for (int p = lane; p < nq * S; p += 64) {
        const int i = p / S, j = p % S;
        const f32x4* qp = (const f32x4*)(qs + i * STR);
        const f32x4* kp = (const f32x4*)(ks + j * STR);
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const f32x4 x = qp[c], y = kp[c];
            d += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
        }
        bool dead = a.causal && j > i;
        if (!dead && a.key_mask) dead = a.key_mask[(size_t)b * a.mask_ld + j] == 0;
        sc[i * (SMAX + 4) + j] = dead ? -INFINITY : d * a.scale;
    }
    __syncthreads();
    if (lane < nq) {
        float* row = sc + lane * (SMAX + 4);
        float m = -INFINITY;
        for (int j = 0; j < S; ++j) m = fmaxf(m, row[j]);
        if (m == -INFINITY) m = 0.f;                  // a fully masked query row: probabilities 0 (as the MFMA kernel)
        float sum = 0.f;
        for (int j = 0; j < S; ++j) { const float e = expf(row[j] - m); row[j] = e; sum += e; }
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
        if (a.drop.thresh) for (int j = 0; j < S; ++j) row[j] *= inv * drop_mul(a.drop, blockIdx.x, lane * 32 + j);
        else for (int j = 0; j < S; ++j) row[j] *= inv;
    }
    __syncthreads();
    if (a.out_kind == 0) {
        for (int i = 0; i < nq; ++i) {
            const float* row = sc + i * (SMAX + 4);
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < SMAX; ++j) acc += (j < S ? row[j] : 0.f) * vreg[j];
            ((float*)a.out)[(size_t)(r0 + i) * a.ldo + h * 64 + lane] = acc;
        }
        return;
    }
    // operand-type outputs: the fp32 rows go through the (now free) q staging rows and leave as 16-byte stores - 8 lanes per 64-column
    // slice, 8 rows per instruction, one instruction per copy (hi | lo | hi) instead of one 2-byte store per lane, row and copy
    typedef typename OpT<T>::v8 v8;
    for (int i = 0; i < nq; ++i) {
        const float* row = sc + i * (SMAX + 4);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < SMAX; ++j) acc += (j < S ? row[j] : 0.f) * vreg[j];
        qs[i * STR + lane] = acc;
    }
    __syncthreads();
    const int c8 = (lane & 7) * 8, r8 = lane >> 3;
    for (int i0 = 0; i0 < nq; i0 += 8) {
        const int i = i0 + r8;
        if (i < nq) {
            const f32x4 x0 = *(const f32x4*)(qs + i * STR + c8), x1 = *(const f32x4*)(qs + i * STR + c8 + 4);
            v8 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hi[e] = (T)x0[e]; hi[4 + e] = (T)x1[e];
                lo[e] = (T)(x0[e] - (float)hi[e]); lo[4 + e] = (T)(x1[e] - (float)hi[4 + e]);
            }
            T* p = (T*)a.out + (size_t)(r0 + i) * a.ldo + h * 64 + c8;
            *(v8*)p = hi;
            if (a.out_kind == 2) { *(v8*)(p + D) = lo; *(v8*)(p + 2 * D) = hi; }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Backward of the varlen set attention on MFMA (training step).  One wave per (outfit, head), S <= 16 * NT rows.
// Both orientations of the score tile are formed on the matrix core - T: [key][query] (as the forward), N: [query][key] -
// because each of the three output products contracts over the ROW index of a C-layout tile (guide: "an accumulator tile as
// the next MFMA's operand"):
//   dQ^T[d][q] = sum_k  K^T[d][k]  dS^T[k][q]      (B operand = dS in T orientation,   A = K^T by transposed LDS reads)
//   dK^T[d][k] = sum_q  Q^T[d][q]  dS  [q][k]      (B operand = dS in N orientation,   A = Q^T)
//   dV^T[d][k] = sum_q dO^T[d][q]  P~  [q][k]      (B operand = dropped-out P in N,    A = dO^T)
// with P = softmax(q k^T * scale), P~ = P . m (dropout mask), dP = (dO V^T) . m, dS = P . (dP - rowsum(P . dP)) * scale.
// Per-query softmax statistics are computed once in the T orientation (in-lane + 2 shuffles) and permuted to the lanes that
// own those queries in the N orientation.
struct AttnBwdK {
    const char* qkv;        // [rows, 3D] operand type (tape)
    const float* d_o;       // [rows, D] fp32, or [nseq, D] when only_row0
    char* dqkv;             // [rows, 3D] operand type
    const int* cu;
    int nseq, n_head, D, only_row0;
    float scale;
    DropArgs drop;
};

template <typename T, int NT>
__global__ __launch_bounds__(256) void set_attention_bwd_mfma_kernel(AttnBwdK a) {
    typedef typename OpT<T>::v8 v8;
    typedef typename OpT<T>::v4 v4;
    constexpr int KS = (NT + 1) / 2, VROWS = 32 * KS, VT = VROWS * V_ROW;
    __shared__ __attribute__((aligned(16))) char smem[4 * 3 * VT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int npairs = a.nseq * a.n_head;
    const int pair_raw = blockIdx.x * 4 + wave;
    const bool live = pair_raw < npairs;
    const int pair = live ? pair_raw : npairs - 1;
    const int seq = pair / a.n_head, head = pair % a.n_head;
    const int D = a.D, ld = 3 * D;
    const int row_first = a.cu[seq];
    const int S = min(a.cu[seq + 1] - row_first, 16 * NT);
    const T* base = (const T*)a.qkv + (size_t)row_first * ld + head * 64;
    const int r16 = lane & 15, q4 = lane >> 4;
    const int n_go = a.only_row0 ? 1 : S;                         // queries that carry an upstream gradient

    // ---- K, Q and dO (converted) -> LDS row images for the transposed reads; rows >= S (dO: >= n_go) are zero
    OFX_LDS char* kl = (OFX_LDS char*)smem + wave * 3 * VT;
    OFX_LDS char* ql = kl + VT;
    OFX_LDS char* gl = ql + VT;
#pragma unroll
    for (int it = 0; it < VROWS / 8; ++it) {
        const int row = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
        v8 kv, qv, gv;
#pragma unroll
        for (int e = 0; e < 8; ++e) { kv[e] = (T)0.0f; qv[e] = (T)0.0f; gv[e] = (T)0.0f; }
        if (row < S) {
            kv = *(const v8*)(base + (size_t)row * ld + D + c8);
            qv = *(const v8*)(base + (size_t)row * ld + c8);
        }
        if (row < n_go) {
            const float* gp = a.d_o + (size_t)(a.only_row0 ? seq : row_first + row) * D + head * 64 + c8;
            const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { gv[e] = (T)g0[e]; gv[4 + e] = (T)g1[e]; }
        }
        *(OFX_LDS v8*)(kl + row * V_ROW + (lane & 7) * 16) = kv;
        *(OFX_LDS v8*)(ql + row * V_ROW + (lane & 7) * 16) = qv;
        *(OFX_LDS v8*)(gl + row * V_ROW + (lane & 7) * 16) = gv;
    }

    // ---- row fragments (A or B operand with k = feature): K, Q, V from global (rows clamped), dO converted, zero for dead queries
    v8 kf[NT][2], qf[NT][2], vf[NT][2], gf[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int rowu = 16 * t + r16;
        const int row = rowu < S ? rowu : S - 1;
        const T* rp = base + (size_t)row * ld + q4 * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[t][ks] = *(const v8*)(rp + ks * 32);
            kf[t][ks] = *(const v8*)(rp + D + ks * 32);
            vf[t][ks] = *(const v8*)(rp + 2 * D + ks * 32);
            v8 g;
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = (T)0.0f;
            if (rowu < n_go) {
                const float* gp = a.d_o + (size_t)(a.only_row0 ? seq : row_first + rowu) * D + head * 64 + q4 * 8 + ks * 32;
                const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { g[e] = (T)g0[e]; g[4 + e] = (T)g1[e]; }
            }
            gf[t][ks] = g;
        }
    }

    // ---- scores and dP in both orientations
    f32x4 sT[NT][NT], sN[NT][NT], pT[NT][NT], pN[NT][NT];       // sT[t][u]: row = key 16t+4q4+r, col = query 16u+r16; sN[u][t]: row = query, col = key
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            sT[t][u] = OpT<T>::mfma16(kf[t][1], qf[u][1], OpT<T>::mfma16(kf[t][0], qf[u][0], z));
            sN[u][t] = OpT<T>::mfma16(qf[u][1], kf[t][1], OpT<T>::mfma16(qf[u][0], kf[t][0], z));
            pT[t][u] = OpT<T>::mfma16(vf[t][1], gf[u][1], OpT<T>::mfma16(vf[t][0], gf[u][0], z));     // dP^T[key][query] = V . dO^T
            pN[u][t] = OpT<T>::mfma16(gf[u][1], vf[t][1], OpT<T>::mfma16(gf[u][0], vf[t][0], z));     // dP[query][key]
        }

    // ---- T orientation: softmax per query column, dS^T; per-query statistics kept for the N orientation
    const float sc = a.scale * 1.4426950408889634f;
    float q_max[NT], q_inv[NT], q_dot[NT];                          // of query 16u + r16 (same value in the 4 lanes q4 = 0..3)
    v8 dsT[NT][KS];                                                 // B operand (k = key, col = query) of dQ^T
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int query = 16 * u + r16;
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * t + 4 * q4 + r;
                const float s = key < S ? sT[t][u][r] * sc : -INFINITY;
                sT[t][u][r] = s;
                m = fmaxf(m, s);
            }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = exp2f(sT[t][u][r] - m); sT[t][u][r] = e; sum += e; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float mk = a.drop.thresh ? drop_mul(a.drop, pair, query * 32 + 16 * t + 4 * q4 + r) : 1.0f;
                const float p = sT[t][u][r] * inv;
                const float dp = pT[t][u][r] * mk;
                sT[t][u][r] = p; pT[t][u][r] = dp;
                dot += p * dp;
            }
        dot += __shfl_xor(dot, 16, 64);
        dot += __shfl_xor(dot, 32, 64);
        q_max[u] = m; q_inv[u] = inv; q_dot[u] = dot;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = 2 * ks + (j >> 2);
                dsT[u][ks][j] = t < NT ? (T)(sT[t % NT][u][j & 3] * (pT[t % NT][u][j & 3] - dot) * a.scale) : (T)0.0f;
            }
    }

    // ---- N orientation: P~ and dS with the statistics of query 16u + 4q4 + r (they live in the lanes whose r16 equals 4q4 + r)
    v8 dsN[NT][KS], pdN[NT][KS];                                   // B operands (k = query, col = key) of dK^T and dV^T
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        float mq[4], iq[4], dq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int src = 4 * q4 + r;                              // any lane with r16 == 4q4 + r holds that query's statistics
            mq[r] = __shfl(q_max[u], src, 64); iq[r] = __shfl(q_inv[u], src, 64); dq[r] = __shfl(q_dot[u], src, 64);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int query = 16 * u + 4 * q4 + r, key = 16 * t + r16;
                const float mk = a.drop.thresh ? drop_mul(a.drop, pair, query * 32 + key) : 1.0f;
                const float p = key < S ? exp2f(sN[u][t][r] * sc - mq[r]) * iq[r] : 0.f;
                const float dp = pN[u][t][r] * mk;
                sN[u][t][r] = p * mk;                                // dropped-out P
                pN[u][t][r] = p * (dp - dq[r]) * a.scale;            // dS
            }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int u = 2 * ks + (j >> 2);
                dsN[t][ks][j] = u < NT ? (T)pN[u % NT][t][j & 3] : (T)0.0f;
                pdN[t][ks][j] = u < NT ? (T)sN[u % NT][t][j & 3] : (T)0.0f;
            }

    // ---- the three products; A operands by transposed LDS reads (EXEC all ones), stores as operand-type rows
    T* obase = (T*)a.dqkv + (size_t)row_first * ld + head * 64;
    auto product = [&](OFX_LDS char* img, v8 (&bop)[NT][KS], int col_off, int n_rows_out) {
#pragma unroll
        for (int nd = 0; nd < 4; ++nd) {
            v8 af[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int row0 = 32 * ks + 16 * h2 + 4 * q4;
                    OFX_LDS s16x4* ap = (OFX_LDS s16x4*)(img + (row0 + (r16 >> 2)) * V_ROW + (16 * nd + 4 * (r16 & 3)) * 2);
                    const v4 trv = __builtin_bit_cast(v4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(ap));
#pragma unroll
                    for (int e = 0; e < 4; ++e) af[ks][4 * h2 + e] = trv[e];
                }
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) c = OpT<T>::mfma16(af[ks], bop[u][ks], c);
                const int orow = 16 * u + r16;                       // c[r] = out[orow][16nd + 4q4 + r]
                if (live && orow < n_rows_out) {
                    v4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (T)c[r];
                    *(v4*)(obase + (size_t)orow * ld + col_off + 16 * nd + 4 * q4) = o;
                }
            }
        }
    };
    product(kl, dsT, 0, S);            // dQ: contraction over keys   (A = K^T)
    product(ql, dsN, D, S);            // dK: contraction over queries (A = Q^T)
    product(gl, pdN, 2 * D, S);        // dV: contraction over queries (A = dO^T)
}

}  // namespace

int ofx_launch_attention_mfma(const AttnArgs& g, int op_dtype, hipStream_t s) {
    OFX_REQUIRE(g.seq_len >= 1 && g.seq_len <= 64, OFX_ESHAPE, "attention: seq_len=%d must be in [1,64]", g.seq_len);
    OFX_REQUIRE(g.nseq > 0 && g.n_head > 0, OFX_ESHAPE, "attention: nseq=%d n_head=%d", g.nseq, g.n_head);
    OFX_REQUIRE(g.ld % 8 == 0 && g.ldo % 4 == 0 && g.k_off % 8 == 0 && g.v_off % 8 == 0, OFX_ESHAPE, "attention: strides must keep 16-byte alignment");
    OFX_REQUIRE((uintptr_t)g.qkv % 16 == 0 && (uintptr_t)g.out % 8 == 0, OFX_EINVAL, "attention: misaligned pointers");
    AttnK k;
    k.qkv = (const char*)g.qkv; k.out = (char*)g.out; k.key_mask = g.key_mask; k.nseq = g.nseq; k.S = g.seq_len;
    k.n_head = g.n_head; k.ld = g.ld; k.ldo = g.ldo; k.k_off = g.k_off; k.v_off = g.v_off; k.mask_ld = g.mask_ld;
    k.causal = g.causal; k.scale = g.scale; k.cu = g.cu_seqlens; k.only_row0 = g.only_row0; k.drop = g.drop; k.split3_w = g.split3_w;
    OFX_REQUIRE(g.split3_w == 0 || (g.split3_w % 8 == 0 && g.ldo >= 3 * g.split3_w), OFX_ESHAPE, "attention: split3 output needs ldo >= 3 W");
    const int grid = (g.nseq * g.n_head + 3) / 4;
    ProfScope prof(PROF_ATTN, s);
#define AT(T, N) hipLaunchKernelGGL((attention_mfma_kernel<T, N>), dim3(grid), dim3(256), 0, s, k)
    const int nt = g.seq_len <= 16 ? 1 : g.seq_len <= 32 ? 2 : 4;
    if (op_dtype == OFX_F16) { if (nt == 1) AT(f16_t, 1); else if (nt == 2) AT(f16_t, 2); else AT(f16_t, 4); }
    else { if (nt == 1) AT(bf16_t, 1); else if (nt == 2) AT(bf16_t, 2); else AT(bf16_t, 4); }
#undef AT
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}

int ofx_launch_set_attention(const SetAttnArgs& g, int op_dtype, hipStream_t s) {
    OFX_REQUIRE(g.D == g.n_head * 64, OFX_ESHAPE, "set_attention: head_dim must be 64 (D=%d heads=%d)", g.D, g.n_head);
    // 64 rows per sequence in both modes; the training features of varlen sets (hashed dropout keyed query * 32 + key, the split-K slab
    // input, the backward kernel) stay at 32
    OFX_REQUIRE(g.max_len >= 1 && g.max_len <= 64, OFX_ESHAPE, "set_attention: %d rows per sequence exceed 64", g.max_len);
    OFX_REQUIRE(g.max_len <= 32 || !g.drop.thresh, OFX_ESHAPE, "set_attention: dropout columns are keyed query * 32 + key");
    OFX_REQUIRE(g.ldo >= (g.out_kind == 2 ? 3 * g.D : g.D) && (g.out_kind == 0 || g.ldo % 8 == 0), OFX_ESHAPE, "set_attention: bad ldo=%d (operand-type rows leave as 16-byte stores)", g.ldo);
    OFX_REQUIRE(g.cu_seqlens || (g.fixed_len >= 1 && g.fixed_len <= g.max_len), OFX_EINVAL, "set_attention: needs cu_seqlens or a fixed length <= max_len");
    SetK k;
    k.qkv = g.qkv; k.out = (char*)g.out; k.cu = g.cu_seqlens; k.nseq = g.nseq; k.n_head = g.n_head; k.D = g.D; k.ldo = g.ldo;
    k.out_kind = g.out_kind; k.only_row0 = g.only_row0; k.scale = g.scale; k.drop = g.drop;
    k.fixed_S = g.fixed_len; k.causal = g.causal; k.mask_ld = g.mask_ld; k.key_mask = g.key_mask;
    OFX_REQUIRE(g.splits <= 1 || (!g.qkv_op && g.bias && g.plane > 0 && g.max_len <= 32), OFX_EINVAL, "set_attention: slab input needs fp32 q|k|v, the bias, the slab stride and sets of <= 32 rows");
    k.splits = g.splits > 1 ? g.splits : 1; k.plane = g.plane; k.bias = g.bias;
    const int grid = g.nseq * g.n_head;
    ProfScope prof(PROF_ATTN, s);
#define SA(T, N) do { if (g.qkv_op) hipLaunchKernelGGL((set_attention_kernel<T, N, T>), dim3(grid), dim3(64), 0, s, k); \
                      else hipLaunchKernelGGL((set_attention_kernel<T, N, float>), dim3(grid), dim3(64), 0, s, k); } while (0)
    // (8 rows: category-name texts computed to their EOS - a third of the 20-row variant's LDS, so 32 instead of 12 one-wave blocks per CU
    // hide each other's load latency)
    if (op_dtype == OFX_F16) { if (g.max_len <= 8) SA(f16_t, 8); else if (g.max_len <= 20) SA(f16_t, 20); else if (g.max_len <= 32) SA(f16_t, 32); else SA(f16_t, 64); }
    else { if (g.max_len <= 8) SA(bf16_t, 8); else if (g.max_len <= 20) SA(bf16_t, 20); else if (g.max_len <= 32) SA(bf16_t, 32); else SA(bf16_t, 64); }
#undef SA
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}

int ofx_launch_set_attention_bwd_mfma(const void* qkv, const float* d_o, void* dqkv, const int* cu, int nseq, int n_head, int D, int max_len,
                                      float scale, int op_dtype, const DropArgs& drop, int only_row0, hipStream_t s) {
    OFX_REQUIRE(D == n_head * 64 && max_len >= 1 && max_len <= 32, OFX_ESHAPE, "set_attention_bwd_mfma: bad shape");
    AttnBwdK k{(const char*)qkv, d_o, (char*)dqkv, cu, nseq, n_head, D, only_row0, scale, drop};
    const int grid = (nseq * n_head + 3) / 4;
    ProfScope prof(PROF_ATTN, s);
#define AB(T, N) hipLaunchKernelGGL((set_attention_bwd_mfma_kernel<T, N>), dim3(grid), dim3(256), 0, s, k)
    if (op_dtype == OFX_F16) { if (max_len <= 16) AB(f16_t, 1); else AB(f16_t, 2); }
    else { if (max_len <= 16) AB(bf16_t, 1); else AB(bf16_t, 2); }
#undef AB
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
