// Backward-pass kernels of the CP training step ("next" row N1): gradients of the global outfit
// Transformer on precomputed embeddings, i.e. what torch autograd computes for
// nn.TransformerEncoderLayer(norm_first, mish) x 6 + Linear(1024,1) in the reference's CP trainer
// (src/trains/trainers/compatibility_prediction_trainer.py:57-81) under FocalLoss (src/losses/focal_loss.py:23-41).
// Dense contractions run on the MFMA GEMMs (dgrad: the forward kernel on W^T; wgrad: the TN kernel of gemm.hip straight
// from row-major copies).  This file holds the rest: column sums (bias grads), LayerNorm backward (which also emits the
// operand-type copy of its output and that output's column sums), fp32 set-attention backward, head and loss kernels.
#include "ofx_common.h"

namespace {

template <typename TI> struct Ld4;
template <> struct Ld4<float> { static __device__ __forceinline__ f32x4 ld(const float* p) { return *(const f32x4*)p; } };
template <> struct Ld4<bf16_t> { static __device__ __forceinline__ f32x4 ld(const bf16_t* p) { const bf16x4 v = *(const bf16x4*)p; return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; } };
template <> struct Ld4<f16_t> { static __device__ __forceinline__ f32x4 ld(const f16_t* p) { const f16x4 v = *(const f16x4*)p; return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; } };

// ---- fp32 [R, C] -> operand-type transpose [C, ldd] (64 x 64 tiles through LDS).  Pack time only: the W^T copies the
// dgrad GEMMs contract against.
template <typename T>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* src, T* dst, int R, int C, int ldd, T* row_dst, int ld_row) {
    __shared__ T tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty * 16 + i, c = c0 + tx;
        const T v = (r < R && c < C) ? (T)src[(size_t)r * C + c] : (T)0.0f;
        tile[ty * 16 + i][tx] = v;
        if (row_dst && r < R && c < C) row_dst[(size_t)r * ld_row + c] = v;        // the row-major operand copy in the same pass
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty * 16 + i, r = r0 + tx;
        if (c < C && r < R) dst[(size_t)c * ldd + r] = tile[tx][ty * 16 + i];
    }
}

// ---- column sums, two deterministic stages.  Stage 1: block (x, y) sums rows [y*per, (y+1)*per) of columns 1024x .. +1023
// (4 per thread, 16-byte loads for fp32), optional row gather and per-row scale -> part[y][C].  Stage 2: out[c] = sum_y.
template <typename TI>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const TI* x, int ld, const int* gather, const float* row_scale, float* part, int C,
                                                           const int* m_dev, int M_static, int nchunk) {
    const int M = m_dev ? min(*m_dev, M_static) : M_static;
    const int col = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (col >= C) return;
    const int per = (M + nchunk - 1) / nchunk, r0 = blockIdx.y * per, r1 = min(M, r0 + per);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = Ld4<TI>::ld(x + (size_t)(gather ? gather[r + u] : r + u) * ld + col);
#pragma unroll
        for (int u = 0; u < 4; ++u) s += row_scale ? v[u] * row_scale[r + u] : v[u];
    }
    for (; r < r1; ++r) {
        const f32x4 v = Ld4<TI>::ld(x + (size_t)(gather ? gather[r] : r) * ld + col);
        s += row_scale ? v * row_scale[r] : v;
    }
    *(f32x4*)(part + (size_t)blockIdx.y * C + col) = s;
}
// out_k[c % seg] for c in segment k = c / seg (up to three destinations: LayerNorm's dgamma | dbeta | column sums)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* part, int C, int nchunk, float* out0, float* out1, float* out2, int seg, int valid,
                                                           int accumulate) {
    __shared__ float red[16][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;          // 16 columns x 16 row groups per block
    const int col = blockIdx.x * 16 + cx;
    float s0 = 0.f, s1 = 0.f;
    if (col < C) {
        int k = ry;
        for (; k + 16 < nchunk; k += 32) { s0 += part[(size_t)k * C + col]; s1 += part[(size_t)(k + 16) * C + col]; }
        if (k < nchunk) s0 += part[(size_t)k * C + col];
    }
    red[ry][cx] = s0 + s1;
    __syncthreads();
    if (ry == 0 && col < C) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][cx];
        float* o = col < seg ? out0 : (col < 2 * seg ? out1 : out2);
        const int c = col % seg;
        if (o && c < valid) o[c] = accumulate ? o[c] + s : s;
    }
}

// ---- LayerNorm backward.  One wave per row (grid-stride): dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma;
// out[r] = dx + add[r]; also stored as operand type (the next GEMMs' A operand; times the dropout mask of the
// linear layer below's output when training with dropout).  Per-block partial sums of
// dgamma = dy * xhat, dbeta = dy and of the OUTPUT's columns (= bias gradient of the linear layer below) -> part[block][3D].
template <int NCH, typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* dy, const float* x, const float* stats, const float* gamma, const float* add,
                                                     const int* add_map, float* dx_out, T* dx_op, float* part, const int* m_dev, int M_static, DropArgs drop) {
    typedef typename OpT<T>::v4 v4;
    constexpr int D = NCH * 256;
    __shared__ __attribute__((aligned(16))) float red[3 * D];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int M = m_dev ? min(*m_dev, M_static) : M_static;
    f32x4 dg[NCH], db[NCH], dc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { dg[c] = f32x4{0.f, 0.f, 0.f, 0.f}; db[c] = dg[c]; dc[c] = dg[c]; }
    for (int r = blockIdx.x * 4 + w; r < M; r += gridDim.x * 4) {
        const float mu = stats[2 * (size_t)r], rstd = stats[2 * (size_t)r + 1];
        f32x4 xh[NCH], g[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = (lane + 64 * c) * 4;
            const f32x4 d = *(const f32x4*)(dy + (size_t)r * D + col);
            xh[c] = (*(const f32x4*)(x + (size_t)r * D + col) - mu) * rstd;
            g[c] = d * *(const f32x4*)(gamma + col);
            dg[c] += d * xh[c];
            db[c] += d;
            s1 += g[c][0] + g[c][1] + g[c][2] + g[c][3];
            s2 += g[c][0] * xh[c][0] + g[c][1] * xh[c][1] + g[c][2] * xh[c][2] + g[c][3] * xh[c][3];
        }
        const float m1 = wave_sum(s1) * (1.0f / D), m2 = wave_sum(s2) * (1.0f / D);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = (lane + 64 * c) * 4;
            f32x4 o = (g[c] - m1 - xh[c] * m2) * rstd;
            if (add) {                 // add_map: `add` holds one row per outfit and only the mapped (prefix) rows receive it
                const int ar = add_map ? add_map[r] : r;
                if (ar >= 0) o += *(const f32x4*)(add + (size_t)ar * D + col);
            }
            *(f32x4*)(dx_out + (size_t)r * D + col) = o;
            if (drop.thresh) {           // the copy that feeds the linear layer below is the gradient of ITS (dropped-out) output
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] *= drop_mul(drop, r, col + e);
            }
            dc[c] += o;
            v4 ob;
#pragma unroll
            for (int e = 0; e < 4; ++e) ob[e] = (T)o[e];
            *(v4*)(dx_op + (size_t)r * D + col) = ob;
        }
    }
    // block reduction in a fixed order (wave 0, 1, 2, 3) through LDS, then one partial row per block
    for (int k = 0; k < 4; ++k) {
        if (w == k) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int col = (lane + 64 * c) * 4;
                OFX_LDS f32x4* a0 = (OFX_LDS f32x4*)(red + col);
                OFX_LDS f32x4* a1 = (OFX_LDS f32x4*)(red + D + col);
                OFX_LDS f32x4* a2 = (OFX_LDS f32x4*)(red + 2 * D + col);
                if (k == 0) { *a0 = dg[c]; *a1 = db[c]; *a2 = dc[c]; }
                else { *a0 += dg[c]; *a1 += db[c]; *a2 += dc[c]; }
            }
        }
        __syncthreads();
    }
    float* pw = part + (size_t)blockIdx.x * 3 * D;
    for (int i = threadIdx.x; i < 3 * D / 4; i += 256) *(f32x4*)(pw + 4 * i) = *(OFX_LDS f32x4*)(red + 4 * i);
}

// ---- fp32 set-attention backward, one wave per (outfit, head): recompute P = softmax(q k^T * scale), then
// dV = P^T dO, dP = dO V^T, dS = P * (dP - rowsum(dP * P)), dQ = dS K * scale, dK = dS^T Q * scale.
struct SetBwdK {
    const void* qkv;    // [rows, 3D] operand type (the tape keeps q|k|v as the GEMM wrote them)
    const float* d_o;   // [rows, D]
    void* dqkv;         // [rows, 3D] operand type (feeds the dgrad / wgrad GEMMs directly)
    const int* cu;
    int n_head, D;
    float scale;
    DropArgs drop;
    int only_row0;      // d_o is [nseq, D]: only query row 0 of every set has an upstream gradient (pruned last layer)
};
template <typename T, int SMAX>
__global__ __launch_bounds__(64) void set_attention_bwd_kernel(SetBwdK a) {
    constexpr int STR = 68, PS = SMAX + 4;
    __shared__ __attribute__((aligned(16))) float qs[SMAX * STR], ks[SMAX * STR], vs[SMAX * STR], gs[SMAX * STR];
    __shared__ __attribute__((aligned(16))) float P[SMAX * PS], dS[SMAX * PS];
    const int lane = threadIdx.x;
    const int b = blockIdx.x / a.n_head, h = blockIdx.x % a.n_head, D = a.D;
    const int r0 = a.cu[b];
    int S = a.cu[b + 1] - r0;
    S = S < SMAX ? S : SMAX;
    for (int j = 0; j < S; ++j) {
        const T* rp = (const T*)a.qkv + (size_t)(r0 + j) * 3 * D + h * 64 + lane;
        qs[j * STR + lane] = (float)rp[0];
        ks[j * STR + lane] = (float)rp[D];
        vs[j * STR + lane] = (float)rp[2 * D];
        gs[j * STR + lane] = a.only_row0 ? (j == 0 ? a.d_o[(size_t)b * D + h * 64 + lane] : 0.f) : a.d_o[(size_t)(r0 + j) * D + h * 64 + lane];
    }
    __syncthreads();
    for (int p = lane; p < S * S; p += 64) {                 // scores and dP = dO . V^T
        const int i = p / S, j = p % S;
        const f32x4 *qp = (const f32x4*)(qs + i * STR), *kp = (const f32x4*)(ks + j * STR);
        const f32x4 *gp = (const f32x4*)(gs + i * STR), *vp = (const f32x4*)(vs + j * STR);
        float s = 0.f, d = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const f32x4 x = qp[c], y = kp[c], g = gp[c], v = vp[c];
            s += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
            d += g[0] * v[0] + g[1] * v[1] + g[2] * v[2] + g[3] * v[3];
        }
        P[i * PS + j] = s * a.scale;
        dS[i * PS + j] = d;
    }
    __syncthreads();
    if (lane < S) {                                          // softmax row + dS row
        float* pr = P + lane * PS;
        float* dr = dS + lane * PS;
        float m = -INFINITY;
        for (int j = 0; j < S; ++j) m = fmaxf(m, pr[j]);
        float sum = 0.f;
        for (int j = 0; j < S; ++j) { const float e = expf(pr[j] - m); pr[j] = e; sum += e; }
        const float inv = 1.0f / sum;
        float dot = 0.f;
        if (a.drop.thresh) {          // O = (P . m) V:  dP = (dO V^T) . m, dV uses P . m; softmax backward uses the undropped P
            for (int j = 0; j < S; ++j) {
                const float mk = drop_mul(a.drop, blockIdx.x, lane * 32 + j);
                pr[j] *= inv; dr[j] *= mk; dot += pr[j] * dr[j];
            }
            for (int j = 0; j < S; ++j) {
                dr[j] = pr[j] * (dr[j] - dot) * a.scale;
                pr[j] *= drop_mul(a.drop, blockIdx.x, lane * 32 + j);
            }
        } else {
            for (int j = 0; j < S; ++j) { pr[j] *= inv; dot += pr[j] * dr[j]; }
            for (int j = 0; j < S; ++j) dr[j] = pr[j] * (dr[j] - dot) * a.scale;
        }
    }
    __syncthreads();
    // per row j (lane = feature): dQ[j] = sum_i dS[j][i] K[i]; dK[j] = sum_i dS[i][j] Q[i]; dV[j] = sum_i P[i][j] dO[i]
    for (int j = 0; j < S; ++j) {
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int i = 0; i < S; ++i) {
            dq += dS[j * PS + i] * ks[i * STR + lane];
            dk += dS[i * PS + j] * qs[i * STR + lane];
            dv += P[i * PS + j] * gs[i * STR + lane];
        }
        T* op = (T*)a.dqkv + (size_t)(r0 + j) * 3 * D + h * 64 + lane;
        op[0] = (T)dq; op[D] = (T)dk; op[2 * D] = (T)dv;
    }
}

// ---- heads and loss
// FocalLoss (mean) forward + dlogits: ce = BCEWithLogits, p = sigmoid, p_t = p y + (1-p)(1-y), a_t = a y + (1-a)(1-y),
// loss_i = a_t ce (1-p_t)^g.  One block.
// reduction (focal_loss.py:36-41): 1 mean (loss and gradient / B), 2 sum, 0 none (`per_elem` receives the B unreduced losses and
// dlogits the per-element derivative; `loss` then still gets their sum)
__global__ __launch_bounds__(256) void focal_loss_kernel(const float* logits, const float* labels, int B, float alpha, float gamma, float up,
                                                         float* loss, float* dlogits, int reduction, float* per_elem) {
    __shared__ float red[256];
    float acc = 0.f;
    const float inv = reduction == 1 ? 1.0f / B : 1.0f;
    for (int i = threadIdx.x; i < B; i += 256) {
        const float x = logits[i], y = labels[i];
        const float ce = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
        const float p = 1.0f / (1.0f + expf(-x));
        const float pt = p * y + (1.f - p) * (1.f - y), at = alpha * y + (1.f - alpha) * (1.f - y);
        const float om = 1.f - pt, mod = powf(fmaxf(om, 0.f), gamma);
        acc += at * ce * mod;
        if (per_elem) per_elem[i] = at * ce * mod;
        if (dlogits) {
            // d ce/dx = p - y ; d pt/dx = p(1-p)(2y-1) ; d mod/dx = -g (1-pt)^(g-1) d pt/dx
            const float dpt = p * (1.f - p) * (2.f * y - 1.f);
            const float dmod = om > 0.f ? -gamma * powf(om, gamma - 1.f) * dpt : 0.f;
            dlogits[i] = up * at * ((p - y) * mod + ce * dmod) * inv;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0 && loss) *loss = red[0] * inv;
}


// CP head backward, row part: dX[cu[b]] = dlogit[b] * w (fp32 + operand copy; the rest of dX was zeroed: gradient of
// "take row 0" then Linear(D, 1)); block 0 also writes db = sum_b dlogit[b].
// With dropout: the head sees row0 . m_head (mask row = outfit), and the operand copy carries the mask of the last
// layer's dropout2 (mask row = global row), like the copies LayerNorm backward emits.
template <typename T>
__global__ __launch_bounds__(256) void cp_head_bwd_kernel(const float* dlogits, const float* w, const int* cu, float* dX, T* dXb, float* db, int B, int D,
                                                         DropArgs head, DropArgs below, int accumulate) {
    typedef typename OpT<T>::v4 v4;
    const int b = blockIdx.x;
    const int r = cu ? cu[b] : b;
    const size_t row = (size_t)r * D;
    // dlogits == nullptr: `w` is a ready [B, D] matrix of row gradients (CIR head) instead of the CP head's weight vector
    const float g = dlogits ? dlogits[b] : 1.0f;
    const float* wv = dlogits ? w : w + (size_t)b * D;
    for (int c = threadIdx.x * 4; c < D; c += 1024) {
        f32x4 v = *(const f32x4*)(wv + c) * g;
        if (head.thresh) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= drop_mul(head, b, c + e);
        }
        *(f32x4*)(dX + row + c) = v;
        v4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (T)(below.thresh ? v[e] * drop_mul(below, r, c + e) : v[e]);
        *(v4*)(dXb + row + c) = o;
    }
    if (b == 0 && db) {
        __shared__ float red[256];
        float s = 0.f;
        for (int i = threadIdx.x; i < B; i += 256) s += dlogits[i];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) db[0] = accumulate ? db[0] + red[0] : red[0];
    }
}

// map[cu[b]] = b (the rest of map was set to -1): which outfit's prefix row a pad-free row is
__global__ __launch_bounds__(256) void row_map_kernel(const int* cu, int* map, int B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B) map[cu[b]] = b;
}

// x[r][c] *= mask(r, c)   (head dropout on the pooled rows; also exports a mask for the tests when x is all ones)
__global__ __launch_bounds__(256) void drop_rows_kernel(float* x, int rows, int cols, DropArgs d) {
    const size_t total = (size_t)rows * cols;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        x[i] *= drop_mul(d, (unsigned)(i / cols), (unsigned)(i % cols));
}

}  // namespace

#define BWD_CHECK() OFX_LAUNCH_CHECK()

int ofx_launch_row_map(const int* cu, int* map, int B, int M, hipStream_t s) {
    OFX_HIP(hipMemsetAsync(map, 0xFF, (size_t)M * 4, s));
    hipLaunchKernelGGL(row_map_kernel, dim3((B + 255) / 256), dim3(256), 0, s, cu, map, B);
    BWD_CHECK();
    return OFX_OK;
}
int ofx_launch_drop_rows(float* x, int rows, int cols, const DropArgs& d, hipStream_t s) {
    if (!d.thresh) return OFX_OK;
    size_t total = (size_t)rows * cols;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(drop_rows_kernel, dim3(grid), dim3(256), 0, s, x, rows, cols, d);
    BWD_CHECK();
    return OFX_OK;
}

int ofx_launch_transpose_cast(const float* src, void* dst, int R, int C, int ldd, int op_dtype, hipStream_t s, void* row_dst, int ld_row) {
    OFX_REQUIRE(ldd >= R && (!row_dst || ld_row >= C), OFX_ESHAPE, "transpose_cast: ldd=%d < R=%d", ldd, R);
    const dim3 grid((C + 63) / 64, (R + 63) / 64);
    if (op_dtype == OFX_F16) hipLaunchKernelGGL(transpose_cast_kernel<f16_t>, grid, dim3(256), 0, s, src, (f16_t*)dst, R, C, ldd, (f16_t*)row_dst, ld_row);
    else hipLaunchKernelGGL(transpose_cast_kernel<bf16_t>, grid, dim3(256), 0, s, src, (bf16_t*)dst, R, C, ldd, (bf16_t*)row_dst, ld_row);
    BWD_CHECK();
    return OFX_OK;
}

// out[c] = sum_r scale[r] * x[gather ? gather[r] : r][c]; C a multiple of 4; part: ofx_colsum_part_floats(C) floats.
// Up to three outputs: column c goes to out_k[c % seg], k = c / seg (seg = C for a single output).
constexpr int COLSUM_MAX_CHUNKS = 256;
size_t ofx_colsum_part_floats(int C) { return (size_t)COLSUM_MAX_CHUNKS * C; }
int ofx_launch_colsum(const void* x, int x_kind /*0 fp32 | 1 operand type*/, int ld, const int* gather, const float* row_scale, float* out0, float* out1,
                      float* out2, int seg, float* part, int C, const int* m_dev, int M, int op_dtype, hipStream_t s, int valid, int accumulate) {
    if (valid <= 0) valid = seg;
    OFX_REQUIRE(C % 4 == 0 && M > 0 && (ld % 4 == 0), OFX_ESHAPE, "colsum: C=%d ld=%d", C, ld);
    int nchunk = M / 16; nchunk = nchunk < 1 ? 1 : (nchunk > COLSUM_MAX_CHUNKS ? COLSUM_MAX_CHUNKS : nchunk);
    ProfScope prof(PROF_OTHER, s);
    const dim3 grid((C + 1023) / 1024, nchunk);
    if (x_kind == 0) hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ld, gather, row_scale, part, C, m_dev, M, nchunk);
    else if (op_dtype == OFX_F16) hipLaunchKernelGGL(colsum_partial_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)x, ld, gather, row_scale, part, C, m_dev, M, nchunk);
    else hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, ld, gather, row_scale, part, C, m_dev, M, nchunk);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16), dim3(256), 0, s, part, C, nchunk, out0, out1, out2, seg, valid, accumulate);
    BWD_CHECK();
    return OFX_OK;
}

constexpr int LN_BWD_BLOCKS = 256;
size_t ofx_ln_bwd_part_floats(int D) { return (size_t)LN_BWD_BLOCKS * 3 * D; }
int ofx_launch_ln_bwd(const float* dy, const float* x, const float* stats, const float* gamma, const float* add, const int* add_map, float* dx_out, void* dx_op,
                      float* dgamma, float* dbeta, float* dcols, float* part, int D, const int* m_dev, int M, int op_dtype, const DropArgs& drop, hipStream_t s,
                      int accumulate) {
    OFX_REQUIRE(D == 512 || D == 768 || D == 1024, OFX_ESHAPE, "ln_bwd: D=%d", D);
    ProfScope prof(PROF_NORM, s);
#define LNB(NCH, T) hipLaunchKernelGGL((ln_bwd_kernel<NCH, T>), dim3(LN_BWD_BLOCKS), dim3(256), 0, s, dy, x, stats, gamma, add, add_map, dx_out, (T*)dx_op, part, m_dev, M, drop)
    if (op_dtype == OFX_F16) { if (D == 1024) LNB(4, f16_t); else if (D == 768) LNB(3, f16_t); else LNB(2, f16_t); }
    else { if (D == 1024) LNB(4, bf16_t); else if (D == 768) LNB(3, bf16_t); else LNB(2, bf16_t); }
#undef LNB
    hipLaunchKernelGGL(colsum_final_kernel, dim3((3 * D + 15) / 16), dim3(256), 0, s, part, 3 * D, LN_BWD_BLOCKS, dgamma, dbeta, dcols, D, D, accumulate);
    BWD_CHECK();
    return OFX_OK;
}

int ofx_launch_set_attention_bwd(const void* qkv, const float* d_o, void* dqkv, const int* cu, int nseq, int n_head, int D, int max_len,
                                 float scale, int op_dtype, const DropArgs& drop, int only_row0, hipStream_t s) {
    OFX_REQUIRE(D == n_head * 64 && max_len >= 1 && max_len <= 32, OFX_ESHAPE, "set_attention_bwd: bad shape");
    SetBwdK k{qkv, d_o, dqkv, cu, n_head, D, scale, drop, only_row0};
    ProfScope prof(PROF_ATTN, s);
#define SAB(T, S) hipLaunchKernelGGL((set_attention_bwd_kernel<T, S>), dim3(nseq * n_head), dim3(64), 0, s, k)
    if (op_dtype == OFX_F16) { if (max_len <= 20) SAB(f16_t, 20); else SAB(f16_t, 32); }
    else { if (max_len <= 20) SAB(bf16_t, 20); else SAB(bf16_t, 32); }
#undef SAB
    BWD_CHECK();
    return OFX_OK;
}

int ofx_launch_focal_loss(const float* logits, const float* labels, int B, float alpha, float gamma, float upstream, float* loss, float* dlogits, hipStream_t s,
                          int reduction, float* per_elem) {
    hipLaunchKernelGGL(focal_loss_kernel, dim3(1), dim3(256), 0, s, logits, labels, B, alpha, gamma, upstream, loss, dlogits, reduction, per_elem);
    BWD_CHECK();
    return OFX_OK;
}
int ofx_launch_cp_head_bwd(const float* dlogits, const float* w, const int* cu, float* dX, void* dXb, float* db, int B, int D, int op_dtype,
                           const DropArgs& head, const DropArgs& below, hipStream_t s, int accumulate) {
    OFX_REQUIRE(D % 4 == 0, OFX_ESHAPE, "cp_head_bwd: D=%d", D);
    if (op_dtype == OFX_F16) hipLaunchKernelGGL(cp_head_bwd_kernel<f16_t>, dim3(B), dim3(256), 0, s, dlogits, w, cu, dX, (f16_t*)dXb, db, B, D, head, below, accumulate);
    else hipLaunchKernelGGL(cp_head_bwd_kernel<bf16_t>, dim3(B), dim3(256), 0, s, dlogits, w, cu, dX, (bf16_t*)dXb, db, B, D, head, below, accumulate);
    BWD_CHECK();
    return OFX_OK;
}
