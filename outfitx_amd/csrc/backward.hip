// Backward-pass kernels of the CP training step ("next" row N1): gradients of the global outfit
// Transformer on precomputed embeddings, i.e. what torch autograd computes for
// nn.TransformerEncoderLayer(norm_first, mish) x 6 + Linear(1024,1) in the reference's CP trainer
// (src/trains/trainers/compatibility_prediction_trainer.py:57-81) under FocalLoss (src/losses/focal_loss.py:23-41).
// Dense contractions (dgrad / wgrad) reuse the forward MFMA GEMM with transposed operands; this file holds the rest:
// cast+transpose, column sums (bias grads), LayerNorm backward, fp32 set-attention backward, head and loss kernels.
#include "ofx_common.h"

namespace {

// ---- src [M_live, C] (fp32 or operand type) -> optional row-major operand copy [M, ldr] and transposed operand copy
// [C, Mpad] with zero columns for rows >= M_live (the wgrad GEMM contracts over Mpad).  64 x 64 tiles through LDS.
template <typename TI, typename T>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const TI* src, int ld_src, T* row_out, int ld_row, T* t_out, int Mpad,
                                                           int C, const int* m_dev, int M_static) {
    __shared__ T tile[64][66];
    const int M = m_dev ? min(*m_dev, M_static) : M_static;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty * 16 + i, c = c0 + tx;
        T v = (T)0.0f;
        if (r < M && c < C) v = (T)(float)src[(size_t)r * ld_src + c];
        tile[ty * 16 + i][tx] = v;
        if (row_out && r < M && c < C) row_out[(size_t)r * ld_row + c] = v;
    }
    __syncthreads();
    if (t_out) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = c0 + ty * 16 + i, r = r0 + tx;
            if (c < C && r < Mpad) t_out[(size_t)c * Mpad + r] = tile[tx][ty * 16 + i];
        }
    }
}

// ---- column sums, two deterministic stages: part[chunk][C] then out[C] (+= when accumulate)
template <typename TI>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const TI* x, int ld, float* part, int C, const int* m_dev, int M_static, int nchunk) {
    const int M = m_dev ? min(*m_dev, M_static) : M_static;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= C) return;
    const int per = (M + nchunk - 1) / nchunk, r0 = blockIdx.y * per, r1 = min(M, r0 + per);
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += (float)x[(size_t)r * ld + col];
    part[(size_t)blockIdx.y * C + col] = s;
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* part, float* out, int C, int nchunk, int accumulate) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= C) return;
    float s = 0.f;
    for (int k = 0; k < nchunk; ++k) s += part[(size_t)k * C + col];
    out[col] = accumulate ? out[col] + s : s;
}

// ---- LayerNorm backward.  One wave per row (grid-stride): dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma;
// dx_out[r] = dx (+ add[r] when given).  Per-wave partial sums of dgamma = dy * xhat and dbeta = dy go to part[wave][2D].
template <int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* dy, const float* x, const float* stats, const float* gamma, const float* add,
                                                     float* dx_out, float* part, const int* m_dev, int M_static) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, D = NCH * 256;
    const int M = m_dev ? min(*m_dev, M_static) : M_static;
    f32x4 dg[NCH], db[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { dg[c] = f32x4{0.f, 0.f, 0.f, 0.f}; db[c] = dg[c]; }
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
            if (add) o += *(const f32x4*)(add + (size_t)r * D + col);
            *(f32x4*)(dx_out + (size_t)r * D + col) = o;
        }
    }
    float* pw = part + (size_t)(blockIdx.x * 4 + w) * 2 * D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (lane + 64 * c) * 4;
        *(f32x4*)(pw + col) = dg[c];
        *(f32x4*)(pw + D + col) = db[c];
    }
}

// ---- fp32 set-attention backward, one wave per (outfit, head): recompute P = softmax(q k^T * scale), then
// dV = P^T dO, dP = dO V^T, dS = P * (dP - rowsum(dP * P)), dQ = dS K * scale, dK = dS^T Q * scale.
struct SetBwdK {
    const float* qkv;   // [rows, 3D]
    const float* d_o;   // [rows, D]
    float* dqkv;        // [rows, 3D]
    const int* cu;
    int n_head, D;
    float scale;
};
template <int SMAX>
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
        const float* rp = a.qkv + (size_t)(r0 + j) * 3 * D + h * 64 + lane;
        qs[j * STR + lane] = rp[0];
        ks[j * STR + lane] = rp[D];
        vs[j * STR + lane] = rp[2 * D];
        gs[j * STR + lane] = a.d_o[(size_t)(r0 + j) * D + h * 64 + lane];
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
        for (int j = 0; j < S; ++j) { pr[j] *= inv; dot += pr[j] * dr[j]; }
        for (int j = 0; j < S; ++j) dr[j] = pr[j] * (dr[j] - dot) * a.scale;
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
        float* op = a.dqkv + (size_t)(r0 + j) * 3 * D + h * 64 + lane;
        op[0] = dq; op[D] = dk; op[2 * D] = dv;
    }
}

// ---- heads and loss
// FocalLoss (mean) forward + dlogits: ce = BCEWithLogits, p = sigmoid, p_t = p y + (1-p)(1-y), a_t = a y + (1-a)(1-y),
// loss_i = a_t ce (1-p_t)^g.  One block.
__global__ __launch_bounds__(256) void focal_loss_kernel(const float* logits, const float* labels, int B, float alpha, float gamma, float up,
                                                         float* loss, float* dlogits) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) {
        const float x = logits[i], y = labels[i];
        const float ce = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
        const float p = 1.0f / (1.0f + expf(-x));
        const float pt = p * y + (1.f - p) * (1.f - y), at = alpha * y + (1.f - alpha) * (1.f - y);
        const float om = 1.f - pt, mod = powf(fmaxf(om, 0.f), gamma);
        acc += at * ce * mod;
        if (dlogits) {
            // d ce/dx = p - y ; d pt/dx = p(1-p)(2y-1) ; d mod/dx = -g (1-pt)^(g-1) d pt/dx
            const float dpt = p * (1.f - p) * (2.f * y - 1.f);
            const float dmod = om > 0.f ? -gamma * powf(om, gamma - 1.f) * dpt : 0.f;
            dlogits[i] = up * at * ((p - y) * mod + ce * dmod) / B;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0 && loss) *loss = red[0] / B;
}

// CP head backward: d_row0[b] = dlogit[b] * w ; dw = sum_b dlogit[b] * row0[b] ; db = sum_b dlogit[b]
__global__ __launch_bounds__(256) void cp_head_bwd_kernel(const float* dlogits, const float* row0, const float* w, float* d_row0, float* dw, float* db,
                                                         int B, int D, int accumulate) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col < D) {
        float s = 0.f;
        const float wc = w[col];
        for (int b = 0; b < B; ++b) {
            const float g = dlogits[b];
            s += g * row0[(size_t)b * D + col];
            d_row0[(size_t)b * D + col] = g * wc;
        }
        dw[col] = accumulate ? dw[col] + s : s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dlogits[b];
        db[0] = accumulate ? db[0] + s : s;
    }
}

// dX[cu[b]] = d_row0[b], everything else zero (gradient of "take row 0")
__global__ __launch_bounds__(256) void scatter_row0_kernel(const float* d_row0, const int* cu, float* dX, int B, int D, size_t total4, const int* m_dev) {
    const size_t live = m_dev ? (size_t)*m_dev * D / 4 : total4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < live; i += (size_t)gridDim.x * blockDim.x) ((f32x4*)dX)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}
__global__ __launch_bounds__(256) void scatter_row0_write_kernel(const float* d_row0, const int* cu, float* dX, int B, int D) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int b = blockIdx.x * 4 + w; b < B; b += gridDim.x * 4)
        for (int c = lane; c < D / 4; c += 64) *(f32x4*)(dX + (size_t)cu[b] * D + c * 4) = *(const f32x4*)(d_row0 + (size_t)b * D + c * 4);
}
// out[col] (+)= sum_b dX[cu[b]][col]  (gradient of the shared prefix token), cols [c0, c0+n)
__global__ __launch_bounds__(256) void prefix_grad_kernel(const float* dX, const int* cu, float* out, int B, int D, int c0, int n, int accumulate) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= n) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dX[(size_t)cu[b] * D + c0 + col];
    out[col] = accumulate ? out[col] + s : s;
}

}  // namespace

#define BWD_CHECK() OFX_LAUNCH_CHECK()

int ofx_launch_cast_transpose(const void* src, int src_is_f32, int ld_src, void* row_out, int ld_row, void* t_out, int Mpad, int C,
                              const int* m_dev, int M, int op_dtype, hipStream_t s) {
    OFX_REQUIRE(Mpad % 64 == 0 && Mpad >= M, OFX_ESHAPE, "cast_transpose: Mpad=%d must be a multiple of 64 and >= M=%d", Mpad, M);
    ProfScope prof(PROF_OTHER, s);
    const dim3 grid((C + 63) / 64, Mpad / 64);
#define CT(TI, T) hipLaunchKernelGGL((cast_transpose_kernel<TI, T>), grid, dim3(256), 0, s, (const TI*)src, ld_src, (T*)row_out, ld_row, (T*)t_out, Mpad, C, m_dev, M)
    if (op_dtype == OFX_F16) { if (src_is_f32) CT(float, f16_t); else CT(f16_t, f16_t); }
    else { if (src_is_f32) CT(float, bf16_t); else CT(bf16_t, bf16_t); }
#undef CT
    BWD_CHECK();
    return OFX_OK;
}

int ofx_launch_colsum(const void* x, int x_is_f32, int ld, float* out, float* part, int C, const int* m_dev, int M, int accumulate, int op_dtype, hipStream_t s) {
    const int nchunk = 32;
    ProfScope prof(PROF_OTHER, s);
    const dim3 grid((C + 255) / 256, nchunk);
    if (x_is_f32) hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ld, part, C, m_dev, M, nchunk);
    else if (op_dtype == OFX_F16) hipLaunchKernelGGL(colsum_partial_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)x, ld, part, C, m_dev, M, nchunk);
    else hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, ld, part, C, m_dev, M, nchunk);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, s, part, out, C, nchunk, accumulate);
    BWD_CHECK();
    return OFX_OK;
}

constexpr int LN_BWD_BLOCKS = 256;
int ofx_launch_ln_bwd(const float* dy, const float* x, const float* stats, const float* gamma, const float* add, float* dx_out,
                      float* dgamma, float* dbeta, float* part /*[LN_BWD_BLOCKS*4, 2D] + [32, 2D]*/, int D, const int* m_dev, int M, int accumulate, hipStream_t s) {
    OFX_REQUIRE(D == 512 || D == 768 || D == 1024, OFX_ESHAPE, "ln_bwd: D=%d", D);
    ProfScope prof(PROF_NORM, s);
    if (D == 1024) hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3(LN_BWD_BLOCKS), dim3(256), 0, s, dy, x, stats, gamma, add, dx_out, part, m_dev, M);
    else if (D == 768) hipLaunchKernelGGL(ln_bwd_kernel<3>, dim3(LN_BWD_BLOCKS), dim3(256), 0, s, dy, x, stats, gamma, add, dx_out, part, m_dev, M);
    else hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(LN_BWD_BLOCKS), dim3(256), 0, s, dy, x, stats, gamma, add, dx_out, part, m_dev, M);
    // reduce the per-wave partials [LN_BWD_BLOCKS*4, 2D]: column sums of a fp32 matrix with 2D columns
    float* part2 = part + (size_t)LN_BWD_BLOCKS * 4 * 2 * D;
    const int nchunk = 32, C = 2 * D, rows = LN_BWD_BLOCKS * 4;
    hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3((C + 255) / 256, nchunk), dim3(256), 0, s, part, C, part2, C, nullptr, rows, nchunk);
    // dgamma = first D columns, dbeta = last D
    hipLaunchKernelGGL(colsum_final_kernel, dim3((D + 255) / 256), dim3(256), 0, s, part2, dgamma, C, nchunk, accumulate);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((D + 255) / 256), dim3(256), 0, s, part2 + D, dbeta, C, nchunk, accumulate);
    BWD_CHECK();
    return OFX_OK;
}
size_t ofx_ln_bwd_part_floats(int D) { return (size_t)LN_BWD_BLOCKS * 4 * 2 * D + (size_t)32 * 2 * D; }

int ofx_launch_set_attention_bwd(const float* qkv, const float* d_o, float* dqkv, const int* cu, int nseq, int n_head, int D, int max_len,
                                 float scale, hipStream_t s) {
    OFX_REQUIRE(D == n_head * 64 && max_len >= 1 && max_len <= 32, OFX_ESHAPE, "set_attention_bwd: bad shape");
    SetBwdK k{qkv, d_o, dqkv, cu, n_head, D, scale};
    ProfScope prof(PROF_ATTN, s);
    if (max_len <= 20) hipLaunchKernelGGL(set_attention_bwd_kernel<20>, dim3(nseq * n_head), dim3(64), 0, s, k);
    else hipLaunchKernelGGL(set_attention_bwd_kernel<32>, dim3(nseq * n_head), dim3(64), 0, s, k);
    BWD_CHECK();
    return OFX_OK;
}

int ofx_launch_focal_loss(const float* logits, const float* labels, int B, float alpha, float gamma, float upstream, float* loss, float* dlogits, hipStream_t s) {
    hipLaunchKernelGGL(focal_loss_kernel, dim3(1), dim3(256), 0, s, logits, labels, B, alpha, gamma, upstream, loss, dlogits);
    BWD_CHECK();
    return OFX_OK;
}
int ofx_launch_cp_head_bwd(const float* dlogits, const float* row0, const float* w, float* d_row0, float* dw, float* db, int B, int D, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(cp_head_bwd_kernel, dim3((D + 255) / 256), dim3(256), 0, s, dlogits, row0, w, d_row0, dw, db, B, D, accumulate);
    BWD_CHECK();
    return OFX_OK;
}
int ofx_launch_scatter_row0(const float* d_row0, const int* cu, float* dX, int B, int D, int M, const int* m_dev, hipStream_t s) {
    const size_t total4 = (size_t)M * D / 4;
    int grid = (int)((total4 + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(scatter_row0_kernel, dim3(grid), dim3(256), 0, s, d_row0, cu, dX, B, D, total4, m_dev);
    hipLaunchKernelGGL(scatter_row0_write_kernel, dim3((B + 3) / 4 > 4096 ? 4096 : (B + 3) / 4), dim3(256), 0, s, d_row0, cu, dX, B, D);
    BWD_CHECK();
    return OFX_OK;
}
int ofx_launch_prefix_grad(const float* dX, const int* cu, float* out, int B, int D, int c0, int n, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(prefix_grad_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dX, cu, out, B, D, c0, n, accumulate);
    BWD_CHECK();
    return OFX_OK;
}
