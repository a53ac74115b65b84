// HBM-bound row kernels of the scoring path: LayerNorm (+gather, +operand cast, +hi/lo split),
// ViT patchify / embedding assembly, CLIP text embedding, L2-normalise+concat fuser, set build
// (prefix token + pad-free compaction), row-0 gather, CP head.  One wave per row, 16-byte
// accesses, fp32 statistics.  Reference arithmetic: SURVEY.md Appendix A items 1-5.
#include <algorithm>
#include "ofx_common.h"

namespace {

template <typename T>
__device__ __forceinline__ void store4(void* y, size_t off, f32x4 v, int out_kind, int D) {
    if (out_kind == 0) {
        *(f32x4*)((float*)y + off) = v;
    } else {
        typedef typename OpT<T>::v4 v4;
        v4 hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) hi[e] = (T)v[e];
        T* p = (T*)y + off;
        *(v4*)p = hi;
        if (out_kind == 2) {
            v4 lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) lo[e] = (T)(v[e] - (float)hi[e]);
            *(v4*)(p + D) = lo;
            *(v4*)(p + 2 * D) = hi;
        }
    }
}

// y[r] = LN(x[row_idx ? row_idx[r] : r]) * gamma + beta.  NCH = D/256 float4 chunks per lane.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs a, const int* rows_dev) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rows = rows_dev ? *rows_dev : a.rows;
    const int D = NCH * 256;
    for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
        const int src = a.row_idx ? a.row_idx[r] : r;
        const f32x4* xp = (const f32x4*)(a.x + (size_t)src * D);
        f32x4 v[NCH];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            v[c] = xp[lane + 64 * c];
            s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
        }
        const float mu = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            v[c] -= mu;
            q += v[c][0] * v[c][0] + v[c][1] * v[c][1] + v[c][2] * v[c][2] + v[c][3] * v[c][3];
        }
        const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + a.eps);
        if (a.stats && lane == 0) { a.stats[2 * (size_t)r] = mu; a.stats[2 * (size_t)r + 1] = rstd; }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = (lane + 64 * c) * 4;
            const f32x4 g = *(const f32x4*)(a.gamma + col), b = *(const f32x4*)(a.beta + col);
            store4<T>(a.y, (size_t)r * a.ldy + col, v[c] * rstd * g + b, a.out_kind, D);
        }
    }
}

}  // namespace

int ofx_launch_layernorm_dev(const LnArgs& a, const int* rows_dev, int op_dtype, hipStream_t s) {
    OFX_REQUIRE(a.D == 512 || a.D == 768 || a.D == 1024, OFX_ESHAPE, "layernorm: D=%d not in {512,768,1024}", a.D);
    OFX_REQUIRE(a.rows > 0, OFX_ESHAPE, "layernorm: rows=%d", a.rows);
    OFX_REQUIRE(a.ldy % 4 == 0 && a.ldy >= (a.out_kind == 2 ? 3 * a.D : a.D), OFX_ESHAPE, "layernorm: bad ldy=%d", a.ldy);
    int grid = (a.rows + 3) / 4;
    if (grid > 8192) grid = 8192;
    ProfScope prof(PROF_NORM, s);
#define LN_CASE(T, N) hipLaunchKernelGGL((layernorm_kernel<T, N>), dim3(grid), dim3(256), 0, s, a, rows_dev)
    if (op_dtype == OFX_F16) {
        if (a.D == 512) LN_CASE(f16_t, 2); else if (a.D == 768) LN_CASE(f16_t, 3); else LN_CASE(f16_t, 4);
    } else {
        if (a.D == 512) LN_CASE(bf16_t, 2); else if (a.D == 768) LN_CASE(bf16_t, 3); else LN_CASE(bf16_t, 4);
    }
#undef LN_CASE
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_layernorm(const LnArgs& a, int op_dtype, hipStream_t s) { return ofx_launch_layernorm_dev(a, nullptr, op_dtype, s); }

// ------------------------------------------------------------------------------------------------
// Split-K second pass fused with the LayerNorm that follows it (small batches of the outfit transformer: one launch instead of
// two): x[r] = sum_s slab[s][r] (fixed order) + bias + resid[r] -> fp32 x (may alias resid), then y[r] = LN(x[r]) (the block-wide sums
// take another order than layernorm_kernel's single-wave ones: the two paths agree to fp32 rounding, not bit for bit).
namespace {
// One block per row, one float4 per thread (D = 4 * blockDim.x): the slab loads of a row are spread over D / 4 lanes, so even
// 16 slabs are one short load burst; the two LayerNorm reductions go wave_sum -> LDS -> every thread re-adds the per-wave partials
// in the same order (any wave count gives every thread the same value).
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_ln_kernel(SplitKLnArgs a) {
    __shared__ float part[2][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int rows = a.m_dev ? min(*a.m_dev, a.rows) : a.rows;
    const int D = a.D, col = threadIdx.x * 4;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const float* sp = a.slab + (size_t)r * D + col;
        f32x4 t = *(const f32x4*)sp;
        for (int k = 1; k < a.splits; ++k) t += *(const f32x4*)(sp + k * a.plane);
        if (a.bias) t += *(const f32x4*)(a.bias + col);
        if (a.resid) t += *(const f32x4*)(a.resid + (size_t)r * a.ldr + col);
        *(f32x4*)(a.x + (size_t)r * a.ldx + col) = t;
        const float s = wave_sum(t[0] + t[1] + t[2] + t[3]);
        if (lane == 0) part[0][w] = s;
        __syncthreads();
        float tot = 0.f;
        for (int i = 0; i < nw; ++i) tot += part[0][i];
        const float mu = tot * (1.0f / D);
        t -= mu;
        const float q = wave_sum(t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3]);
        if (lane == 0) part[1][w] = q;
        __syncthreads();
        float tq = 0.f;
        for (int i = 0; i < nw; ++i) tq += part[1][i];
        const float rstd = rsqrtf(tq * (1.0f / D) + a.eps);
        const f32x4 g = *(const f32x4*)(a.gamma + col), b = *(const f32x4*)(a.beta + col);
        store4<T>(a.y, (size_t)r * a.ldy + col, t * rstd * g + b, a.out_kind, D);
        __syncthreads();                    // part[] is reused by the block's next row
    }
}
}  // namespace

int ofx_launch_splitk_reduce_ln(const SplitKLnArgs& a, int op_dtype, hipStream_t s, bool in_gemm_scope) {
    OFX_REQUIRE(a.D == 512 || a.D == 768 || a.D == 1024, OFX_ESHAPE, "splitk_reduce_ln: D=%d not in {512,768,1024}", a.D);
    OFX_REQUIRE(a.rows > 0 && a.splits >= 1 && a.ldx % 4 == 0 && a.ldx >= a.D && (!a.resid || (a.ldr % 4 == 0 && a.ldr >= a.D)), OFX_ESHAPE, "splitk_reduce_ln: bad shape");
    OFX_REQUIRE(a.ldy % 4 == 0 && a.ldy >= (a.out_kind == 2 ? 3 * a.D : a.D), OFX_ESHAPE, "splitk_reduce_ln: bad ldy=%d", a.ldy);
    const int grid = a.rows > 8192 ? 8192 : a.rows;
    ProfScope prof(PROF_NORM, s, 0.0, false, !in_gemm_scope);
    if (op_dtype == OFX_F16) OFX_PLAUNCH(in_gemm_scope, splitk_reduce_ln_kernel<f16_t>, dim3(grid), dim3(a.D / 4), 0, s, a);
    else OFX_PLAUNCH(in_gemm_scope, splitk_reduce_ln_kernel<bf16_t>, dim3(grid), dim3(a.D / 4), 0, s, a);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}

// ------------------------------------------------------------------------------------------------
// fp32 -> operand type (optionally hi|lo|hi or hi|hi|lo split along the row): weight packing and
// activation casts.  mode 0: plain, 1: [hi|lo|hi] (activations), 2: [hi|hi|lo] (weights).
// src [rows, K] fp32 with K_src valid columns (zero-padded to K_dst); dst [rows_dst, ld].
namespace {
template <typename T>
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* src, T* dst, int rows_src, int rows_dst, int K_src,
                                                       int K_dst, int ld_src, int mode) {
    const int chunks = K_dst / 4;
    const size_t total = (size_t)rows_dst * chunks;
    const int ld = mode == 3 ? 2 * K_dst : (mode ? 3 * K_dst : K_dst);      // mode 3: split weights [hi | lo] (GemmArgs::a_wrap)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / chunks), c = (int)(i % chunks) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < rows_src) {
            if (c + 3 < K_src && (ld_src % 4 == 0)) v = *(const f32x4*)(src + (size_t)r * ld_src + c);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c + e < K_src) v[e] = src[(size_t)r * ld_src + c + e];
            }
        }
        typedef typename OpT<T>::v4 v4;
        v4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { hi[e] = (T)v[e]; lo[e] = (T)(v[e] - (float)hi[e]); }
        T* p = dst + (size_t)r * ld + c;
        *(v4*)p = hi;
        if (mode == 1) { *(v4*)(p + K_dst) = lo; *(v4*)(p + 2 * K_dst) = hi; }
        if (mode == 2) { *(v4*)(p + K_dst) = hi; *(v4*)(p + 2 * K_dst) = lo; }
        if (mode == 3) *(v4*)(p + K_dst) = lo;
    }
}
}  // namespace

int ofx_launch_pack_rows(const float* src, void* dst, int rows_src, int rows_dst, int K_src, int K_dst, int ld_src,
                         int mode, int op_dtype, hipStream_t s) {
    OFX_REQUIRE(K_dst % 4 == 0 && K_dst >= K_src && rows_dst >= rows_src, OFX_ESHAPE, "pack_rows: bad shape");
    size_t total = (size_t)rows_dst * (K_dst / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 16384) grid = 16384;
    if (grid < 1) grid = 1;
    if (op_dtype == OFX_F16)
        hipLaunchKernelGGL(pack_rows_kernel<f16_t>, dim3(grid), dim3(256), 0, s, src, (f16_t*)dst, rows_src, rows_dst, K_src, K_dst, ld_src, mode);
    else
        hipLaunchKernelGGL(pack_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, src, (bf16_t*)dst, rows_src, rows_dst, K_src, K_dst, ld_src, mode);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}

// ------------------------------------------------------------------------------------------------
// ViT patchify: pixels [N,3,224,224] fp32 -> patch matrix [N*49, 3072] operand type, column order
// (c, ky, kx) = Conv2d weight.reshape(768,-1) order (HF CLIPVisionEmbeddings.patch_embedding).
namespace {
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* px, T* out, int N, int img, int patch) {
    const int g = img / patch;                       // 7
    const int per_img = 3 * img * img / 8;           // 8-pixel groups per image
    const size_t total = (size_t)N * per_img;
    const int kdim = 3 * patch * patch;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / per_img);
        int e = (int)(i % per_img) * 8;              // element offset inside the image (c, y, x)
        const int c = e / (img * img); e -= c * img * img;
        const int y = e / img, x = e % img;
        const float* sp = px + (size_t)n * 3 * img * img + (size_t)c * img * img + (size_t)y * img + x;
        const f32x4 a = *(const f32x4*)sp, b = *(const f32x4*)(sp + 4);
        const int py = y / patch, ky = y % patch, pxi = x / patch, kx = x % patch;
        typename OpT<T>::v8 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = (T)a[k]; v[4 + k] = (T)b[k]; }
        *(typename OpT<T>::v8*)(out + ((size_t)n * g * g + py * g + pxi) * kdim + c * patch * patch + ky * patch + kx) = v;
    }
}

// x[n*50 + t] = LN( (t == 0 ? cls : patch_out[n*49 + t-1]) + pos[t] )  -> fp32 residual stream
// Optionally (LayerNorm folding) also the operand-type copy of the output row and its (mean, rstd): layer 0's LayerNorm-1 inputs.
template <int NCH, typename T>
__global__ __launch_bounds__(256) void vit_embed_ln_kernel(const float* patch_out, const float* cls, const float* pos,
                                                          const float* gamma, const float* beta, float* x, int N, int S,
                                                          float eps, T* xb, float* stat, T* xlo) {
    typedef typename OpT<T>::v4 v4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int D = NCH * 256, rows = N * S;
    for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
        const int n = r / S, t = r % S;
        const f32x4* sp = (const f32x4*)(t == 0 ? cls : patch_out + ((size_t)n * (S - 1) + t - 1) * D);
        const f32x4* pp = (const f32x4*)(pos + (size_t)t * D);
        f32x4 v[NCH];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            v[c] = sp[lane + 64 * c] + pp[lane + 64 * c];
            s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
        }
        const float mu = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            v[c] -= mu;
            q += v[c][0] * v[c][0] + v[c][1] * v[c][1] + v[c][2] * v[c][2] + v[c][3] * v[c][3];
        }
        const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
        float so = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = (lane + 64 * c) * 4;
            v[c] = v[c] * rstd * *(const f32x4*)(gamma + col) + *(const f32x4*)(beta + col);
            if (x) *(f32x4*)(x + (size_t)r * D + col) = v[c];          // NULL: the stream lives in (xb, xlo) only
            so += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
        }
        if (xb) {
            const float mo = wave_sum(so) * (1.0f / D);
            float qo = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int col = (lane + 64 * c) * 4;
                v4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float dlt = v[c][e] - mo; qo += dlt * dlt; o[e] = (T)v[c][e]; }
                *(v4*)(xb + (size_t)r * D + col) = o;
                if (xlo) {              // residual stream as a (hi, lo) operand-type pair: lo = x - hi
                    v4 l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) l[e] = (T)(v[c][e] - (float)o[e]);
                    *(v4*)(xlo + (size_t)r * D + col) = l;
                }
            }
            qo = wave_sum(qo);
            if (lane == 0) { stat[2 * (size_t)r] = mo; stat[2 * (size_t)r + 1] = rsqrtf(qo * (1.0f / D) + eps); }
        }
    }
}

// CLIP text embedding: x[i*Tc + t] = tok_emb[ids[i*T + t]] + pos_emb[t], t < Tc   (fp32, D = 512)
__global__ __launch_bounds__(256) void text_embed_kernel(const int64_t* ids, const float* tok, const float* pos, float* x,
                                                        int N, int T, int Tc, int D, int vocab) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rows = N * Tc;
    for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
        const int i = r / Tc, t = r % Tc;
        long long id = ids[(size_t)i * T + t];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const f32x4* tp = (const f32x4*)(tok + (size_t)id * D);
        const f32x4* pp = (const f32x4*)(pos + (size_t)t * D);
        for (int c = lane; c < D / 4; c += 64) *(f32x4*)(x + (size_t)r * D + c * 4) = tp[c] + pp[c];
    }
}

// pooled row index per text: i*Tc + min(eos_pos, Tc-1); eos_pos = argmax(ids) (legacy eos_id==2)
// or first position == eos_id (0 if none) — HF CLIPTextModel pooling rule.
__global__ void text_eos_index_kernel(const int64_t* ids, int* row_idx, int N, int T, int Tc, int eos_id) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int64_t* p = ids + (size_t)i * T;
    int pos = 0;
    if (eos_id == 2) {
        long long best = p[0];
        for (int t = 1; t < T; ++t) if (p[t] > best) { best = p[t]; pos = t; }
    } else {
        for (int t = 0; t < T; ++t) if (p[t] == eos_id) { pos = t; break; }
    }
    row_idx[i] = i * Tc + (pos < Tc ? pos : Tc - 1);
}

__global__ void iota_rows_kernel(int* idx, int n, int stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = i * stride;
}

// dst[r, col .. col+D) = src[r] / max(||src[r]||, eps)  (normalize != 0) else copy.   F.normalize.
__global__ __launch_bounds__(256) void l2norm_store_kernel(const float* src, float* dst, int rows, int D, int ld, int col,
                                                          int normalize) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
        const f32x4* sp = (const f32x4*)(src + (size_t)r * D);
        float q = 0.f;
        for (int c = lane; c < D / 4; c += 64) { f32x4 v = sp[c]; q += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
        const float inv = normalize ? 1.0f / fmaxf(sqrtf(wave_sum(q)), 1e-12f) : 1.0f;
        for (int c = lane; c < D / 4; c += 64) *(f32x4*)(dst + (size_t)r * ld + col + c * 4) = sp[c] * inv;
    }
}

// ---- set build: prefix token + pad-free compaction ------------------------------------------
// cu[b] = b + #unmasked items of outfits < b; one block, B <= 65536.
__global__ __launch_bounds__(1024) void set_offsets_kernel(const uint8_t* mask, int* cu, int B, int L) {
    __shared__ int part[1024];
    const int t = threadIdx.x, per = (B + 1023) / 1024;
    int cnt = 0;
    for (int b = t * per; b < min(B, (t + 1) * per); ++b) {
        int c = 1;
        for (int l = 0; l < L; ++l) c += mask[(size_t)b * L + l] == 0;
        cnt += c;
    }
    part[t] = cnt;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = t ? part[t - 1] : 0;
    for (int b = t * per; b < min(B, (t + 1) * per); ++b) {
        cu[b] = run;
        int c = 1;
        for (int l = 0; l < L; ++l) c += mask[(size_t)b * L + l] == 0;
        run += c;
    }
    if (t == 1023) cu[B] = part[1023];
}

// X[cu[b]] = prefix (shared or per outfit); X[cu[b]+1+j] = j-th unmasked item of outfit b.  One wave per (b, slot).
__global__ __launch_bounds__(256) void set_build_kernel(const float* x, const uint8_t* mask, const float* prefix,
                                                       int prefix_stride, const int* cu, float* X, int B, int L, int D) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int slots = B * (L + 1);
    for (int s = blockIdx.x * 4 + w; s < slots; s += gridDim.x * 4) {
        const int b = s / (L + 1), l = s % (L + 1) - 1;
        const float* src;
        int dst;
        if (l < 0) {
            src = prefix + (size_t)b * prefix_stride;
            dst = cu[b];
        } else {
            if (mask[(size_t)b * L + l]) continue;
            int k = 0;
            for (int j = 0; j < l; ++j) k += mask[(size_t)b * L + j] == 0;
            src = x + ((size_t)b * L + l) * D;
            dst = cu[b] + 1 + k;
        }
        for (int c = lane; c < D / 4; c += 64) *(f32x4*)(X + (size_t)dst * D + c * 4) = *(const f32x4*)(src + c * 4);
    }
}

// Indexed (varlen) set build: outfit b = prefix + rows item_index[cu_items[b] .. cu_items[b+1]) of a device-resident embedding
// table; writes the pad-free rows AND the row offsets cu_rows[b] = cu_items[b] + b (cu_rows[B] = live row count).
// One wave per outfit.  Out-of-range indices are clamped (the host processor validates them).
__global__ __launch_bounds__(256) void set_build_indexed_kernel(const float* table, int ld, long long n_table, const int* idx, const int* cu_items,
                                                               const float* prefix, int prefix_stride, int* cu_rows, float* X, int B, int D) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int b = blockIdx.x * 4 + w; b < B; b += gridDim.x * 4) {
        const int i0 = cu_items[b], n = cu_items[b + 1] - i0, r0 = i0 + b;
        if (lane == 0) { cu_rows[b] = r0; if (b == B - 1) cu_rows[B] = cu_items[B] + B; }
        const float* pp = prefix + (size_t)b * prefix_stride;
        for (int c = lane; c < D / 4; c += 64) *(f32x4*)(X + (size_t)r0 * D + c * 4) = *(const f32x4*)(pp + c * 4);
        for (int j = 0; j < n; ++j) {
            long long t = idx[i0 + j];
            t = t < 0 ? 0 : (t >= n_table ? n_table - 1 : t);
            const float* src = table + (size_t)t * ld;
            for (int c = lane; c < D / 4; c += 64) *(f32x4*)(X + (size_t)(r0 + 1 + j) * D + c * 4) = *(const f32x4*)(src + c * 4);
        }
    }
}

// dst[r] = src[idx[r]] for rows of row_bytes (multiple of 16)
__global__ __launch_bounds__(256) void gather_rows_kernel(const char* src, const int* idx, char* dst, int rows, int row_bytes, int src_ld_bytes) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
        const char* sp = src + (size_t)idx[r] * src_ld_bytes;
        for (int c = lane * 16; c < row_bytes; c += 64 * 16) *(f32x4*)(dst + (size_t)r * row_bytes + c) = *(const f32x4*)(sp + c);
    }
}

// dst[r] = hi[idx[r]] + lo[idx[r]] (fp32): pooled rows of a residual stream kept as an operand-type (hi, lo) pair
template <typename T>
__global__ __launch_bounds__(256) void gather_hilo_kernel(const T* hi, const T* lo, const int* idx, float* dst, int rows, int W) {
    typedef typename OpT<T>::v4 v4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
        const size_t so = (size_t)idx[r] * W;
        for (int c = lane * 4; c < W; c += 256) {
            const v4 h = *(const v4*)(hi + so + c), l = *(const v4*)(lo + so + c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (float)h[e] + (float)l[e];
            *(f32x4*)(dst + (size_t)r * W + c) = o;
        }
    }
}

// out[b] = X[cu[b]]
__global__ __launch_bounds__(256) void gather_row0_kernel(const float* X, const int* cu, float* out, int B, int D) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int b = blockIdx.x * 4 + w; b < B; b += gridDim.x * 4)
        for (int c = lane; c < D / 4; c += 64) *(f32x4*)(out + (size_t)b * D + c * 4) = *(const f32x4*)(X + (size_t)cu[b] * D + c * 4);
}

// CIR prefix token: out[b] = [img_emb (D/2) | text_emb[b] (D/2)]
__global__ __launch_bounds__(256) void cir_prefix_kernel(const float* img_emb, const float* txt, float* out, int B, int D) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, H = D / 2;
    for (int b = blockIdx.x * 4 + w; b < B; b += gridDim.x * 4)
        for (int c = lane; c < D / 4; c += 64) {
            const int col = c * 4;
            *(f32x4*)(out + (size_t)b * D + col) = col < H ? *(const f32x4*)(img_emb + col) : *(const f32x4*)(txt + (size_t)b * H + col - H);
        }
}

// ---- LayerNorm folding support (GemmArgs::row_stat / col_sum) ---------------------------------------------------------
// First layer of a tower: operand-type copy of the raw rows + their (mean, rstd).  One wave per row.
template <typename T>
__global__ __launch_bounds__(256) void row_stats_cast_kernel(const float* X, T* Xb, float* stat, int rows, int W, float eps, T* Xlo) {
    typedef typename OpT<T>::v4 v4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
        const float* xr = X + (size_t)r * W;
        float s = 0.f;
        for (int c = lane * 4; c < W; c += 256) { const f32x4 v = *(const f32x4*)(xr + c); s += (v[0] + v[1]) + (v[2] + v[3]); }
        const float mu = wave_sum(s) / W;
        float q = 0.f;
        for (int c = lane * 4; c < W; c += 256) {
            const f32x4 v = *(const f32x4*)(xr + c);
            v4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[e] - mu; q += d * d; o[e] = (T)v[e]; }
            *(v4*)(Xb + (size_t)r * W + c) = o;
            if (Xlo) {
                v4 l;
#pragma unroll
                for (int e = 0; e < 4; ++e) l[e] = (T)(v[e] - (float)o[e]);
                *(v4*)(Xlo + (size_t)r * W + c) = l;
            }
        }
        q = wave_sum(q);
        if (lane == 0) { stat[2 * (size_t)r] = mu; stat[2 * (size_t)r + 1] = rsqrtf(q / W + eps); }
    }
}
// (sum, sum of squares) per 64-column segment -> (mean, rstd) per row
__global__ __launch_bounds__(256) void stats_finalize_kernel(const float* part, int slots, int W, float eps, float* stat, int rows) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float* p = part + (size_t)r * slots * 2;
    float s = 0.f, q = 0.f;
    for (int k = 0; k < slots; ++k) { s += p[2 * k]; q += p[2 * k + 1]; }
    const float mu = s / W;
    const float var = fmaxf(q / W - mu * mu, 0.f);
    stat[2 * (size_t)r] = mu;
    stat[2 * (size_t)r + 1] = rsqrtf(var + eps);
}
// Pack time: W'[n,:] = round(W[n,:] * gamma), col_sum[n] = sum_k W'[n,k] (of the ROUNDED values), bias'[n] = bias[n] + sum_k beta[k] W[n,k].
// split != 0: row n of Wf is [hi(K) | lo(K)] (hi = round(W gamma), lo = round(W gamma - hi): GemmArgs::a_wrap) and col_sum sums hi + lo.
template <typename T>
__global__ __launch_bounds__(256) void fold_pack_kernel(const float* Wsrc, const float* gamma, const float* beta, const float* bias, T* Wf,
                                                       float* col_sum, float* bias_f, int N, int K, int split) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ld = split ? 2 * K : K;
    for (int n = blockIdx.x * 4 + w; n < N; n += gridDim.x * 4) {
        float cs = 0.f, bb = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float wv = Wsrc[(size_t)n * K + k];
            const float wg = wv * gamma[k];
            const T r = (T)wg;
            Wf[(size_t)n * ld + k] = r;
            cs += (float)r;
            if (split) {
                const T lo = (T)(wg - (float)r);
                Wf[(size_t)n * ld + K + k] = lo;
                cs += (float)lo;
            }
            bb += beta[k] * wv;
        }
        cs = wave_sum(cs); bb = wave_sum(bb);
        if (lane == 0) { col_sum[n] = cs; bias_f[n] = bias[n] + bb; }
    }
}

// table[i] = {src, dst, n floats}: block i copies entry i (pack time: every small fp32 tensor of a model in one launch)
// The table travels BY VALUE in the kernel arguments (<= 128 entries = 3 KiB per launch): no staging buffer, no host-side wait.
struct MultiCopyEntry { const float* src; float* dst; long long n; };
struct MultiCopyTable { MultiCopyEntry e[128]; };
__global__ __launch_bounds__(256) void multi_copy_kernel(const MultiCopyTable table) {
    const MultiCopyEntry e = table.e[blockIdx.x];          // blockIdx.y strides over the entry (the text tower's 101 MB token table is one entry)
    for (long long i = (long long)blockIdx.y * 256 + threadIdx.x; i < e.n; i += (long long)gridDim.y * 256) e.dst[i] = e.src[i];
}

// logits[b] = row0[b] . w + bias    (fp32, exact-order independent of B)
__global__ __launch_bounds__(256) void cp_head_kernel(const float* row0, const float* w, const float* bias, float* logits, int B, int D) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int b = blockIdx.x * 4 + wv; b < B; b += gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane; c < D / 4; c += 64) {
            const f32x4 a = *(const f32x4*)(row0 + (size_t)b * D + c * 4), k = *(const f32x4*)(w + c * 4);
            s += a[0] * k[0] + a[1] * k[1] + a[2] * k[2] + a[3] * k[3];
        }
        s = wave_sum(s);
        if (lane == 0) logits[b] = s + bias[0];
    }
}
}  // namespace

static inline int rows_grid(int rows) { int g = (rows + 3) / 4; return g > 8192 ? 8192 : (g < 1 ? 1 : g); }

int ofx_launch_patchify(const float* px, void* out, int N, int img, int patch, int op_dtype, hipStream_t s) {
    OFX_REQUIRE(img % patch == 0 && img % 8 == 0 && patch % 8 == 0, OFX_ESHAPE, "patchify: img=%d patch=%d", img, patch);
    size_t total = (size_t)N * 3 * img * img / 8;
    int grid = (int)((total + 255) / 256);
    if (grid > 32768) grid = 32768;
    ProfScope prof(PROF_OTHER, s);
    if (op_dtype == OFX_F16) hipLaunchKernelGGL(patchify_kernel<f16_t>, dim3(grid), dim3(256), 0, s, px, (f16_t*)out, N, img, patch);
    else hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, px, (bf16_t*)out, N, img, patch);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_vit_embed_ln(const float* patch_out, const float* cls, const float* pos, const float* g, const float* b,
                            float* x, int N, int S, int D, float eps, hipStream_t s, void* xb, float* stat, int op_dtype, void* xlo) {
    OFX_REQUIRE(D == 768 || D == 512 || D == 1024, OFX_ESHAPE, "vit_embed_ln: D=%d", D);
    const int grid = rows_grid(N * S);
    ProfScope prof(PROF_NORM, s);
#define VEL(NCH, T) hipLaunchKernelGGL((vit_embed_ln_kernel<NCH, T>), dim3(grid), dim3(256), 0, s, patch_out, cls, pos, g, b, x, N, S, eps, (T*)xb, stat, (T*)xlo)
    if (op_dtype == OFX_F16) { if (D == 768) VEL(3, f16_t); else if (D == 512) VEL(2, f16_t); else VEL(4, f16_t); }
    else { if (D == 768) VEL(3, bf16_t); else if (D == 512) VEL(2, bf16_t); else VEL(4, bf16_t); }
#undef VEL
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_text_embed(const int64_t* ids, const float* tok, const float* pos, float* x, int N, int T, int Tc, int D,
                          int vocab, hipStream_t s) {
    hipLaunchKernelGGL(text_embed_kernel, dim3(rows_grid(N * Tc)), dim3(256), 0, s, ids, tok, pos, x, N, T, Tc, D, vocab);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_text_eos_index(const int64_t* ids, int* row_idx, int N, int T, int Tc, int eos_id, hipStream_t s) {
    hipLaunchKernelGGL(text_eos_index_kernel, dim3((N + 255) / 256), dim3(256), 0, s, ids, row_idx, N, T, Tc, eos_id);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_iota_rows(int* idx, int n, int stride, hipStream_t s) {
    hipLaunchKernelGGL(iota_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, s, idx, n, stride);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_l2norm_store(const float* src, float* dst, int rows, int D, int ld, int col, int normalize, hipStream_t s) {
    OFX_REQUIRE(D % 4 == 0 && ld % 4 == 0 && col % 4 == 0, OFX_ESHAPE, "l2norm_store: D/ld/col must be multiples of 4");
    hipLaunchKernelGGL(l2norm_store_kernel, dim3(rows_grid(rows)), dim3(256), 0, s, src, dst, rows, D, ld, col, normalize);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_set_build(const float* x, const uint8_t* mask, const float* prefix, int prefix_stride, int* cu, float* X,
                         int B, int L, int D, hipStream_t s) {
    OFX_REQUIRE(B > 0 && B <= 65536 * 16 && L >= 0 && D % 4 == 0, OFX_ESHAPE, "set_build: B=%d L=%d D=%d", B, L, D);
    hipLaunchKernelGGL(set_offsets_kernel, dim3(1), dim3(1024), 0, s, mask, cu, B, L);
    hipLaunchKernelGGL(set_build_kernel, dim3(rows_grid(B * (L + 1))), dim3(256), 0, s, x, mask, prefix, prefix_stride, cu, X, B, L, D);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_set_build_indexed(const float* table, int ld, long long n_table, const int* idx, const int* cu_items, const float* prefix,
                                 int prefix_stride, int* cu_rows, float* X, int B, int D, hipStream_t s) {
    OFX_REQUIRE(B > 0 && D % 4 == 0 && ld % 4 == 0 && ld >= D && n_table > 0, OFX_ESHAPE, "set_build_indexed: B=%d D=%d ld=%d", B, D, ld);
    ProfScope prof(PROF_OTHER, s);
    hipLaunchKernelGGL(set_build_indexed_kernel, dim3(rows_grid(B)), dim3(256), 0, s, table, ld, n_table, idx, cu_items, prefix, prefix_stride, cu_rows, X, B, D);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_gather_rows(const void* src, const int* idx, void* dst, int rows, int row_bytes, int src_ld_bytes, hipStream_t s) {
    OFX_REQUIRE(row_bytes % 16 == 0 && src_ld_bytes % 16 == 0, OFX_ESHAPE, "gather_rows: row bytes must be multiples of 16");
    ProfScope prof(PROF_OTHER, s);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows_grid(rows)), dim3(256), 0, s, (const char*)src, idx, (char*)dst, rows, row_bytes, src_ld_bytes);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_gather_hilo(const void* hi, const void* lo, const int* idx, float* dst, int rows, int W, int op_dtype, hipStream_t s) {
    OFX_REQUIRE(W % 4 == 0 && rows > 0, OFX_ESHAPE, "gather_hilo: rows=%d W=%d", rows, W);
    if (op_dtype == OFX_F16) hipLaunchKernelGGL(gather_hilo_kernel<f16_t>, dim3(rows_grid(rows)), dim3(256), 0, s, (const f16_t*)hi, (const f16_t*)lo, idx, dst, rows, W);
    else hipLaunchKernelGGL(gather_hilo_kernel<bf16_t>, dim3(rows_grid(rows)), dim3(256), 0, s, (const bf16_t*)hi, (const bf16_t*)lo, idx, dst, rows, W);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_row_stats_cast(const float* X, void* Xb, float* stat, int rows, int W, float eps, int op_dtype, hipStream_t s, void* Xlo) {
    OFX_REQUIRE(W % 4 == 0 && rows > 0, OFX_ESHAPE, "row_stats_cast: rows=%d W=%d", rows, W);
    ProfScope prof(PROF_NORM, s);
    if (op_dtype == OFX_F16) hipLaunchKernelGGL(row_stats_cast_kernel<f16_t>, dim3(rows_grid(rows)), dim3(256), 0, s, X, (f16_t*)Xb, stat, rows, W, eps, (f16_t*)Xlo);
    else hipLaunchKernelGGL(row_stats_cast_kernel<bf16_t>, dim3(rows_grid(rows)), dim3(256), 0, s, X, (bf16_t*)Xb, stat, rows, W, eps, (bf16_t*)Xlo);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_stats_finalize(const float* part, int slots, int W, float eps, float* stat, int rows, hipStream_t s) {
    ProfScope prof(PROF_NORM, s);
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, part, slots, W, eps, stat, rows);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_fold_pack(const float* Wsrc, const float* gamma, const float* beta, const float* bias, void* Wf, float* col_sum, float* bias_f,
                         int N, int K, int op_dtype, hipStream_t s, int split) {
    if (op_dtype == OFX_F16) hipLaunchKernelGGL(fold_pack_kernel<f16_t>, dim3(rows_grid(N)), dim3(256), 0, s, Wsrc, gamma, beta, bias, (f16_t*)Wf, col_sum, bias_f, N, K, split);
    else hipLaunchKernelGGL(fold_pack_kernel<bf16_t>, dim3(rows_grid(N)), dim3(256), 0, s, Wsrc, gamma, beta, bias, (bf16_t*)Wf, col_sum, bias_f, N, K, split);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_multi_copy(const void* table_host, int n, hipStream_t s) {
    const MultiCopyEntry* src = (const MultiCopyEntry*)table_host;
    for (int i0 = 0; i0 < n; i0 += 128) {
        MultiCopyTable t;
        const int m = n - i0 < 128 ? n - i0 : 128;
        for (int i = 0; i < m; ++i) t.e[i] = src[i0 + i];
        hipLaunchKernelGGL(multi_copy_kernel, dim3(m, 64), dim3(256), 0, s, t);
    }
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
namespace {
__global__ __launch_bounds__(256) void fill_f32_kernel(float* p, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
}  // namespace
int ofx_launch_fill_f32(float* p, size_t n, float v, hipStream_t s) {
    const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(fill_f32_kernel, dim3(grid < 1 ? 1 : grid), dim3(256), 0, s, p, n, v);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_gather_row0(const float* X, const int* cu, float* out, int B, int D, hipStream_t s) {
    hipLaunchKernelGGL(gather_row0_kernel, dim3(rows_grid(B)), dim3(256), 0, s, X, cu, out, B, D);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_cir_prefix(const float* img_emb, const float* txt, float* out, int B, int D, hipStream_t s) {
    hipLaunchKernelGGL(cir_prefix_kernel, dim3(rows_grid(B)), dim3(256), 0, s, img_emb, txt, out, B, D);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
int ofx_launch_cp_head(const float* row0, const float* w, const float* bias, float* logits, int B, int D, hipStream_t s) {
    hipLaunchKernelGGL(cp_head_kernel, dim3(rows_grid(B)), dim3(256), 0, s, row0, w, bias, logits, B, D);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
