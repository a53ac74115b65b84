// Internal helpers shared by the HIP translation units of libofx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <mutex>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ofx.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define OFX_LDS __attribute__((address_space(3)))
#define OFX_GLB __attribute__((address_space(1)))

// thread-local last-error text (ofx_last_error)
void ofx_set_error(const char* fmt, ...);

#define OFX_REQUIRE(cond, code, ...)            \
    do {                                        \
        if (!(cond)) {                          \
            ofx_set_error(__VA_ARGS__);         \
            return (code);                      \
        }                                       \
    } while (0)

#define OFX_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            ofx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return OFX_EHIP;                                                           \
        }                                                                              \
    } while (0)

#define OFX_LAUNCH_CHECK()                                                             \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            ofx_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return OFX_EHIP;                                                           \
        }                                                                              \
    } while (0)

// ---- operand-type traits (bf16 / f16 share every fragment shape and MFMA rate) ----
template <typename T> struct OpT;
template <> struct OpT<bf16_t> {
    typedef bf16x8 v8;
    typedef bf16x4 v4;
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct OpT<f16_t> {
    typedef f16x8 v8;
    typedef f16x4 v4;
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

// ---- activations (fp32) ----
// v_rcp_f32 (1 ulp) instead of an IEEE division: results feed a bf16/f16 operand or are compared at 1e-5
__device__ __forceinline__ float act_quick_gelu(float u) { return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * u)); }   // exp(-1.702 u) = 2^(-1.702 log2(e) u): one multiply, one v_exp_f32
__device__ __forceinline__ float act_gelu(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float act_mish(float u) {
    // u * tanh(softplus(u)); softplus threshold 20 as torch.  tanh(log(1+e^u)) = ((1+e^u)^2-1)/((1+e^u)^2+1)
    if (u > 20.0f) return u;
    float e = __expf(u);
    float n = e * (e + 2.0f);
    return u * (n * __builtin_amdgcn_rcpf(n + 2.0f));
}
// d/du [u * tanh(softplus(u))] = tanh(sp) + u * sigmoid(u) * (1 - tanh(sp)^2)
__device__ __forceinline__ float act_mish_grad(float u) {
    if (u > 20.0f) return 1.0f;
    const float e = __expf(u), n = e * (e + 2.0f);
    const float t = n * __builtin_amdgcn_rcpf(n + 2.0f);               // tanh(softplus(u))
    const float sg = e * __builtin_amdgcn_rcpf(1.0f + e);              // sigmoid(u)
    return t + u * sg * (1.0f - t * t);
}
__device__ __forceinline__ float apply_act(float u, int act) {
    switch (act) {
        case OFX_ACT_QUICK_GELU: return act_quick_gelu(u);
        case OFX_ACT_GELU: return act_gelu(u);
        case OFX_ACT_MISH: return act_mish(u);
        default: return u;
    }
}

// sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15); the total is valid in lane 15 of the row
__device__ __forceinline__ float row16_sum_to_lane15(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));   // row_shr:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));   // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));   // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));   // row_shr:8
    return v;
}
// sum over aligned groups of 8 lanes (xor 1, xor 2 within quads, then the mirrored half row); every lane of the group gets the total
__device__ __forceinline__ float row8_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm:[1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm:[2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- optional HIP-event profiling of launches by category (bench.py's live roofline) ----
enum { PROF_GEMM = 0, PROF_NORM = 1, PROF_ATTN = 2, PROF_OTHER = 3, PROF_NCAT = 4 };
extern bool g_ofx_prof_on;
extern int g_ofx_prof_mask;   // bit per category
void ofx_prof_begin(int cat, hipStream_t s, double flops);
void ofx_prof_end(hipStream_t s);
void ofx_prof_set_tag(int M, int N, int K, int kind, int kmul, double bytes);      // labels the NEXT record (GEMM dispatcher: shape, kernel kind, algorithmic HBM bytes)
// ext = true: nothing is recorded on the stream; the scope's launches carry the two events themselves (OFX_PLAUNCH ->
// hipExtLaunchKernelGGL start / stop events: the timestamps come from the dispatch packet's completion signal, so no marker
// packets are put between the kernels).  The first launch of the scope takes the start event, the one flagged `last` the stop.
// Profiling state is process-global and single-threaded by contract (benchmarks / tests only, include/ofx.h).
bool ofx_prof_ext_begin(int cat, double flops);
void ofx_prof_ext_end();
extern hipEvent_t g_ofx_launch_e0, g_ofx_launch_e1;
struct ProfScope {
    hipStream_t s; bool on, ext;
    ProfScope(int cat, hipStream_t st, double flops = 0.0, bool ext_ = false, bool enable = true) : s(st), on(enable && g_ofx_prof_on && ((g_ofx_prof_mask >> cat) & 1)), ext(ext_) {
        if (on && ext) on = ofx_prof_ext_begin(cat, flops);
        else if (on) ofx_prof_begin(cat, s, flops);
    }
    ~ProfScope() {
        if (on && ext) ofx_prof_ext_end();
        else if (on) ofx_prof_end(s);
    }
};
#define OFX_PLAUNCH(last, kernel, grid, block, lds, stream, ...)                                                                  \
    do {                                                                                                                          \
        hipEvent_t pe0_ = g_ofx_launch_e0, pe1_ = (last) ? g_ofx_launch_e1 : nullptr;                                             \
        g_ofx_launch_e0 = nullptr;                                                                                                \
        if (last) g_ofx_launch_e1 = nullptr;                                                                                      \
        if (pe0_ || pe1_) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, pe0_, pe1_, 0, __VA_ARGS__);                      \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                                   \
    } while (0)

// ---- internal launchers (defined in the .hip files, used by api.hip) ----
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int pad128(int v) { return (v + 127) / 128 * 128; }

// hipFuncSetAttribute applies to the CURRENT device: raise a kernel's dynamic-LDS limit once per device (a process may hold
// handles on several devices, and several host threads may launch concurrently).
struct DeviceOnce {
    std::mutex mu;
    unsigned long long done = 0;
    template <typename F>
    int run(F&& f) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        const unsigned long long bit = 1ull << (dev & 63);
        std::lock_guard<std::mutex> lk(mu);
        if (done & bit) return OFX_OK;
        const int rc = f();
        if (rc == OFX_OK) done |= bit;
        return rc;
    }
};

struct Bump {                                       // carve a caller-provided workspace
    char* base; size_t cap, off = 0; bool ok = true;
    Bump(void* p, size_t c) : base((char*)p), cap(c) {}
    template <typename T> T* take(size_t n) {
        off = align_up(off, 256);
        T* r = (T*)(base + off);
        off += n * sizeof(T);
        if (off > cap) ok = false;
        return r;
    }
};
#define TRY(x) do { int rc_ = (x); if (rc_ != OFX_OK) return rc_; } while (0)

// ---- dropout (training step): stateless masks.  Element (row, col) of dropout site `site` is kept iff
// hash(seed, site, row, col) >= thresh, thresh = p * 2^32; kept values are scaled by 1 / (1 - p).  Forward and backward
// recompute the same mask from the same counters, so no mask is ever stored.  thresh == 0 means "no dropout".
struct DropArgs {
    unsigned seed = 0, site = 0, thresh = 0;
    float scale = 1.0f;
};
__host__ __device__ __forceinline__ unsigned ofx_fmix32(unsigned h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ float drop_mul(const DropArgs& d, unsigned row, unsigned col) {
    unsigned h = ofx_fmix32(d.seed + d.site * 0x9E3779B9u + row * 0x632BE5ABu);
    h = ofx_fmix32(h ^ (col * 0x9E3779B1u + 0x7F4A7C15u));
    return h >= d.thresh ? d.scale : 0.0f;
}
inline DropArgs make_drop(float p, unsigned seed, unsigned site) {
    DropArgs d;
    if (p > 0.f) { d.seed = seed; d.site = site; d.thresh = (unsigned)((double)p * 4294967296.0); d.scale = 1.0f / (1.0f - p); }
    return d;
}

struct GemmArgs {
    const void* A;      // [M, lda] operand type, K-contiguous
    const void* W;      // [N, K] operand type, K-contiguous (torch Linear layout)
    void* C;            // output, see out_kind
    const float* bias;  // [N] or null
    const float* resid; // fp32 [M, ldr] or null; may alias C when out_kind == F32
    const int* m_dev = nullptr; // optional device-side live row count (<= M)
    float* aux_out = nullptr;   // optional fp32 [M, N] copy of (acc + bias) BEFORE the activation (training tape)
    void* slab = nullptr;       // optional split-K scratch (fp32 [splits, M, N]); see ofx_gemm_splitk_bytes
    size_t slab_bytes = 0;
    int M, N, K, lda, ldc, ldr;
    int act;            // ofx_act
    int out_kind;       // 0 fp32 | 1 operand type | 2 split3 (hi|lo|hi at column blocks of width N, ldc >= 3N; bf16 only)
    DropArgs drop;      // applied to act(acc + bias) BEFORE the residual add (MISH_GRAD: to acc before the mish' factor)
    // ---- LayerNorm folding (CLIP towers): the LayerNorm between a residual-stream producer and the next linear layer is
    // never materialised.  Producer side (fp32 output): also store the operand-type copy of the output row and, per row
    // and per 64-column segment, (sum, sum of squares) -> stat_part[row][segment] (float2).  Consumer side: A is that raw
    // copy, W is pre-scaled by gamma, and the epilogue applies  rstd[row] * (acc - mean[row] * col_sum[n]) + bias[n].
    void* xb_out = nullptr;         // [M, N] operand type
    float* stat_part = nullptr;     // [M, N / 64, 2]
    const float* row_stat = nullptr;   // [M, 2] (mean, rstd)
    int stat_ld = 1;                   // row m's statistics sit at row_stat[2 * m * stat_ld] (strided A rows)
    void* xlo = nullptr;               // producer with xb_out: the residual stream is the operand-type pair (xb_out, xlo), updated in place;
                                       // `resid` / `C` are then unused (no fp32 copy of the stream exists)
    const float* col_sum = nullptr;    // [N] column sums of the (rounded) gamma-scaled weight rows
    // ---- split weights against ONE copy of A ("W2"): W = [N, K] with K = 2 a_wrap, row n = [hi(a_wrap) | lo(a_wrap)]; A is
    // [M, a_wrap] and its k index wraps, so C = A . (hi + lo)^T: ~22 significant weight bits for two MFMA products
    int a_wrap = 0;
    // fp8 correction product (gemm_w2f8.hip): with a_wrap, the e4m3 copy of the rows' lo halves [N, a_wrap] bytes and its per-row scale
    // bytes (ofx_launch_pack_lo8); when the dual-weight 256x256 kernel would run, its fp8 variant runs instead (f16 only, a_wrap % 128 == 0)
    const void* W8 = nullptr; const void* w8_scale = nullptr;
    int k_mult = 1;     // informational (profile records): K = k_mult x the logical depth (3: three-product K-concatenation; a_wrap implies 2)
    // ---- small batches (split-K plans): let the NEXT kernel do the second pass, one launch less per linear layer.
    // defer_splits: when the plan splits K, the reduce pass is skipped and *defer_splits = number of slabs (fp32 [splits, M, N] at
    // `slab`, bias / activation / residual NOT applied: the consumer sums them); left at 1 when the output was written normally.
    int* defer_splits = nullptr;
    // ln_gamma: when the plan splits K (fp32 output, no activation), the reduce pass also emits LayerNorm(C) * gamma + beta to
    // ln_out (ln_kind: 1 operand type | 2 split3, row stride ln_ld) and sets *ln_done; otherwise the caller runs its LayerNorm.
    const float* ln_gamma = nullptr; const float* ln_beta = nullptr; void* ln_out = nullptr; int ln_ld = 0, ln_kind = 0; float ln_eps = 0.f;
    bool* ln_done = nullptr;
};
int ofx_launch_gemm(const GemmArgs& g, int op_dtype /*OFX_BF16|OFX_F16*/, hipStream_t s);
int ofx_gemm_splitk_plan(int M, int N, int K);
size_t ofx_gemm_splitk_bytes(int M, int N, int K);

struct LnArgs {
    const float* x;        // [rows_in, D] fp32
    const int* row_idx;    // optional gather: output row r reads x[row_idx[r]]
    const float* gamma;    // [D]
    const float* beta;     // [D]
    void* y;               // [rows, ldy]
    int rows, D, ldy;
    int out_kind;          // 0 fp32 | 1 operand type | 2 split3
    float eps;
    float* stats = nullptr;  // optional [rows, 2] (mean, rstd) for the backward pass
};
int ofx_gemm_tn_splits(int M, int N, int K);
size_t ofx_gemm_tn_slab_bytes(int M, int N, int K);
int ofx_launch_gemm_tn(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K, const int* k_dev,
                       void* slab, size_t slab_bytes, int op_dtype, hipStream_t s, int m_valid = 0, int n_valid = 0, int accumulate = 0);
int ofx_launch_layernorm(const LnArgs& a, int op_dtype, hipStream_t s);
struct SplitKLnArgs {       // split-K second pass + LayerNorm in one launch (norm_act.hip)
    const float* slab; size_t plane; int splits;       // fp32 [splits][rows, D]
    const float* bias; const float* resid; int ldr;    // optional [D]; optional fp32 [rows, ldr] (may alias x)
    float* x; int ldx;                                 // fp32 [rows, ldx]: the reduced rows
    const float* gamma; const float* beta; void* y; int ldy, out_kind; float eps;
    int rows, D; const int* m_dev;
};
int ofx_launch_splitk_reduce_ln(const SplitKLnArgs& a, int op_dtype, hipStream_t s, bool in_gemm_scope = false);   // in_gemm_scope: the launch closes the GEMM dispatcher's profile record
int ofx_launch_layernorm_dev(const LnArgs& a, const int* rows_dev, int op_dtype, hipStream_t s);
// fp8 (e4m3) copy of the lo half of split-weight rows [hi | lo] (f16) + per-row E8M0 scale bytes, in gemm_w2f8.hip's layouts
int ofx_launch_pack_lo8(const void* w2_rows, void* dst8, void* scale8, int N, int K, hipStream_t s);
int ofx_launch_pack_rows(const float* src, void* dst, int rows_src, int rows_dst, int K_src, int K_dst, int ld_src,
                         int mode, int op_dtype, hipStream_t s);
int ofx_launch_patchify(const float* px, void* out, int N, int img, int patch, int op_dtype, hipStream_t s);
int ofx_launch_vit_embed_ln(const float* patch_out, const float* cls, const float* pos, const float* g, const float* b,
                            float* x, int N, int S, int D, float eps, hipStream_t s, void* xb = nullptr, float* stat = nullptr, int op_dtype = OFX_BF16,
                            void* xlo = nullptr);
int ofx_launch_text_embed(const int64_t* ids, const float* tok, const float* pos, float* x, int N, int T, int Tc, int D,
                          int vocab, hipStream_t s);
int ofx_launch_text_eos_index(const int64_t* ids, int* row_idx, int N, int T, int Tc, int eos_id, hipStream_t s);
int ofx_launch_iota_rows(int* idx, int n, int stride, hipStream_t s);
int ofx_launch_l2norm_store(const float* src, float* dst, int rows, int D, int ld, int col, int normalize, hipStream_t s);
int ofx_launch_set_build(const float* x, const uint8_t* mask, const float* prefix, int prefix_stride, int* cu, float* X,
                         int B, int L, int D, hipStream_t s);
int ofx_launch_gather_rows(const void* src, const int* idx, void* dst, int rows, int row_bytes, int src_ld_bytes, hipStream_t s);
int ofx_launch_set_build_indexed(const float* table, int ld, long long n_table, const int* idx, const int* cu_items, const float* prefix,
                                 int prefix_stride, int* cu_rows, float* X, int B, int D, hipStream_t s);
int ofx_preprocess_to(const uint8_t* src, const long long* offsets, const int* heights, const int* widths, int N, int channels, int size,
                      const float* mean, const float* stdv, float* out, void* patches, int patch, int op_dtype, void* ws, size_t ws_bytes, hipStream_t stream);
int ofx_launch_row_stats_cast(const float* X, void* Xb, float* stat, int rows, int W, float eps, int op_dtype, hipStream_t s, void* Xlo = nullptr);
int ofx_launch_gather_hilo(const void* hi, const void* lo, const int* idx, float* dst, int rows, int W, int op_dtype, hipStream_t s);
int ofx_launch_stats_finalize(const float* part, int slots, int W, float eps, float* stat, int rows, hipStream_t s);
int ofx_launch_fold_pack(const float* Wsrc, const float* gamma, const float* beta, const float* bias, void* Wf, float* col_sum, float* bias_f,
                         int N, int K, int op_dtype, hipStream_t s, int split = 0);
int ofx_launch_gather_row0(const float* X, const int* cu, float* out, int B, int D, hipStream_t s);
int ofx_launch_cir_prefix(const float* img_emb, const float* txt, float* out, int B, int D, hipStream_t s);
int ofx_launch_cp_head(const float* row0, const float* w, const float* bias, float* logits, int B, int D, hipStream_t s);

struct AttnArgs {
    const void* qkv;          // [nseq*seq_len, ld] operand type: q at col 0, k at col k_off, v at col v_off; head h at +h*64
    void* out;                // [nseq*seq_len, ldo] operand type
    const int64_t* key_mask;  // optional [nseq, mask_ld]: 0 = key ignored (HF attention_mask)
    int nseq, seq_len, n_head, ld, ldo, k_off, v_off, mask_ld;
    int causal;
    float scale;
    // varlen mode (outfit sets, training): sequence b = rows cu_seqlens[b] .. cu_seqlens[b+1] (at most seq_len <= 32 of them)
    const int* cu_seqlens = nullptr;
    int only_row0 = 0;
    DropArgs drop;
    int split3_w = 0;         // > 0: out rows are [hi(W) | lo(W) | hi(W)], W = split3_w, ldo >= 3 W (A operand of a three-product GEMM)
};
int ofx_launch_attention_mfma(const AttnArgs& a, int op_dtype, hipStream_t s);

struct SetAttnArgs {
    const void* qkv;       // [rows, 3D] (q|k|v): fp32, or the operand type when qkv_op (training tape)
    void* out;             // [rows, ldo]: out_kind 0 fp32 | 1 op | 2 split3
    const int* cu_seqlens; // [nseq+1] device row offsets
    int nseq, n_head, D, ldo, out_kind;
    int max_len;           // upper bound of 1 + items (<= 32)
    int only_row0;         // compute query row 0 of every set only (last layer)
    float scale;
    DropArgs drop;         // attention-probability dropout (row = set * n_head + head, col = query * 32 + key)
    int qkv_op = 0;
    // fixed-length mode (cu_seqlens == nullptr): every sequence has fixed_len <= 32 rows; optional causal AND key-padding mask
    int fixed_len = 0, causal = 0, mask_ld = 0;
    const int64_t* key_mask = nullptr;
    // fp32 qkv still in split-K slabs (GemmArgs::defer_splits): qkv = slab 0, element = sum_s qkv[s * plane + ...] + bias[col]
    int splits = 1; size_t plane = 0; const float* bias = nullptr;
};
int ofx_launch_set_attention(const SetAttnArgs& a, int op_dtype, hipStream_t s);
