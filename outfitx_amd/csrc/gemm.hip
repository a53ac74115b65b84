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
#include "ofx_common.h"

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
    const int* m_dev;   // optional device-side row count (pad-free varlen sets); M is then the upper bound
    int M, N, K, lda, ldc, ldr, act, out_kind, tiles_n, nwg;
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
            v += bias4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (ACT == OFX_ACT_QUICK_GELU) v[e] = act_quick_gelu(v[e]);
                else if (ACT == OFX_ACT_GELU) v[e] = act_gelu(v[e]);
                else if (ACT == OFX_ACT_MISH) v[e] = act_mish(v[e]);
            }
            if (p.resid) v += *(const f32x4*)(p.resid + (size_t)gm * p.ldr + gn);
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
}

template <typename T>
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
    const int tm = bid / p.tiles_n, tn = bid % p.tiles_n;
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

    const int nk = p.K / BK;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            issue(kt + 1, cur ^ 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        OFX_LDS char* base = lds + cur * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chk = ((ks * 4 + fq) ^ fsw) * 16;
            v8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(OFX_LDS v8*)(base + a_frag + i * 16 * 128 + chk);
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *(OFX_LDS v8*)(base + w_frag + j * 16 * 128 + chk);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = OpT<T>::mfma16(wf[j], af[i], acc[i][j]);
        }
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
    switch (p.act) {
        case OFX_ACT_QUICK_GELU: epilogue<T, OFX_ACT_QUICK_GELU>(p, ep, m0 + wm * 64, n0 + wn * 64, lane); break;
        case OFX_ACT_GELU: epilogue<T, OFX_ACT_GELU>(p, ep, m0 + wm * 64, n0 + wn * 64, lane); break;
        case OFX_ACT_MISH: epilogue<T, OFX_ACT_MISH>(p, ep, m0 + wm * 64, n0 + wn * 64, lane); break;
        default: epilogue<T, OFX_ACT_NONE>(p, ep, m0 + wm * 64, n0 + wn * 64, lane); break;
    }
}

}  // namespace

int ofx_launch_gemm(const GemmArgs& g, int op_dtype, hipStream_t s) {
    OFX_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, OFX_ESHAPE, "gemm: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
    OFX_REQUIRE(g.N % BN == 0, OFX_ESHAPE, "gemm: N=%d must be a multiple of %d (pad the weight at pack time)", g.N, BN);
    OFX_REQUIRE(g.K % BK == 0, OFX_ESHAPE, "gemm: K=%d must be a multiple of %d", g.K, BK);
    OFX_REQUIRE(g.lda >= g.K && g.lda % 8 == 0, OFX_ESHAPE, "gemm: lda=%d must be >= K and a multiple of 8", g.lda);
    OFX_REQUIRE(g.ldc % 4 == 0 && g.ldc >= (g.out_kind == 2 ? 3 * g.N : g.N), OFX_ESHAPE, "gemm: bad ldc=%d", g.ldc);
    OFX_REQUIRE(!g.resid || (g.ldr % 4 == 0 && g.ldr >= g.N), OFX_ESHAPE, "gemm: bad ldr=%d", g.ldr);
    OFX_REQUIRE(((uintptr_t)g.A % 16 == 0) && ((uintptr_t)g.W % 16 == 0) && ((uintptr_t)g.C % 16 == 0), OFX_EINVAL,
                "gemm: operands must be 16-byte aligned");
    OFX_REQUIRE(op_dtype == OFX_BF16 || op_dtype == OFX_F16, OFX_EINVAL, "gemm: operand dtype must be bf16 or f16");
    KArgs k;
    k.A = (const char*)g.A; k.W = (const char*)g.W; k.C = (char*)g.C; k.bias = g.bias; k.resid = g.resid; k.m_dev = g.m_dev;
    k.M = g.M; k.N = g.N; k.K = g.K; k.lda = g.lda; k.ldc = g.ldc; k.ldr = g.ldr; k.act = g.act; k.out_kind = g.out_kind;
    k.tiles_n = g.N / BN;
    const int tiles_m = (g.M + BM - 1) / BM;
    k.nwg = tiles_m * k.tiles_n;
    static bool attr_set = false;
    if (!attr_set) {
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_128x128_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
        OFX_HIP(hipFuncSetAttribute((const void*)gemm_128x128_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
        attr_set = true;
    }
    ProfScope prof(PROF_GEMM, s, 2.0 * g.M * g.N * g.K);
    if (op_dtype == OFX_BF16)
        hipLaunchKernelGGL(gemm_128x128_kernel<bf16_t>, dim3(k.nwg), dim3(256), GEMM_LDS_BYTES, s, k);
    else
        hipLaunchKernelGGL(gemm_128x128_kernel<f16_t>, dim3(k.nwg), dim3(256), GEMM_LDS_BYTES, s, k);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
