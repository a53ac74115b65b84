// Post-model scoring of the FITB / CIR tasks — index results must be
// bit-exact, so every distance is computed in fp32 (the reference's torch.cdist is on autocast's
// fp32 list) and ties are broken deterministically (smaller pool index first).
//
//  fitb_argmin : torch.cdist(y[B,1,D], cand[B,C,D]).squeeze(1).argmin(-1)
//                (reference src/trains/trainers/fill_in_the_blank_trainer.py:50-56)
//  l2_topk     : torch.cdist(Q,P) -> torch.topk(k, largest=False)
//                (reference src/trains/trainers/complementary_item_retrieval_trainer.py:240-249)
//                = row norms + fp32-MFMA (v_mfma_f32_32x32x2_f32, an exact fmaf chain) distance
//                tiles + per-query radix select of the k-th value + index-ordered collection.
//  topk_merge  : merge of per-shard candidate lists after the RCCL all-gather (pool row-sharding).
#include <algorithm>

#include "ofx_common.h"

namespace {

__global__ __launch_bounds__(256) void fitb_kernel(const float* y, const float* cand, int B, int C, int D, int64_t* idx, float* dist) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int b = blockIdx.x * 4 + w; b < B; b += gridDim.x * 4) {
        float best = INFINITY;
        int arg = 0;
        for (int c = 0; c < C; ++c) {
            const float* cp = cand + ((size_t)b * C + c) * D;
            float s = 0.f;
            for (int i = lane; i < D / 4; i += 64) {
                const f32x4 a = *(const f32x4*)(y + (size_t)b * D + i * 4), k = *(const f32x4*)(cp + i * 4);
                const f32x4 d = a - k;
                s += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
            }
            const float dd = sqrtf(wave_sum(s));
            if (dist && lane == 0) dist[(size_t)b * C + c] = dd;
            if (dd < best) { best = dd; arg = c; }          // strict <: first minimum wins (torch argmin)
        }
        if (lane == 0) idx[b] = arg;
    }
}

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* x, float* out, int rows, int D) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
        float s = 0.f;
        for (int i = lane; i < D / 4; i += 64) {
            const f32x4 a = *(const f32x4*)(x + (size_t)r * D + i * 4);
            s += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
        }
        s = wave_sum(s);
        if (lane == 0) out[r] = s;
    }
}

// dist[q, p] = sqrt(max(qn[q] + pn[p] - 2 q.p, 0)); tile 128 x 128, k-step 32, fp32 MFMA.
constexpr int DT = 128, DK = 32, DS = DK + 1;
__global__ __launch_bounds__(256) void dist_tile_kernel(const float* Q, const float* P, const float* qn, const float* pn,
                                                       float* dist, int nq, int np, int D, int ld) {
    __shared__ float As[DT * DS];
    __shared__ float Bs[DT * DS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int q0 = blockIdx.y * DT, p0 = blockIdx.x * DT;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int lrow = tid >> 3, lchk = (tid & 7) * 4;
    for (int k0 = 0; k0 < D; k0 += DK) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = lrow + 32 * it;
            int gq = q0 + r; gq = gq < nq ? gq : nq - 1;
            int gp = p0 + r; gp = gp < np ? gp : np - 1;
            const f32x4 a = *(const f32x4*)(Q + (size_t)gq * D + k0 + lchk);
            const f32x4 b = *(const f32x4*)(P + (size_t)gp * D + k0 + lchk);
#pragma unroll
            for (int e = 0; e < 4; ++e) { As[r * DS + lchk + e] = a[e]; Bs[r * DS + lchk + e] = b[e]; }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < DK / 2; ++kk) {
            const int kc = kk * 2 + (lane >> 5);
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[(wm * 64 + i * 32 + (lane & 31)) * DS + kc];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Bs[(wn * 64 + j * 32 + (lane & 31)) * DS + kc];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = p0 + wn * 64 + j * 32 + (lane & 31);
            const float pnv = p < np ? pn[p] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int q = q0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (q < nq && p < np) {
                    const float d2 = qn[q] + pnv - 2.0f * acc[i][j][e];
                    dist[(size_t)q * ld + p] = d2 > 0.f ? sqrtf(d2) : 0.f;   // never -0: bit order == value order
                }
            }
        }
}

// block-wide exclusive scan of one int per thread (256 threads); returns exclusive prefix, *total = sum
__device__ __forceinline__ int block_excl_scan(int v, int* sh /*[5]*/, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int s = sh[i]; if (i < w) base += s; tot += s; }
    *total = tot;
    return base + inc - v;
}

constexpr int KMAX = 128;

// sort KMAX (key = (dist bits << 32) | idx) ascending in LDS, 256 threads
__device__ __forceinline__ void bitonic_sort(unsigned long long* keys, int n /*power of two*/) {
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
        }
    __syncthreads();
}

// one block per query row: exact k smallest of row[0..np) (non-negative floats), ties -> smaller index
__global__ __launch_bounds__(256) void topk_select_kernel(const float* dist, int ld, int np, int k, int64_t index_base,
                                                         int64_t* idx_out, float* dist_out) {
    __shared__ int hist[2048];
    __shared__ int sh[8];
    __shared__ unsigned sel_prefix;
    __shared__ int sel_k;
    __shared__ unsigned long long keys[KMAX];
    const int tid = threadIdx.x;
    const unsigned* row = (const unsigned*)(dist + (size_t)blockIdx.x * ld);
    if (tid == 0) { sel_prefix = 0; sel_k = k; }
    // three radix passes: bits [31:21], [20:10], [9:0]
    const int shift[3] = {21, 10, 0}, nbins[3] = {2048, 2048, 1024};
    unsigned mask_hi = 0;
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        for (int i = tid; i < 2048; i += 256) hist[i] = 0;
        __syncthreads();
        const unsigned pref = sel_prefix;
        for (int i = tid; i < np; i += 256) {
            const unsigned v = row[i];
            if ((v & mask_hi) == pref) atomicAdd(&hist[(v >> shift[pass]) & (nbins[pass] - 1)], 1);
        }
        __syncthreads();
        // locate the bin holding the sel_k-th element: each thread owns 8 consecutive bins
        int loc[8], s = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) { loc[e] = hist[tid * 8 + e]; s += loc[e]; }
        int total;
        const int before = block_excl_scan(s, sh, &total);
        const int kk = sel_k;
        __syncthreads();
        if (kk > before && kk <= before + s) {
            int run = before;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (kk > run && kk <= run + loc[e]) {
                    sel_prefix = pref | ((unsigned)(tid * 8 + e) << shift[pass]);
                    sel_k = kk - run;
                }
                run += loc[e];
            }
        }
        mask_hi |= (unsigned)(nbins[pass] - 1) << shift[pass];
        __syncthreads();
    }
    const unsigned T = sel_prefix;          // bits of the k-th smallest value
    const int need_eq = sel_k;              // how many == T to take (smallest indices)
    const int c_less = k - need_eq;
    for (int i = tid; i < KMAX; i += 256) keys[i] = ~0ull;
    __syncthreads();
    int run_less = 0, run_eq = 0;
    for (int base = 0; base < np; base += 1024) {
        unsigned v[4];
        int nl = 0, ne = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = base + tid * 4 + e;
            v[e] = i < np ? row[i] : 0xffffffffu;
            nl += v[e] < T;
            ne += v[e] == T;
        }
        int tot;
        const int packed = block_excl_scan(nl | (ne << 16), sh, &tot);
        int pl = run_less + (packed & 0xffff), pe = run_eq + (packed >> 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned long long key = ((unsigned long long)v[e] << 32) | (unsigned)(base + tid * 4 + e);
            if (v[e] < T) { keys[pl++] = key; }
            else if (v[e] == T) { if (pe < need_eq) keys[c_less + pe] = key; ++pe; }
        }
        run_less += tot & 0xffff;
        run_eq += tot >> 16;
        if (run_less >= c_less && run_eq >= need_eq) break;     // block-uniform
    }
    bitonic_sort(keys, KMAX);
    for (int i = tid; i < k; i += 256) {
        const unsigned long long key = keys[i];
        idx_out[(size_t)blockIdx.x * k + i] = (int64_t)(key & 0xffffffffu) + index_base;
        dist_out[(size_t)blockIdx.x * k + i] = __uint_as_float((unsigned)(key >> 32));
    }
}

// merge parts x k sorted candidate lists per query (global indices) -> k best; ties -> smaller index
constexpr int MMAX = 1024;
__global__ __launch_bounds__(256) void topk_merge_kernel(const int64_t* idx_in, const float* dist_in, int parts, int nq, int k,
                                                        int64_t* idx, float* dist) {
    __shared__ unsigned long long keys[MMAX];
    const int q = blockIdx.x, n = parts * k;
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    // key = (dist bits, rank of global index among candidates is unknown) -> sort on (dist, idx) directly:
    // idx fits 40 bits in practice, but keep exactness by sorting (dist bits << 32 | slot) after a stable pre-order:
    // candidates are unique pool rows, so (dist, idx) pairs are unique; use two-key compare via 64-bit idx array.
    for (int i = threadIdx.x; i < n2; i += 256) {
        if (i < n) {
            const int p = i / k, j = i % k;
            const size_t off = ((size_t)p * nq + q) * k + j;
            keys[i] = ((unsigned long long)__float_as_uint(dist_in[off]) << 32) | (unsigned)i;
        } else keys[i] = ~0ull;
    }
    __syncthreads();
    // bitonic sort with (dist bits, global idx) comparison
    for (int kk = 2; kk <= n2; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < n2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    bool gt;
                    if ((a >> 32) != (b >> 32)) gt = (a >> 32) > (b >> 32);
                    else if (a == ~0ull || b == ~0ull) gt = a > b;
                    else {
                        const unsigned sa = (unsigned)a, sb = (unsigned)b;
                        const int64_t ia = idx_in[((size_t)(sa / k) * nq + q) * k + sa % k], ib = idx_in[((size_t)(sb / k) * nq + q) * k + sb % k];
                        gt = ia > ib;
                    }
                    const bool up = (i & kk) == 0;
                    if (gt == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
        }
    __syncthreads();
    for (int i = threadIdx.x; i < k; i += 256) {
        const unsigned s = (unsigned)keys[i];
        const size_t off = ((size_t)(s / k) * nq + q) * k + s % k;
        idx[(size_t)q * k + i] = idx_in[off];
        dist[(size_t)q * k + i] = dist_in[off];
    }
}

}  // namespace

int ofx_launch_fitb(const float* y, const float* cand, int B, int C, int D, int64_t* idx, float* dist, hipStream_t s) {
    int grid = (B + 3) / 4; if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(fitb_kernel, dim3(grid), dim3(256), 0, s, y, cand, B, C, D, idx, dist);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}

static inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }
size_t ofx_l2_topk_ws(int nq, int np) {
    const size_t ld = (size_t)(np + 3) / 4 * 4;
    return al256((size_t)nq * 4) + al256((size_t)np * 4) + al256((size_t)nq * ld * 4);
}

int ofx_launch_l2_topk(const float* Q, const float* P, int nq, int np, int D, int k, int64_t index_base, int64_t* idx,
                       float* dist, void* ws, size_t ws_bytes, hipStream_t s) {
    OFX_REQUIRE(Q && P && idx && dist && ws, OFX_EINVAL, "l2_topk: NULL argument");
    OFX_REQUIRE(nq > 0 && np > 0 && D > 0 && D % 32 == 0, OFX_ESHAPE, "l2_topk: nq=%d np=%d D=%d (D must be a multiple of 32)", nq, np, D);
    OFX_REQUIRE(k >= 1 && k <= KMAX && k <= np, OFX_ESHAPE, "l2_topk: k=%d must be in [1, min(%d, np)]", k, KMAX);
    OFX_REQUIRE(ws_bytes >= ofx_l2_topk_ws(nq, np), OFX_EWORKSPACE, "l2_topk: workspace %zu < %zu bytes", ws_bytes, ofx_l2_topk_ws(nq, np));
    const int ld = (np + 3) / 4 * 4;
    float* qn = (float*)ws;
    float* pn = (float*)((char*)ws + al256((size_t)nq * 4));
    float* dm = (float*)((char*)pn + al256((size_t)np * 4));
    hipLaunchKernelGGL(sqnorm_kernel, dim3(std::min((nq + 3) / 4, 4096)), dim3(256), 0, s, Q, qn, nq, D);
    hipLaunchKernelGGL(sqnorm_kernel, dim3(std::min((np + 3) / 4, 4096)), dim3(256), 0, s, P, pn, np, D);
    hipLaunchKernelGGL(dist_tile_kernel, dim3((np + DT - 1) / DT, (nq + DT - 1) / DT), dim3(256), 0, s, Q, P, qn, pn, dm, nq, np, D, ld);
    hipLaunchKernelGGL(topk_select_kernel, dim3(nq), dim3(256), 0, s, dm, ld, np, k, index_base, idx, dist);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}

int ofx_launch_topk_merge(const int64_t* idx_in, const float* dist_in, int parts, int nq, int k, int64_t* idx, float* dist,
                          hipStream_t s) {
    OFX_REQUIRE(idx_in && dist_in && idx && dist, OFX_EINVAL, "topk_merge: NULL argument");
    OFX_REQUIRE(parts >= 1 && nq >= 1 && k >= 1 && parts * k <= MMAX, OFX_ESHAPE, "topk_merge: parts*k=%d exceeds %d", parts * k, MMAX);
    hipLaunchKernelGGL(topk_merge_kernel, dim3(nq), dim3(256), 0, s, idx_in, dist_in, parts, nq, k, idx, dist);
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
