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

// dist[q, p] = sqrt(max(qn[q] + pn[p] - 2 q.p, 0)) on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32: an exact fmaf chain, so a
// (q, p) pair's distance has the same bits whichever tile, shard or launch computes it - what the sharded == unsharded merge relies on).
//
// Round 4 rewrite (the round-3 kernel staged each k-step with synchronous loads + 32 scalar LDS stores per thread and read its
// fragments as 4-byte words: 66 TF/s with the select behind it).  Now: tile 128 queries x 128 pool rows per 256-thread block (4 waves
// as 2 x 2 of 64 x 64), k-step 32 floats = 128-byte rows, both operands by LDS-DMA (global_load_lds_dwordx4, 8 KiB-row pieces of 8 rows
// x 128 B, the 16-byte chunk index XOR-swizzled on the SOURCE side and again on the read side: conflict-free ds_read_b128), two stages,
// counted vmcnt + raw s_barrier, two blocks per CU so that one block's epilogue runs beside the other's MFMAs.  One ds_read_b128 per
// lane feeds FOUR MFMAs: lanes 0-31 hold A[row][8 t + 0..3], lanes 32-63 A[row][8 t + 4..7], and MFMA jj of the group multiplies
// component jj - the k index runs 8 t + jj, 8 t + 4 + jj inside one instruction: a fixed permutation of the summation order, the same
// for every (q, p), every tile and every shard (the order itself is free: the reference's own sgemm order is unspecified).
// Persistent over pool tiles: block (x, y) walks tiles x, x + gridDim.x, ... of query panel y (its Q panel stays L2-resident).
// FILTER = false: the tile goes to the distance matrix (small pools, the sample pass, the fallback).
// FILTER = true : nothing is written but the (distance, row) pairs with distance <= tau[q] - appended to the query's candidate list
//                 (the k-th smallest distance of the SAMPLE rows [0, S) bounds the k-th smallest of the whole pool from above, so every
//                 global top-k member outside the sample passes the filter): the 400 MB distance matrix of a 1000 x 100k problem never
//                 exists, and the select no longer reads it three times.
constexpr int DT = 128, DK = 32;
constexpr int DIST_STAGE = 2 * DT * DK * 4;            // A (16 KiB) + B (16 KiB)
constexpr int DIST_LDS = 2 * DIST_STAGE;               // 64 KiB: two blocks per CU

__device__ __forceinline__ void dist_glds16(const char* g, OFX_LDS char* l) {
    __builtin_amdgcn_global_load_lds((const OFX_GLB void*)g, (OFX_LDS void*)l, 16, 0, 0);
}

template <bool FILTER>
__global__ __launch_bounds__(256, 2) void dist_mfma_kernel(const float* Q, const float* P, const float* qn, const float* pn, float* dist, int nq, int np, int D, int ld,
                                                          int p_begin, const float* tau, int tau_ld, int* cnt, unsigned long long* cand, int cap, const int* only_flagged, int tiles_q) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    OFX_LDS char* lds = (OFX_LDS char*)dsm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    // 1-D grid, query panel = block id mod tiles_q: with 8 panels (1000 queries) every XCD (blocks b, b + 8, ... share one) serves ONE panel, whose
    // 512 KiB of queries stay in that XCD's L2 while the pool tiles stream through
    const int q0 = ((int)blockIdx.x % tiles_q) * DT, bx = (int)blockIdx.x / tiles_q, gx = (int)gridDim.x / tiles_q;
    if (only_flagged) {                                  // fallback launch: only query panels that hold a flagged query do anything
        const int f = (tid < DT && q0 + tid < nq) ? only_flagged[q0 + tid] : 0;
        if (!__syncthreads_or(f)) return;
    }
    const int tiles_p = (np - p_begin + DT - 1) / DT;
    // LDS-DMA addressing: wave w moves rows (w * 4 + i) * 8 + (lane >> 3), i = 0..3, of each operand; physical 16-byte slot lane & 7 of a
    // row holds logical chunk (lane & 7) ^ ((row >> 1) & 7)
    const int lrow = lane >> 3, lslot = lane & 7;
    const char* a_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + lrow;
        int gq = q0 + row; gq = gq < nq ? gq : nq - 1;
        a_src[i] = (const char*)(Q + (size_t)gq * D) + ((lslot ^ ((row >> 1) & 7)) << 4);
    }
    const int dst = wave * 4 * 1024;                     // + i * 1024 (+ DT * DK * 4 for B); the DMA adds lane * 16
    // fragment reads: row = tile row of lane & 31, logical chunk 2 t + (lane >> 5)
    const int fr = lane & 31, fh = lane >> 5;
    int a_frag[2], b_frag[2], a_sw[2], b_sw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + fr, rb = wn * 64 + i * 32 + fr;
        a_frag[i] = ra * 128; a_sw[i] = (ra >> 1) & 7;
        b_frag[i] = DT * DK * 4 + rb * 128; b_sw[i] = (rb >> 1) & 7;
    }
    const int nk = D / DK;
    for (int tp = bx; tp < tiles_p; tp += gx) {
        const int p0 = p_begin + tp * DT;
        const char* b_src[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 8 + lrow;
            int gp = p0 + row; gp = gp < np ? gp : np - 1;
            b_src[i] = (const char*)(P + (size_t)gp * D) + ((lslot ^ ((row >> 1) & 7)) << 4);
        }
        auto issue = [&](int kt, int stage) {
            OFX_LDS char* base = lds + stage * DIST_STAGE + dst;
            const size_t koff = (size_t)kt * DK * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) dist_glds16(a_src[i] + koff, base + i * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) dist_glds16(b_src[i] + koff, base + DT * DK * 4 + i * 1024);
        };
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        issue(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) { issue(kt + 1, cur ^ 1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // every wave's pieces of stage `cur` have landed
            OFX_LDS char* st = lds + cur * DIST_STAGE;
            f32x4 af[4][2], bf[4][2];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[t][i] = *(OFX_LDS f32x4*)(st + a_frag[i] + (((2 * t + fh) ^ a_sw[i]) << 4));
                    bf[t][i] = *(OFX_LDS f32x4*)(st + b_frag[i] + (((2 * t + fh) ^ b_sw[i]) << 4));
                }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t][i][jj], bf[t][j][jj], acc[i][j], 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // stage `cur` may be refilled (by the issue of iteration kt + 1)
        }
        // epilogue: acc[i][j][e] = q . p for q = q0 + wm*64 + i*32 + (e & 3) + 8 (e >> 2) + 4 (lane >> 5), p = p0 + wn*64 + j*32 + (lane & 31)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = p0 + wn * 64 + j * 32 + (lane & 31);
                const float pnv = p < np ? pn[p] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int q = q0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    asm volatile("" : "+v"(q));          // opaque: the per-query loads below stay inside this element's scope (hoisting all 32 queries' norms and thresholds above the element loops spilled)
                    if (q < nq && p < np) {
                        const float d2 = qn[q] + pnv - 2.0f * acc[i][j][e];
                        const float d = d2 > 0.f ? sqrtf(d2) : 0.f;                 // never -0: bit order == value order
                        if (!FILTER) dist[(size_t)q * ld + (p - p_begin)] = d;
                        else if (d <= tau[(size_t)q * tau_ld]) {
                            const int pos = atomicAdd(&cnt[q], 1);
                            if (pos < cap) cand[(size_t)q * cap + pos] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)p;
                        }
                    }
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
                                                         int64_t* idx_out, float* dist_out, const int* only_flagged = nullptr) {
    if (only_flagged && !only_flagged[blockIdx.x]) return;          // fallback launch: only the flagged queries
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

// Last step of the filtered path, one block per query: the sample's k best (local rows < S) + the candidates that passed the filter
// (rows >= S, unordered: their append order is a race) -> sorted by (distance bits, row) -> the k best.  A query whose candidates
// overflowed its list is flagged instead: the fallback launches behind this one recompute it exactly (distance row + radix select).
constexpr int FMAX = 4096, FCAP = 2048;
__global__ __launch_bounds__(256) void topk_finish_kernel(const int64_t* idx_s, const float* dist_s, const int* cnt, const unsigned long long* cand, int cap, int* flags,
                                                         int k, int64_t index_base, int64_t* idx_out, float* dist_out) {
    __shared__ unsigned long long keys[FMAX];
    const int q = blockIdx.x, n = cnt[q];
    if (n > cap) { if (threadIdx.x == 0) flags[q] = 1; return; }
    const int total = k + n;
    int n2 = 64;
    while (n2 < total) n2 <<= 1;
    for (int i = threadIdx.x; i < n2; i += 256) {
        unsigned long long key = ~0ull;
        if (i < k) key = ((unsigned long long)__float_as_uint(dist_s[(size_t)q * k + i]) << 32) | (unsigned)idx_s[(size_t)q * k + i];
        else if (i < total) key = cand[(size_t)q * cap + (i - k)];
        keys[i] = key;
    }
    __syncthreads();
    bitonic_sort(keys, n2);
    for (int i = threadIdx.x; i < k; i += 256) {
        const unsigned long long key = keys[i];
        idx_out[(size_t)q * k + i] = (int64_t)(key & 0xffffffffu) + index_base;
        dist_out[(size_t)q * k + i] = __uint_as_float((unsigned)(key >> 32));
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
int g_topk_filter = 1;          // ofx_tune(17, v): 1 (default) pools of >= 32,768 rows take the sample + filter path, 0 = always distance matrix + radix select
// rows of the sample pass: a sixteenth of the pool (the sample's k-th distance then admits ~16 k candidates per query), at least 4,096
static inline int topk_sample_rows(int np) { int s = std::max(4096, np / 16); return (s + DT - 1) / DT * DT; }
static inline bool topk_filtered(int np, int k) { return g_topk_filter && np >= 32768 && k <= 128 && topk_sample_rows(np) + DT <= np; }
size_t ofx_l2_topk_ws(int nq, int np) {
    const size_t ld = (size_t)(np + 3) / 4 * 4;
    // norms, the distance matrix (whole: the fallback of the filtered path may need every row), and the filtered path's lists:
    // per-query counters + flags, the sample's k best (k <= 128), FCAP candidate keys per query
    return al256((size_t)nq * 4) + al256((size_t)np * 4) + al256((size_t)nq * ld * 4) +
           al256((size_t)nq * 8) + al256((size_t)nq * 128 * 8) + al256((size_t)nq * 128 * 4) + al256((size_t)nq * FCAP * 8);
}

static int dist_grid(int tiles_q, int tiles_p) {
    static int cus = 0;
    if (cus == 0) { int dev = 0, v = 0; (void)hipGetDevice(&dev); cus = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256; }
    const int gx = std::max(1, std::min(tiles_p, (2 * cus + tiles_q - 1) / tiles_q));      // two blocks per CU over all panels
    return gx * tiles_q;
}

int ofx_launch_l2_topk(const float* Q, const float* P, int nq, int np, int D, int k, int64_t index_base, int64_t* idx,
                       float* dist, void* ws, size_t ws_bytes, hipStream_t s) {
    OFX_REQUIRE(Q && P && idx && dist && ws, OFX_EINVAL, "l2_topk: NULL argument");
    OFX_REQUIRE(nq > 0 && np > 0 && D > 0 && D % 32 == 0, OFX_ESHAPE, "l2_topk: nq=%d np=%d D=%d (D must be a multiple of 32)", nq, np, D);
    OFX_REQUIRE(k >= 1 && k <= KMAX && k <= np, OFX_ESHAPE, "l2_topk: k=%d must be in [1, min(%d, np)]", k, KMAX);
    OFX_REQUIRE(ws_bytes >= ofx_l2_topk_ws(nq, np), OFX_EWORKSPACE, "l2_topk: workspace %zu < %zu bytes", ws_bytes, ofx_l2_topk_ws(nq, np));
    OFX_REQUIRE(((uintptr_t)Q % 16 == 0) && ((uintptr_t)P % 16 == 0), OFX_EINVAL, "l2_topk: Q and P must be 16-byte aligned");
    static DeviceOnce attr;
    TRY(attr.run([]() -> int {
        OFX_HIP(hipFuncSetAttribute((const void*)dist_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DIST_LDS));
        OFX_HIP(hipFuncSetAttribute((const void*)dist_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DIST_LDS));
        return OFX_OK;
    }));
    const int ld = (np + 3) / 4 * 4;
    char* w = (char*)ws;
    float* qn = (float*)w; w += al256((size_t)nq * 4);
    float* pn = (float*)w; w += al256((size_t)np * 4);
    float* dm = (float*)w; w += al256((size_t)nq * ld * 4);
    int* cnt = (int*)w; int* flags = cnt + nq; w += al256((size_t)nq * 8);
    int64_t* idx_s = (int64_t*)w; w += al256((size_t)nq * 128 * 8);
    float* dist_s = (float*)w; w += al256((size_t)nq * 128 * 4);
    unsigned long long* cand = (unsigned long long*)w;
    const int tiles_q = (nq + DT - 1) / DT;
    hipLaunchKernelGGL(sqnorm_kernel, dim3(std::min((nq + 3) / 4, 4096)), dim3(256), 0, s, Q, qn, nq, D);
    hipLaunchKernelGGL(sqnorm_kernel, dim3(std::min((np + 3) / 4, 4096)), dim3(256), 0, s, P, pn, np, D);
    if (!topk_filtered(np, k)) {
        hipLaunchKernelGGL(dist_mfma_kernel<false>, dim3(dist_grid(tiles_q, (np + DT - 1) / DT)), dim3(256), DIST_LDS, s, Q, P, qn, pn, dm, nq, np, D, ld, 0,
                           (const float*)nullptr, 0, (int*)nullptr, (unsigned long long*)nullptr, 0, (const int*)nullptr, tiles_q);
        hipLaunchKernelGGL(topk_select_kernel, dim3(nq), dim3(256), 0, s, dm, ld, np, k, index_base, idx, dist, (const int*)nullptr);
        OFX_LAUNCH_CHECK();
        return OFX_OK;
    }
    // 1. the sample: rows [0, S) -> distance matrix [nq, S] -> its k best per query (local rows) and with them tau = the k-th distance
    const int S = topk_sample_rows(np), ldS = S;
    hipLaunchKernelGGL(dist_mfma_kernel<false>, dim3(dist_grid(tiles_q, S / DT)), dim3(256), DIST_LDS, s, Q, P, qn, pn, dm, nq, S, D, ldS, 0,
                       (const float*)nullptr, 0, (int*)nullptr, (unsigned long long*)nullptr, 0, (const int*)nullptr, tiles_q);
    hipLaunchKernelGGL(topk_select_kernel, dim3(nq), dim3(256), 0, s, dm, ldS, S, k, (int64_t)0, idx_s, dist_s, (const int*)nullptr);
    OFX_HIP(hipMemsetAsync(cnt, 0, (size_t)nq * 8, s));
    // 2. rows [S, np): distances stay in registers; what is <= tau goes to the query's candidate list
    hipLaunchKernelGGL(dist_mfma_kernel<true>, dim3(dist_grid(tiles_q, (np - S + DT - 1) / DT)), dim3(256), DIST_LDS, s, Q, P, qn, pn, (float*)nullptr, nq, np, D, 0, S,
                       (const float*)(dist_s + (k - 1)), k, cnt, cand, FCAP, (const int*)nullptr, tiles_q);
    // 3. sample's k best + candidates -> the k best; overflowed lists are flagged ...
    hipLaunchKernelGGL(topk_finish_kernel, dim3(nq), dim3(256), 0, s, idx_s, dist_s, cnt, cand, FCAP, flags, k, index_base, idx, dist);
    // 4. ... and recomputed exactly (whole distance rows + radix select); both launches return at once when no query is flagged
    hipLaunchKernelGGL(dist_mfma_kernel<false>, dim3(dist_grid(tiles_q, (np + DT - 1) / DT)), dim3(256), DIST_LDS, s, Q, P, qn, pn, dm, nq, np, D, ld, 0,
                       (const float*)nullptr, 0, (int*)nullptr, (unsigned long long*)nullptr, 0, (const int*)flags, tiles_q);
    hipLaunchKernelGGL(topk_select_kernel, dim3(nq), dim3(256), 0, s, dm, ld, np, k, index_base, idx, dist, (const int*)flags);
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
