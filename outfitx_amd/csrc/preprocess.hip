// CLIP image preprocessing on the GPU from packed uint8 images ("next" row N2): what the reference does on
// the host with CLIPImageProcessor(do_convert_rgb=False) (src/models/encoders/image_encoders/clip_image_encoder.py:29-31,
// 69-71): resize the shortest edge to `size` (PIL BICUBIC with antialiasing), centre crop, x 1/255, (x - mean) / std.
//
// The resize is PIL's ImagingResample (Pillow src/libImaging/Resample.c; third-party, not vendored in the reference):
// per axis, every output coordinate owns a window [xmin, xmin+n) of input pixels and n coefficients
//   k = bicubic((x + xmin - center + 0.5) / filterscale) / sum,  center = (xx + 0.5) * scale,  support = 2 * max(scale, 1)
// rounded to 22-bit fixed point; out = clip8((2^21 + sum_x pixel[x] * k[x]) >> 22); horizontal pass first, 8-bit
// intermediate, then vertical.  The coefficient tables are computed on the host in double precision in PIL's operation
// order (so the integers are PIL's integers), cached per (in, out, crop) and shipped with the batch; the two passes below are
// integer and therefore bit-exact.  Only the 224 x 224 crop window is ever computed.
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include <cmath>
#include <cstring>
#include <type_traits>

#include "ofx_common.h"

// No fused multiply-adds in this file: the float epilogue must round like numpy's separate *, -, / and the host-side
// coefficient maths like PIL's C (x86-64 baseline, no FMA).
#pragma clang fp contract(off)

namespace {

constexpr int PBITS = 22;

struct ImgDesc {
    long long src_off;     // byte offset of the image in the packed source (row-major, `c` interleaved channels)
    long long inter_off;   // byte offset of its intermediate (horizontally resampled) rows
    int h, w, c;
    int hplan, vplan;      // int offsets of the axis plans in the blob: [ksize, bounds(2 * size), coeffs(size * ksize)]
    int y0, nrows;         // input rows [y0, y0 + nrows) feed the vertical pass
};

// ---- horizontal pass: inter[row][col] = one RGBX dword per pixel, for the needed input rows and the crop's columns.
// RGB sources are read with ONE (unaligned) dword load per tap - the three channels plus one byte of the next pixel - so
// the packed source must stay readable 1 byte past every image (offsets are 16-byte aligned and the buffer ends in slack).
__device__ __forceinline__ unsigned load_u32_unaligned(const uint8_t* p) {
    unsigned v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
// One thread per output column: its taps (window start + up to KT fixed-point coefficients) stay in registers while the
// block walks ROWS_H input rows, so the inner loop is one dword load + three integer MACs per tap.  KT is the smallest
// bucket covering the batch's widest window (taps past a column's own count carry coefficient 0 and a clamped address).
constexpr int ROWS_H = 16;
template <int KT>
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* src, const ImgDesc* descs, const int* blob, uint8_t* inter, int size) {
    const ImgDesc d = descs[blockIdx.y];
    const int row0 = blockIdx.x * ROWS_H;
    if (row0 >= d.nrows) return;
    const int* plan = blob + d.hplan;
    const int ksize = plan[0];
    const int* bounds = plan + 1;
    const int* coef = plan + 1 + 2 * size;
    unsigned* out = (unsigned*)(inter + d.inter_off);
    const int rows = min(ROWS_H, d.nrows - row0);
    for (int col = threadIdx.x; col < size; col += 256) {
        const int xmin = bounds[2 * col], n = bounds[2 * col + 1];
        const uint8_t* px = src + d.src_off + ((size_t)(d.y0 + row0) * d.w + xmin) * d.c;
        const size_t pitch = (size_t)d.w * d.c;
        if (KT > 0 && d.c == 3) {
            int kv[KT > 0 ? KT : 1], xo[KT > 0 ? KT : 1];
#pragma unroll
            for (int x = 0; x < KT; ++x) { kv[x] = x < n ? coef[col * ksize + x] : 0; xo[x] = 3 * max(min(x, n - 1), 0); }
            for (int r = 0; r < rows; ++r) {
                int s0 = 1 << (PBITS - 1), s1 = s0, s2 = s0;
#pragma unroll
                for (int x = 0; x < KT; ++x) {
                    const unsigned v = load_u32_unaligned(px + xo[x]);
                    s0 += (int)(v & 255u) * kv[x]; s1 += (int)((v >> 8) & 255u) * kv[x]; s2 += (int)((v >> 16) & 255u) * kv[x];
                }
                out[(size_t)(row0 + r) * size + col] = (unsigned)min(max(s0 >> PBITS, 0), 255) | ((unsigned)min(max(s1 >> PBITS, 0), 255) << 8) |
                                                       ((unsigned)min(max(s2 >> PBITS, 0), 255) << 16);
                px += pitch;
            }
        } else {                                            // wide windows (heavy down-scaling) and grey sources: taps from memory
            const int* k = coef + col * ksize;
            for (int r = 0; r < rows; ++r) {
                int s0 = 1 << (PBITS - 1), s1 = s0, s2 = s0;
                if (d.c == 3) {
                    for (int x = 0; x < n; ++x) {
                        const unsigned v = load_u32_unaligned(px + 3 * x);
                        s0 += (int)(v & 255u) * k[x]; s1 += (int)((v >> 8) & 255u) * k[x]; s2 += (int)((v >> 16) & 255u) * k[x];
                    }
                } else {
                    for (int x = 0; x < n; ++x) s0 += px[x] * k[x];
                    s1 = s2 = s0;
                }
                out[(size_t)(row0 + r) * size + col] = (unsigned)min(max(s0 >> PBITS, 0), 255) | ((unsigned)min(max(s1 >> PBITS, 0), 255) << 8) |
                                                       ((unsigned)min(max(s2 >> PBITS, 0), 255) << 16);
                px += pitch;
            }
        }
    }
}

// a * b and a - b as two separately rounded instructions: the compiler may not fuse them into v_fma (numpy does not)
__device__ __forceinline__ float mul_then_sub(float a, float b, float c) {
    float t, r;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b));
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(c));
    return r;
}

// ---- vertical pass + rescale + normalise -> planar fp32 [N, 3, size, size].  float ops are individually rounded
// (no fma contraction) so the result equals numpy's (a * f32(1/255) - mean) / std bit for bit.
constexpr int ROWS_V = 8;
// TO = float: planar fp32 pixel_values [N, 3, size, size].  TO = bf16 / f16: the im2col operand of the ViT patch-embedding GEMM,
// [N * (size/patch)^2, 3 * patch^2] with column c * patch^2 + ky * patch + kx (what patchify_kernel makes of the fp32 pixels) -
// the fp32 pixel tensor is then never written.
template <typename TO>
__global__ __launch_bounds__(256) void resample_v_kernel(const ImgDesc* descs, const int* blob, const uint8_t* inter, TO* out, int size, int patch,
                                                         float m0, float m1, float m2, float s0d, float s1d, float s2d) {
    const ImgDesc d = descs[blockIdx.y];
    const int* plan = blob + d.vplan;
    const int ksize = plan[0];
    const int* bounds = plan + 1;
    const int* coef = plan + 1 + 2 * size;
    const unsigned* in = (const unsigned*)(inter + d.inter_off);
    const float r255 = (float)(1.0 / 255.0);
    constexpr bool PATCHES = !std::is_same<TO, float>::value;
    const int total = size * size, g = PATCHES ? size / patch : 1, pp = patch * patch;
    TO* o = out + (size_t)blockIdx.y * 3 * size * size;          // both layouts hold 3 * size^2 elements per image
    for (int rr = 0; rr < ROWS_V; ++rr) {
        const int row = blockIdx.x * ROWS_V + rr;               // block-uniform: bounds and coefficients come through scalar loads
        if (row >= size) break;
        const int ymin = bounds[2 * row] - d.y0, n = bounds[2 * row + 1];
        const int* k = coef + row * ksize;
        for (int col = threadIdx.x; col < size; col += 256) {
            int a0 = 1 << (PBITS - 1), a1 = a0, a2 = a0;
            const unsigned* px = in + (size_t)ymin * size + col;
            for (int y = 0; y < n; ++y) {
                const int kv = k[y];
                const unsigned v = px[(size_t)y * size];
                a0 += (int)(v & 255u) * kv; a1 += (int)((v >> 8) & 255u) * kv; a2 += (int)((v >> 16) & 255u) * kv;
            }
            const float v0 = (float)min(max(a0 >> PBITS, 0), 255), v1 = (float)min(max(a1 >> PBITS, 0), 255), v2 = (float)min(max(a2 >> PBITS, 0), 255);
            const float f0 = __fdiv_rn(mul_then_sub(v0, r255, m0), s0d), f1 = __fdiv_rn(mul_then_sub(v1, r255, m1), s1d),
                        f2 = __fdiv_rn(mul_then_sub(v2, r255, m2), s2d);
            if (PATCHES) {
                const int py = row / patch, ky = row - py * patch, pxi = col / patch, kx = col - pxi * patch;
                TO* q = o + (size_t)(py * g + pxi) * 3 * pp + ky * patch + kx;
                q[0] = (TO)f0; q[pp] = (TO)f1; q[2 * pp] = (TO)f2;
            } else {
                const int p = row * size + col;
                o[p] = (TO)f0; o[total + p] = (TO)f1; o[2 * total + p] = (TO)f2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------- host planning
double bicubic_filter(double x) {                     // Resample.c: a = -0.5
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// plan of one axis, restricted to outputs [crop0, crop0 + n_out): [ksize | bounds | coeffs]
std::vector<int> plan_axis(int in_size, int out_size, int crop0, int n_out) {
    std::vector<int> v;
    if (in_size == out_size) {                        // PIL skips the pass: identity, exact with k = 2^22
        v.assign(1 + 2 * n_out + n_out, 0);
        v[0] = 1;
        for (int i = 0; i < n_out; ++i) { v[1 + 2 * i] = crop0 + i; v[2 + 2 * i] = 1; v[1 + 2 * n_out + i] = 1 << PBITS; }
        return v;
    }
    const double scale = (double)in_size / out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    v.assign(1 + 2 * n_out + (size_t)n_out * ksize, 0);
    v[0] = ksize;
    std::vector<double> k(ksize);
    const double ss = 1.0 / filterscale;
    for (int i = 0; i < n_out; ++i) {
        const int xx = crop0 + i;
        const double center = 0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) if (ww != 0.0) k[x] /= ww;
        v[1 + 2 * i] = xmin; v[2 + 2 * i] = xmax;
        int* kk = v.data() + 1 + 2 * n_out + (size_t)i * ksize;
        for (int x = 0; x < xmax; ++x)
            kk[x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PBITS)) : (int)(0.5 + k[x] * (1 << PBITS));
    }
    return v;
}

struct Geometry { int nw, nh, left, top; };
Geometry geometry(int h, int w, int size) {            // transformers get_resize_output_image_size(default_to_square=False) + center_crop
    const int s = w <= h ? w : h, l = w <= h ? h : w;
    const int new_long = (int)((double)size * l / s);
    Geometry g;
    g.nw = w <= h ? size : new_long; g.nh = w <= h ? new_long : size;
    g.left = (g.nw - size) / 2; g.top = (g.nh - size) / 2;
    return g;
}

std::mutex g_plan_mu;
std::map<std::tuple<int, int, int, int>, std::vector<int>> g_plan_cache;
const std::vector<int>& cached_plan(int in_size, int out_size, int crop0, int n_out) {
    auto key = std::make_tuple(in_size, out_size, crop0, n_out);
    auto it = g_plan_cache.find(key);
    if (it == g_plan_cache.end()) {      // std::map: inserting never moves the plans already handed out
        it = g_plan_cache.emplace(key, plan_axis(in_size, out_size, crop0, n_out)).first;
    }
    return it->second;
}

// Pinned host staging for the per-batch plan (descriptors + coefficient tables, a few KB): a ring of 4 slots PER DEVICE, each with
// an event that marks when its copy has left it.  A call waits on the host only when the copy issued four calls earlier on the same
// device is still in flight (documented in include/ofx.h; every other entry point is wait-free).
struct PinnedRing {
    struct Slot { char* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; };
    Slot slot[4]; int next = 0; std::mutex mu;
};
std::mutex g_rings_mu;
std::map<int, PinnedRing> g_rings;
PinnedRing& ring_for_current_device() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_rings_mu);
    return g_rings[dev];
}

struct Batch {
    std::vector<ImgDesc> descs;
    std::vector<int> blob;
    size_t inter_bytes = 0;
    int max_rows = 0, max_ksize_h = 0;
};
int plan_batch(const long long* offsets, const int* hs, const int* ws, int N, int channels, int size, Batch* b) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (g_plan_cache.size() > 4096) g_plan_cache.clear();      // evict between batches only: the loop below holds pointers into the cache
    std::map<const std::vector<int>*, int> placed;
    b->descs.resize(N);
    for (int i = 0; i < N; ++i) {
        const int h = hs[i], w = ws[i];
        OFX_REQUIRE(h > 0 && w > 0 && h <= 16384 && w <= 16384, OFX_ESHAPE, "clip_preprocess: image %d is %d x %d", i, h, w);
        const Geometry g = geometry(h, w, size);
        const std::vector<int>* hp = &cached_plan(w, g.nw, g.left, size);
        const std::vector<int>* vp = &cached_plan(h, g.nh, g.top, size);
        for (const std::vector<int>* p : {hp, vp})
            if (!placed.count(p)) { placed[p] = (int)b->blob.size(); b->blob.insert(b->blob.end(), p->begin(), p->end()); }
        ImgDesc& d = b->descs[i];
        d.src_off = offsets[i]; d.h = h; d.w = w; d.c = channels; d.hplan = placed[hp]; d.vplan = placed[vp];
        const int* vb = vp->data() + 1;
        d.y0 = vb[0];
        d.nrows = vb[2 * (size - 1)] + vb[2 * (size - 1) + 1] - d.y0;
        d.inter_off = (long long)b->inter_bytes;
        b->inter_bytes += align_up((size_t)d.nrows * size * 4, 16);        // one RGBX dword per intermediate pixel
        b->max_rows = std::max(b->max_rows, d.nrows);
        b->max_ksize_h = std::max(b->max_ksize_h, (*hp)[0]);
    }
    return OFX_OK;
}

}  // namespace

extern "C" size_t ofx_clip_preprocess_ws(const int* heights, const int* widths, int N, int channels, int size) {
    if (!heights || !widths || N <= 0 || size <= 0) return 0;
    Batch b;
    std::vector<long long> off(N, 0);
    if (plan_batch(off.data(), heights, widths, N, channels, size, &b) != OFX_OK) return 0;
    return align_up(b.descs.size() * sizeof(ImgDesc), 256) + align_up(b.blob.size() * 4, 256) + align_up(b.inter_bytes, 256) + 1024;
}

extern "C" int ofx_clip_preprocess(const uint8_t* src, const long long* offsets, const int* heights, const int* widths, int N, int channels, int size,
                                   const float* mean, const float* stdv, float* out, void* ws, size_t ws_bytes, ofx_stream stream) {
    return ofx_preprocess_to(src, offsets, heights, widths, N, channels, size, mean, stdv, out, nullptr, 0, 0, ws, ws_bytes, (hipStream_t)stream);
}

// out != NULL: fp32 pixel_values; patches != NULL: operand-type im2col rows of the patch-embedding GEMM (exactly one of the two)
int ofx_preprocess_to(const uint8_t* src, const long long* offsets, const int* heights, const int* widths, int N, int channels, int size,
                      const float* mean, const float* stdv, float* out, void* patches, int patch, int op_dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
    OFX_REQUIRE((out != nullptr) != (patches != nullptr), OFX_EINVAL, "clip_preprocess: exactly one output");
    OFX_REQUIRE(!patches || (patch > 0 && size % patch == 0), OFX_ESHAPE, "clip_preprocess: size=%d patch=%d", size, patch);
    OFX_REQUIRE(src && offsets && heights && widths && mean && stdv && ws && N > 0, OFX_EINVAL, "clip_preprocess: NULL argument");
    OFX_REQUIRE((channels == 3 || channels == 1) && size > 0 && size <= 1024, OFX_ESHAPE, "clip_preprocess: channels=%d size=%d", channels, size);
    hipStream_t s = (hipStream_t)stream;
    Batch b;
    TRY(plan_batch(offsets, heights, widths, N, channels, size, &b));
    Bump bump(ws, ws_bytes);
    ImgDesc* d_desc = (ImgDesc*)bump.take<char>(b.descs.size() * sizeof(ImgDesc));
    int* d_blob = bump.take<int>(b.blob.size());
    uint8_t* d_inter = (uint8_t*)bump.take<char>(b.inter_bytes);
    OFX_REQUIRE(bump.ok, OFX_EWORKSPACE, "clip_preprocess: workspace %zu < %zu bytes", ws_bytes, bump.off);
    {
        PinnedRing& ring = ring_for_current_device();
        std::lock_guard<std::mutex> lk(ring.mu);
        PinnedRing::Slot& st = ring.slot[ring.next];
        ring.next = (ring.next + 1) & 3;
        const size_t nd = align_up(b.descs.size() * sizeof(ImgDesc), 256), nb = b.blob.size() * 4;
        if (!st.ev) OFX_HIP(hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
        else if (hipEventQuery(st.ev) != hipSuccess) OFX_HIP(hipEventSynchronize(st.ev));      // only when 4 earlier plans are still in flight
        if (st.cap < nd + nb) {
            if (st.p) (void)hipHostFree(st.p);
            st.cap = (nd + nb) * 2;
            OFX_HIP(hipHostMalloc((void**)&st.p, st.cap, hipHostMallocDefault));
        }
        memcpy(st.p, b.descs.data(), b.descs.size() * sizeof(ImgDesc));
        memcpy(st.p + nd, b.blob.data(), nb);
        OFX_HIP(hipMemcpyAsync(d_desc, st.p, b.descs.size() * sizeof(ImgDesc), hipMemcpyHostToDevice, s));
        OFX_HIP(hipMemcpyAsync(d_blob, st.p + nd, nb, hipMemcpyHostToDevice, s));
        OFX_HIP(hipEventRecord(st.ev, s));
    }
    ProfScope prof(PROF_OTHER, s);
    const dim3 gh((b.max_rows + ROWS_H - 1) / ROWS_H, N);
    if (channels == 3 && b.max_ksize_h <= 8) hipLaunchKernelGGL(resample_h_kernel<8>, gh, dim3(256), 0, s, src, d_desc, d_blob, d_inter, size);
    else if (channels == 3 && b.max_ksize_h <= 16) hipLaunchKernelGGL(resample_h_kernel<16>, gh, dim3(256), 0, s, src, d_desc, d_blob, d_inter, size);
    else hipLaunchKernelGGL(resample_h_kernel<0>, gh, dim3(256), 0, s, src, d_desc, d_blob, d_inter, size);
    const dim3 gv((size + ROWS_V - 1) / ROWS_V, N);
#define RV(TO, dst) hipLaunchKernelGGL(resample_v_kernel<TO>, gv, dim3(256), 0, s, d_desc, d_blob, d_inter, (TO*)(dst), size, patch, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2])
    if (out) RV(float, out);
    else if (op_dtype == OFX_F16) RV(f16_t, patches);
    else RV(bf16_t, patches);
#undef RV
    OFX_LAUNCH_CHECK();
    return OFX_OK;
}
