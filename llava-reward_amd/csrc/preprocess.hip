// Image hand-over on the GPU.  Phi-3.5-V first; the Qwen2-VL and LLaVA-NeXT processors (third party,
// transformers) reuse the same resampler further down.
// Phi-3.5-V: uint8 RGB image in HBM -> pixel_values [num_crops+1, 3, 336, 336]
// fp32, the tensor custom_forward consumes.  Replaces the reference's single-process CPU processor
// (llava_reward/models/base_mllm/phi3_v/processing_phi3_v.py):
//   :85-107   HD_transform: portrait images are transposed, scale search over hd_num, torchvision resize of the PIL image
//             (= Pillow's two-pass 8-bit bilinear resampler, src/libImaging/Resample.c), padding_336 (:62-72, white rows),
//             transposed back;
//   :262-288  ToTensor + Normalize (fp32), bicubic 336x336 global view (torch F.interpolate, A = -0.75, clamped taps),
//             crops tiled row-major behind the global view, zero crops up to num_crops + 1.
// All of it is byte / integer work bounded by HBM: 3 bytes read per source pixel, 23 MB of fp32 written per image.  The
// integer part (the resampled uint8 image, hence every local crop) is bit-exact with Pillow; the bicubic view is fp32
// arithmetic in the oracle's tap order.
//   resample_table_kernel   Pillow's coefficient tables (double arithmetic, one thread per output index; contraction off
//                           so that every operation rounds as the C code on the host does)
//   resample_kernel         one pass along x or y, 22-bit fixed point, uint8 out
//   hd_tile_kernel          pad / transpose back / normalise / tile: float4 stores, x fastest
//   hd_global_kernel        16-tap bicubic of the normalised padded image
#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <string>

#include "../../include/llava_reward_hip.h"
#include "common.h"
#include "engine.h"

namespace lr {

constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr int CROP = 336;

// Source view in the orientation the resize runs in: pixel (Y, X) lives at base + (Y * sy + X * sx) * 3.
struct U8View { const unsigned char* p; int sy, sx; };

enum { FILTER_BILINEAR = 0, FILTER_BICUBIC = 1 };              // Pillow's BILINEAR (support 1) and BICUBIC (support 2, a = -0.5)

#pragma clang fp contract(off)
__device__ __forceinline__ double pil_filter(int filter, double x) {
    if (x < 0.0) x = -x;
    if (filter == FILTER_BILINEAR) return x < 1.0 ? 1.0 - x : 0.0;
    const double a = -0.5;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

__global__ void resample_table_kernel(int in_size, int out_size, int ksize, int filter, int* __restrict__ bounds, int* __restrict__ kk) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    if (xx >= out_size) return;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = (filter == FILTER_BICUBIC ? 2.0 : 1.0) * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = 0.0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    int* k = kk + (size_t)xx * ksize;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += pil_filter(filter, (x + xmin - center + 0.5) * ss);
    for (int x = 0; x < ksize; ++x) {
        double w = 0.0;
        if (x < xmax) {
            w = pil_filter(filter, (x + xmin - center + 0.5) * ss);
            if (ww != 0.0) w /= ww;
        }
        k[x] = w < 0 ? (int)(-0.5 + w * (1 << PRECISION_BITS)) : (int)(0.5 + w * (1 << PRECISION_BITS));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
}

// dst[oy][ox][c] (contiguous) = clip8((2^21 + sum_t src[..tap t..][c] * k[t]) >> 22); AXIS 1: taps along X, AXIS 0: along Y.
template <int AXIS>
__global__ void resample_kernel(U8View src, unsigned char* __restrict__ dst, int OH, int OW, const int* __restrict__ bounds,
                                const int* __restrict__ kk, int ksize) {
    const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
    if (ox >= OW) return;
    const int o = AXIS ? ox : oy;
    const int x0 = bounds[2 * o], n = bounds[2 * o + 1];
    const int* k = kk + (size_t)o * ksize;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    const size_t step = (size_t)(AXIS ? src.sx : src.sy) * 3;
    const unsigned char* s = src.p + (AXIS ? ((size_t)oy * src.sy + (size_t)x0 * src.sx) : ((size_t)x0 * src.sy + (size_t)ox * src.sx)) * 3;
    for (int t = 0; t < n; ++t, s += step) {
        const int w = k[t];
        a0 += s[0] * w; a1 += s[1] * w; a2 += s[2] * w;
    }
    auto clip8 = [](int v) { v >>= PRECISION_BITS; return (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v); };
    unsigned char* d = dst + ((size_t)oy * OW + ox) * 3;
    d[0] = clip8(a0); d[1] = clip8(a1); d[2] = clip8(a2);
}

struct HdImage {            // the padded image, addressed in FINAL orientation (Yf, Xf), channel c
    U8View r;               // resized image in resize orientation, new_h x new_w
    int new_h, new_w, top, trans;
    int H, W;               // final padded size
};

__device__ __forceinline__ float hd_norm(unsigned char u, int c) {
    // ToTensor: u / 255; Normalize: (x - mean) / std, each step rounded to fp32 (correctly rounded division)
    const float mean = c == 0 ? 0.48145466f : c == 1 ? 0.4578275f : 0.40821073f;
    const float sd = c == 0 ? 0.26862954f : c == 1 ? 0.26130258f : 0.27577711f;
    const float x = (float)u / 255.0f;
    return (x - mean) / sd;
}

__device__ __forceinline__ float hd_pixel(const HdImage& im, int yf, int xf, int c) {
    const int Y = (im.trans ? xf : yf) - im.top, X = im.trans ? yf : xf;
    unsigned char u = 255;                                      // padding_336: white
    if (Y >= 0 && Y < im.new_h) u = im.r.p[((size_t)Y * im.r.sy + (size_t)X * im.r.sx) * 3 + c];
    return hd_norm(u, c);
}

// out[1 + cy * (W/336) + cx][c][y][x] = normalised padded pixel; crops past the last local one are zero-filled.
__global__ void hd_tile_kernel(HdImage im, float* __restrict__ out, int n_slots) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;        // float4 index inside one crop-channel plane row set
    const int per_crop = 3 * CROP * CROP / 4;
    const int crop = blockIdx.y + 1;
    if (q >= per_crop || crop >= n_slots) return;
    const int c = q / (CROP * CROP / 4), rem = q - c * (CROP * CROP / 4);
    const int y = rem / (CROP / 4), x = (rem - y * (CROP / 4)) * 4;
    const int wc = im.W / CROP, n_local = (im.H / CROP) * wc;
    float4 v = {0.f, 0.f, 0.f, 0.f};
    if (crop - 1 < n_local) {
        const int cy = (crop - 1) / wc, cx = (crop - 1) - cy * wc;
        const int yf = cy * CROP + y, xf = cx * CROP + x;
        v.x = hd_pixel(im, yf, xf, c); v.y = hd_pixel(im, yf, xf + 1, c);
        v.z = hd_pixel(im, yf, xf + 2, c); v.w = hd_pixel(im, yf, xf + 3, c);
    }
    *(float4*)(out + (size_t)crop * 3 * CROP * CROP + (size_t)q * 4) = v;
}

__device__ __forceinline__ void cubic_taps(int in_size, int o, int* idx, float* w) {
    const float A = -0.75f;
    const float scale = (float)in_size / (float)CROP;
    const float real = scale * ((float)o + 0.5f) - 0.5f;
    const float fl = floorf(real);
    const float t = real - fl;
    const int i0 = (int)fl;
    auto c1 = [A](float x) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; };
    auto c2 = [A](float x) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; };
    w[0] = c2(t + 1.f); w[1] = c1(t); w[2] = c1(1.f - t); w[3] = c2(2.f - t);
#pragma unroll
    for (int j = 0; j < 4; ++j) idx[j] = min(max(i0 - 1 + j, 0), in_size - 1);
}

// out[0][c][oy][ox]: F.interpolate(bicubic, align_corners=False) of the normalised padded image, taps summed x first.
__global__ void hd_global_kernel(HdImage im, float* __restrict__ out) {
    const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y, c = blockIdx.z;
    if (ox >= CROP) return;
    int iy[4], ix[4];
    float wy[4], wx[4];
    cubic_taps(im.H, oy, iy, wy);
    cubic_taps(im.W, ox, ix, wx);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float inner = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) inner += hd_pixel(im, iy[i], ix[j], c) * wx[j];
        acc += inner * wy[i];
    }
    out[((size_t)c * CROP + oy) * CROP + ox] = acc;
}

// Qwen2-VL image processor tail (transformers image_processing_qwen2_vl, PIL backend = the slow processor of the pinned 4.50):
// rescale = float64(u) * (1/255) cast to fp32, normalize = (x - mean) / std in fp32, then patchify: row = ((by * GW/2 + bx) * 2
// + my) * 2 + mx over 2x2 merge blocks, column = c * 392 + t * 196 + py * 14 + px with the single frame repeated for t = 0, 1.
__global__ void qwen_patchify_kernel(U8View r, float* __restrict__ out, int gh, int gw) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;        // (c, py, px) of one patch
    const int row = blockIdx.y;
    if (e >= 3 * 196) return;
    const int c = e / 196, rem = e - c * 196, py = rem / 14, px = rem - py * 14;
    const int blk = row >> 2, my = (row >> 1) & 1, mx = row & 1;
    const int by = blk / (gw / 2), bx = blk - by * (gw / 2);
    const int y = ((by * 2 + my) * 14) + py, x = ((bx * 2 + mx) * 14) + px;
    const unsigned char u = r.p[((size_t)y * r.sy + (size_t)x * r.sx) * 3 + c];
    const float mean = c == 0 ? 0.48145466f : c == 1 ? 0.4578275f : 0.40821073f;
    const float sd = c == 0 ? 0.26862954f : c == 1 ? 0.26130258f : 0.27577711f;
    const float v = ((float)((double)u * 0.00392156862745098) - mean) / sd;
    float* o = out + (size_t)row * 1176 + c * 392 + rem;
    o[0] = v;
    o[196] = v;
}

// LLaVA-NeXT image processor tail (transformers image_processing_llava_next, PIL arithmetic): crop 0 = the whole image resized
// to 336x336, crops 1.. = the aspect-preserving resize centred on a zero (uint8 0) canvas of the best grid resolution, cut
// into 336x336 tiles row-major; rescale / normalize as above; crops past the last one are zero-filled.
struct LlavaImage { U8View base, hi; int new_h, new_w, top, left, gh, gw; };

__global__ void llava_tile_kernel(LlavaImage im, float* __restrict__ out, int n_slots) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int per_crop = 3 * CROP * CROP / 4;
    const int crop = blockIdx.y;
    if (q >= per_crop || crop >= n_slots) return;
    const int c = q / (CROP * CROP / 4), rem = q - c * (CROP * CROP / 4);
    const int y = rem / (CROP / 4), x = (rem - y * (CROP / 4)) * 4;
    const float mean = c == 0 ? 0.48145466f : c == 1 ? 0.4578275f : 0.40821073f;
    const float sd = c == 0 ? 0.26862954f : c == 1 ? 0.26130258f : 0.27577711f;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (crop < 1 + im.gh * im.gw) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned char u = 0;
            if (crop == 0) {
                u = im.base.p[((size_t)y * im.base.sy + (size_t)(x + i) * im.base.sx) * 3 + c];
            } else {
                const int cy = (crop - 1) / im.gw, cx = (crop - 1) - cy * im.gw;
                const int Y = cy * CROP + y - im.top, X = cx * CROP + x + i - im.left;
                if (Y >= 0 && Y < im.new_h && X >= 0 && X < im.new_w) u = im.hi.p[((size_t)Y * im.hi.sy + (size_t)X * im.hi.sx) * 3 + c];
            }
            v[i] = ((float)((double)u * 0.00392156862745098) - mean) / sd;
        }
    }
    *(float4*)(out + (size_t)crop * 3 * CROP * CROP + (size_t)q * 4) = float4{v[0], v[1], v[2], v[3]};
}
#pragma clang fp contract(fast)

// ---- host side: geometry of HD_transform in the same double arithmetic as the reference's Python ----
struct HdGeom { int trans, IH, IW, new_h, new_w, top, tar; };

static HdGeom hd_geometry(int height, int width, int hd_num) {
    if (height < 1 || width < 1 || hd_num < 1 || hd_num > 64) throw std::runtime_error("hd_transform: bad image size or num_crops");
    HdGeom g{};
    g.trans = width < height;
    int w = width, h = height;
    if (g.trans) { w = height; h = width; }
    g.IH = h; g.IW = w;
    const double ratio = (double)w / (double)h;
    int scale = 1;
    while (scale * std::ceil(scale / ratio) <= hd_num) ++scale;
    --scale;
    g.new_w = scale * CROP;
    g.new_h = (int)(g.new_w / ratio);
    if (g.new_h < 1) throw std::runtime_error("hd_transform: aspect ratio too extreme (resized height is 0)");
    g.tar = (int)(std::ceil(g.new_h / 336.0) * 336);
    g.top = (int)((g.tar - g.new_h) / 2.0);
    return g;
}

static size_t al256(size_t n) { return (n + 255) & ~(size_t)255; }

static int pil_ksize(int in, int out, int filter) {
    const double s = (double)in / out;
    return (int)std::ceil((filter == FILTER_BICUBIC ? 2.0 : 1.0) * (s < 1.0 ? 1.0 : s)) * 2 + 1;
}

// Scratch of one two-pass resize IH x IW -> OH x OW: both coefficient tables, the horizontal-pass image, the result.
struct ResizeLayout { size_t bx, kx, by, ky, pass1, pass2, total; int ksx, ksy; };
static ResizeLayout resize_layout(int IH, int IW, int OH, int OW, int filter) {
    ResizeLayout l{};
    l.ksx = pil_ksize(IW, OW, filter);
    l.ksy = pil_ksize(IH, OH, filter);
    size_t o = 0;
    l.bx = o; o += al256((size_t)OW * 2 * 4);
    l.kx = o; o += al256((size_t)OW * l.ksx * 4);
    l.by = o; o += al256((size_t)OH * 2 * 4);
    l.ky = o; o += al256((size_t)OH * l.ksy * 4);
    l.pass1 = o; o += al256((size_t)IH * OW * 3);
    l.pass2 = o; o += al256((size_t)OH * OW * 3);
    l.total = o;
    return l;
}

// Pillow's Image.resize((OW, OH), filter) of the view `v` (IH x IW); returns the view of the result (the source itself when
// neither size changes).  Horizontal pass first, then vertical, uint8 in between (Resample.c ImagingResample).
static U8View enqueue_resize(U8View v, int IH, int IW, int OH, int OW, int filter, char* ws, const ResizeLayout& l, hipStream_t st) {
    if (OW != IW) {
        int* bx = (int*)(ws + l.bx); int* kx = (int*)(ws + l.kx);
        hipLaunchKernelGGL(resample_table_kernel, dim3((OW + 127) / 128), dim3(128), 0, st, IW, OW, l.ksx, filter, bx, kx);
        unsigned char* d = (unsigned char*)(ws + l.pass1);
        hipLaunchKernelGGL(resample_kernel<1>, dim3((OW + 127) / 128, IH), dim3(128), 0, st, v, d, IH, OW, bx, kx, l.ksx);
        v = U8View{d, OW, 1};
    }
    if (OH != IH) {
        int* by = (int*)(ws + l.by); int* ky = (int*)(ws + l.ky);
        hipLaunchKernelGGL(resample_table_kernel, dim3((OH + 127) / 128), dim3(128), 0, st, IH, OH, l.ksy, filter, by, ky);
        unsigned char* d = (unsigned char*)(ws + l.pass2);
        hipLaunchKernelGGL(resample_kernel<0>, dim3((OW + 127) / 128, OH), dim3(128), 0, st, v, d, OH, OW, by, ky, l.ksy);
        v = U8View{d, OW, 1};
    }
    return v;
}

// smart_resize (transformers image_processing_qwen2_vl.py:62-88): sizes divisible by 28 inside [min_pixels, max_pixels];
// Python's round() is round-half-to-even = nearbyint under the default rounding mode.
static void qwen_smart_resize(int height, int width, int64_t min_pixels, int64_t max_pixels, int* oh, int* ow) {
    if (height < 1 || width < 1 || min_pixels < 1 || max_pixels < min_pixels) throw std::runtime_error("qwen_image: bad size or pixel bounds");
    const int factor = 28;
    if ((double)std::max(height, width) / std::min(height, width) > 200)
        throw std::runtime_error("qwen_image: absolute aspect ratio must be smaller than 200");
    int64_t h_bar = (int64_t)std::nearbyint((double)height / factor) * factor;
    int64_t w_bar = (int64_t)std::nearbyint((double)width / factor) * factor;
    if (h_bar * w_bar > max_pixels) {
        const double beta = std::sqrt(((double)height * width) / (double)max_pixels);
        h_bar = std::max<int64_t>(factor, (int64_t)std::floor(height / beta / factor) * factor);
        w_bar = std::max<int64_t>(factor, (int64_t)std::floor(width / beta / factor) * factor);
    } else if (h_bar * w_bar < min_pixels) {
        const double beta = std::sqrt((double)min_pixels / ((double)height * width));
        h_bar = (int64_t)std::ceil(height * beta / factor) * factor;
        w_bar = (int64_t)std::ceil(width * beta / factor) * factor;
    }
    if (h_bar < factor || w_bar < factor || h_bar > 16384 || w_bar > 16384) throw std::runtime_error("qwen_image: resized size out of range");
    *oh = (int)h_bar; *ow = (int)w_bar;
}

// select_best_resolution + get_patch_output_size (transformers image_processing_utils.py), same double arithmetic.
struct LlavaGeom { int best_h, best_w, new_h, new_w, top, left; };
static LlavaGeom llava_geometry(int height, int width, const int32_t* pin, int n_pin) {
    if (height < 1 || width < 1 || !pin || n_pin < 1) throw std::runtime_error("llava_image: bad size or pinpoints");
    LlavaGeom g{};
    long long max_eff = 0, min_waste = -1;
    bool have = false;
    for (int i = 0; i < n_pin; ++i) {
        const int h = pin[2 * i], w = pin[2 * i + 1];
        if (h < CROP || w < CROP || h % CROP || w % CROP || h > 64 * CROP || w > 64 * CROP) throw std::runtime_error("llava_image: pinpoints must be multiples of 336");
        const double sw = (double)w / width, sh = (double)h / height;
        const double scale = sw < sh ? sw : sh;
        const long long dw = (long long)(width * scale), dh = (long long)(height * scale);
        const long long eff = std::min(dw * dh, (long long)width * height);
        const long long waste = (long long)w * h - eff;
        if (!have || eff > max_eff || (eff == max_eff && waste < min_waste)) {
            max_eff = eff; min_waste = waste; g.best_h = h; g.best_w = w; have = true;
        }
    }
    const double scale_w = (double)g.best_w / width, scale_h = (double)g.best_h / height;
    if (scale_w < scale_h) {
        g.new_w = g.best_w;
        g.new_h = std::min((int)std::ceil(height * scale_w), g.best_h);
    } else {
        g.new_h = g.best_h;
        g.new_w = std::min((int)std::ceil(width * scale_h), g.best_w);
    }
    g.top = (g.best_h - g.new_h) / 2;
    g.left = (g.best_w - g.new_w) / 2;
    return g;
}

}  // namespace lr

using namespace lr;

extern "C" {

size_t lr_hd_transform_workspace(int height, int width, int num_crops) {
    try {
        const HdGeom g = hd_geometry(height, width, num_crops);
        return resize_layout(g.IH, g.IW, g.new_h, g.new_w, FILTER_BILINEAR).total;
    }
    catch (const std::exception& ex) { g_create_error = ex.what(); return 0; }
}

int lr_hd_transform(const uint8_t* rgb, int height, int width, int num_crops, float* pixel_values, int64_t* image_size,
                    int32_t* num_img_tokens, void* workspace, size_t workspace_bytes, void* hip_stream) {
    try {
        if (!rgb || !pixel_values || !workspace) throw std::runtime_error("hd_transform: null pointer");
        const HdGeom g = hd_geometry(height, width, num_crops);
        const ResizeLayout l = resize_layout(g.IH, g.IW, g.new_h, g.new_w, FILTER_BILINEAR);
        if (workspace_bytes < l.total) throw std::runtime_error("hd_transform: workspace too small (ask lr_hd_transform_workspace)");
        hipStream_t st = (hipStream_t)hip_stream;
        // source in resize orientation: a portrait image is walked transposed instead of being copied
        U8View v{rgb, g.trans ? 1 : width, g.trans ? width : 1};
        v = enqueue_resize(v, g.IH, g.IW, g.new_h, g.new_w, FILTER_BILINEAR, (char*)workspace, l, st);
        HdImage im{v, g.new_h, g.new_w, g.top, g.trans, g.trans ? g.new_w : g.tar, g.trans ? g.tar : g.new_w};
        const int per_crop4 = 3 * CROP * CROP / 4;
        hipLaunchKernelGGL(hd_tile_kernel, dim3((per_crop4 + 255) / 256, num_crops), dim3(256), 0, st, im, pixel_values, num_crops + 1);
        hipLaunchKernelGGL(hd_global_kernel, dim3((CROP + 63) / 64, CROP, 3), dim3(64), 0, st, im, pixel_values);
        LR_HIP_CHECK(hipGetLastError());
        if (image_size) { image_size[0] = im.H; image_size[1] = im.W; }
        if (num_img_tokens) *num_img_tokens = ((im.H / CROP) * (im.W / CROP) + 1) * 144 + 1 + (im.H / CROP + 1) * 12;
        return LR_OK;
    } catch (const std::exception& ex) { g_create_error = ex.what(); return LR_EINVAL; }
}

int lr_qwen_image_grid(int height, int width, int64_t min_pixels, int64_t max_pixels, int64_t* grid_thw) {
    try {
        int oh, ow;
        qwen_smart_resize(height, width, min_pixels, max_pixels, &oh, &ow);
        if (grid_thw) { grid_thw[0] = 1; grid_thw[1] = oh / 14; grid_thw[2] = ow / 14; }
        return LR_OK;
    } catch (const std::exception& ex) { g_create_error = ex.what(); return LR_EINVAL; }
}

size_t lr_qwen_image_workspace(int height, int width, int64_t min_pixels, int64_t max_pixels) {
    try {
        int oh, ow;
        qwen_smart_resize(height, width, min_pixels, max_pixels, &oh, &ow);
        return resize_layout(height, width, oh, ow, FILTER_BICUBIC).total;
    } catch (const std::exception& ex) { g_create_error = ex.what(); return 0; }
}

int lr_qwen_image_transform(const uint8_t* rgb, int height, int width, int64_t min_pixels, int64_t max_pixels, float* pixel_values,
                            int64_t* grid_thw, void* workspace, size_t workspace_bytes, void* hip_stream) {
    try {
        if (!rgb || !pixel_values || !workspace) throw std::runtime_error("qwen_image: null pointer");
        int oh, ow;
        qwen_smart_resize(height, width, min_pixels, max_pixels, &oh, &ow);
        const ResizeLayout l = resize_layout(height, width, oh, ow, FILTER_BICUBIC);
        if (workspace_bytes < l.total) throw std::runtime_error("qwen_image: workspace too small (ask lr_qwen_image_workspace)");
        hipStream_t st = (hipStream_t)hip_stream;
        U8View v{rgb, width, 1};
        v = enqueue_resize(v, height, width, oh, ow, FILTER_BICUBIC, (char*)workspace, l, st);
        const int gh = oh / 14, gw = ow / 14;
        hipLaunchKernelGGL(qwen_patchify_kernel, dim3((3 * 196 + 63) / 64, gh * gw), dim3(64), 0, st, v, pixel_values, gh, gw);
        LR_HIP_CHECK(hipGetLastError());
        if (grid_thw) { grid_thw[0] = 1; grid_thw[1] = gh; grid_thw[2] = gw; }
        return LR_OK;
    } catch (const std::exception& ex) { g_create_error = ex.what(); return LR_EINVAL; }
}

int lr_llava_image_geometry(int height, int width, const int32_t* pinpoints, int n_pinpoints, int32_t* out5) {
    try {
        const LlavaGeom g = llava_geometry(height, width, pinpoints, n_pinpoints);
        if (out5) { out5[0] = g.best_h; out5[1] = g.best_w; out5[2] = g.new_h; out5[3] = g.new_w; out5[4] = 1 + (g.best_h / CROP) * (g.best_w / CROP); }
        return LR_OK;
    } catch (const std::exception& ex) { g_create_error = ex.what(); return LR_EINVAL; }
}

size_t lr_llava_image_workspace(int height, int width, const int32_t* pinpoints, int n_pinpoints) {
    try {
        const LlavaGeom g = llava_geometry(height, width, pinpoints, n_pinpoints);
        return resize_layout(height, width, CROP, CROP, FILTER_BICUBIC).total + resize_layout(height, width, g.new_h, g.new_w, FILTER_BICUBIC).total;
    } catch (const std::exception& ex) { g_create_error = ex.what(); return 0; }
}

int lr_llava_image_transform(const uint8_t* rgb, int height, int width, const int32_t* pinpoints, int n_pinpoints, int max_crops,
                             float* pixel_values, int64_t* image_size, void* workspace, size_t workspace_bytes, void* hip_stream) {
    try {
        if (!rgb || !pixel_values || !workspace) throw std::runtime_error("llava_image: null pointer");
        const LlavaGeom g = llava_geometry(height, width, pinpoints, n_pinpoints);
        const int gh = g.best_h / CROP, gw = g.best_w / CROP;
        if (max_crops < 1 + gh * gw) throw std::runtime_error("llava_image: max_crops is smaller than 1 + grid crops of this image");
        const ResizeLayout lb = resize_layout(height, width, CROP, CROP, FILTER_BICUBIC);
        const ResizeLayout lh = resize_layout(height, width, g.new_h, g.new_w, FILTER_BICUBIC);
        if (workspace_bytes < lb.total + lh.total) throw std::runtime_error("llava_image: workspace too small (ask lr_llava_image_workspace)");
        hipStream_t st = (hipStream_t)hip_stream;
        const U8View src{rgb, width, 1};
        LlavaImage im{};
        im.base = enqueue_resize(src, height, width, CROP, CROP, FILTER_BICUBIC, (char*)workspace, lb, st);
        im.hi = enqueue_resize(src, height, width, g.new_h, g.new_w, FILTER_BICUBIC, (char*)workspace + lb.total, lh, st);
        im.new_h = g.new_h; im.new_w = g.new_w; im.top = g.top; im.left = g.left; im.gh = gh; im.gw = gw;
        const int per_crop4 = 3 * CROP * CROP / 4;
        hipLaunchKernelGGL(llava_tile_kernel, dim3((per_crop4 + 255) / 256, max_crops), dim3(256), 0, st, im, pixel_values, max_crops);
        LR_HIP_CHECK(hipGetLastError());
        if (image_size) { image_size[0] = height; image_size[1] = width; }
        return LR_OK;
    } catch (const std::exception& ex) { g_create_error = ex.what(); return LR_EINVAL; }
}

}  // extern "C"
