// Phi-3.5-V image hand-over on the GPU: uint8 RGB image in HBM -> pixel_values [num_crops+1, 3, 336, 336]
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

#pragma clang fp contract(off)
__global__ void resample_table_kernel(int in_size, int out_size, int ksize, int* __restrict__ bounds, int* __restrict__ kk) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    if (xx >= out_size) return;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;                  // bilinear: support 1
    const double ss = 1.0 / filterscale;
    const double center = 0.0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    int* k = kk + (size_t)xx * ksize;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double a = (x + xmin - center + 0.5) * ss;
        if (a < 0.0) a = -a;
        ww += a < 1.0 ? 1.0 - a : 0.0;
    }
    for (int x = 0; x < ksize; ++x) {
        double w = 0.0;
        if (x < xmax) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            w = a < 1.0 ? 1.0 - a : 0.0;
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
#pragma clang fp contract(fast)

// ---- host side: geometry of HD_transform in the same double arithmetic as the reference's Python ----
struct HdGeom { int trans, IH, IW, new_h, new_w, top, tar, ksx, ksy; };

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
    auto ks = [](int in, int out) { const double s = (double)in / out; return (int)std::ceil(s < 1.0 ? 1.0 : s) * 2 + 1; };
    g.ksx = ks(g.IW, g.new_w);
    g.ksy = ks(g.IH, g.new_h);
    return g;
}

static size_t al256(size_t n) { return (n + 255) & ~(size_t)255; }

struct HdLayout { size_t bx, kx, by, ky, pass1, pass2, total; };
static HdLayout hd_layout(const HdGeom& g) {
    HdLayout l{};
    size_t o = 0;
    l.bx = o; o += al256((size_t)g.new_w * 2 * 4);
    l.kx = o; o += al256((size_t)g.new_w * g.ksx * 4);
    l.by = o; o += al256((size_t)g.new_h * 2 * 4);
    l.ky = o; o += al256((size_t)g.new_h * g.ksy * 4);
    l.pass1 = o; o += al256((size_t)g.IH * g.new_w * 3);
    l.pass2 = o; o += al256((size_t)g.new_h * g.new_w * 3);
    l.total = o;
    return l;
}

}  // namespace lr

using namespace lr;

extern "C" {

size_t lr_hd_transform_workspace(int height, int width, int num_crops) {
    try { return hd_layout(hd_geometry(height, width, num_crops)).total; }
    catch (const std::exception& ex) { g_create_error = ex.what(); return 0; }
}

int lr_hd_transform(const uint8_t* rgb, int height, int width, int num_crops, float* pixel_values, int64_t* image_size,
                    int32_t* num_img_tokens, void* workspace, size_t workspace_bytes, void* hip_stream) {
    try {
        if (!rgb || !pixel_values || !workspace) throw std::runtime_error("hd_transform: null pointer");
        const HdGeom g = hd_geometry(height, width, num_crops);
        const HdLayout l = hd_layout(g);
        if (workspace_bytes < l.total) throw std::runtime_error("hd_transform: workspace too small (ask lr_hd_transform_workspace)");
        hipStream_t st = (hipStream_t)hip_stream;
        char* ws = (char*)workspace;
        // source in resize orientation: a portrait image is walked transposed instead of being copied
        U8View v{rgb, g.trans ? 1 : width, g.trans ? width : 1};
        if (g.new_w != g.IW) {
            int* bx = (int*)(ws + l.bx); int* kx = (int*)(ws + l.kx);
            hipLaunchKernelGGL(resample_table_kernel, dim3((g.new_w + 127) / 128), dim3(128), 0, st, g.IW, g.new_w, g.ksx, bx, kx);
            unsigned char* d = (unsigned char*)(ws + l.pass1);
            hipLaunchKernelGGL(resample_kernel<1>, dim3((g.new_w + 127) / 128, g.IH), dim3(128), 0, st, v, d, g.IH, g.new_w, bx, kx, g.ksx);
            v = U8View{d, g.new_w, 1};
        }
        if (g.new_h != g.IH) {
            int* by = (int*)(ws + l.by); int* ky = (int*)(ws + l.ky);
            hipLaunchKernelGGL(resample_table_kernel, dim3((g.new_h + 127) / 128), dim3(128), 0, st, g.IH, g.new_h, g.ksy, by, ky);
            unsigned char* d = (unsigned char*)(ws + l.pass2);
            hipLaunchKernelGGL(resample_kernel<0>, dim3((g.new_w + 127) / 128, g.new_h), dim3(128), 0, st, v, d, g.new_h, g.new_w, by, ky, g.ksy);
            v = U8View{d, g.new_w, 1};
        }
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

}  // extern "C"
