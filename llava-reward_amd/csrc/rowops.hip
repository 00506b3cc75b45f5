// HBM-bound row kernels of the scoring path: norms, patchify, embedding scatter, RoPE, HD gather,
// the fp32 reward tail, and weight packing / synthesis.  One wave (64 lanes) per row wherever a
// row reduction is needed; 16-byte fp32 loads and 8-byte operand stores.
#include <cmath>
#include <cstring>

#include "common.h"
#include "kernels.h"

namespace lr {

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

template <typename OT> __device__ __forceinline__ void store4(void* base, size_t elem, float a, float b, float c, float d) {
    uint2 w;
    w.x = pack2<OT>(a, b);
    w.y = pack2<OT>(c, d);
    *(uint2*)((unsigned short*)base + elem) = w;
}

// split-operand mode (prec = 1): an operand row of width W is stored 2W wide, [hi | lo] (common.h split2)
template <typename OT> __device__ __forceinline__ void store4s(void* base, size_t row, int W, int prec, int col, float a, float b,
                                                               float c, float d) {
    unsigned short* dst = (unsigned short*)base + row * (size_t)(W << prec) + col;
    uint2 h, l;
    split2<OT>(a, b, h.x, l.x);
    split2<OT>(c, d, h.y, l.y);
    *(uint2*)dst = h;
    if (prec) *(uint2*)(dst + W) = l;
}


// ------------------------------------------------------------------------------------------ norms
// modeling_phi3_v.py:377-391 (RMSNorm: w * (x * rsqrt(mean(x^2) + eps))) and CLIP LayerNorm.
constexpr int NORM_MAXC = 16;   // H <= 4096

template <typename OT, bool LAYERNORM>
__global__ __launch_bounds__(256) void norm_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, void* __restrict__ y, int rows,
                                                        int H, float eps, int prec, int group) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nch = H >> 2;
    const float4* xr = (const float4*)(x + (size_t)row * H);
    float4 v[NORM_MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NORM_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            v[i] = xr[c];
            s += LAYERNORM ? (v[i].x + v[i].y + v[i].z + v[i].w)
                           : (v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w);
        }
    }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (LAYERNORM) {
        mean = s / H;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NORM_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
                q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
            }
        }
        q = wave_sum(q);
        rstd = rsqrtf(q / H + eps);
    } else {
        rstd = rsqrtf(s / H + eps);
    }
#pragma unroll
    for (int i = 0; i < NORM_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const float4 ww = ((const float4*)w)[c];
            float o0 = (v[i].x - mean) * rstd * ww.x, o1 = (v[i].y - mean) * rstd * ww.y;
            float o2 = (v[i].z - mean) * rstd * ww.z, o3 = (v[i].w - mean) * rstd * ww.w;
            if (LAYERNORM) {
                const float4 bb = ((const float4*)b)[c];
                o0 += bb.x; o1 += bb.y; o2 += bb.z; o3 += bb.w;
            }
            store4s<OT>(y, row / group, H * group, prec, (row % group) * H + 4 * c, o0, o1, o2, o3);
        }
    }
}

// The same norm with 8 consecutive elements per lane and iteration (H % 8 == 0, group == 1: every norm of the three backbones at
// their real widths): 32 contiguous bytes read, 16 bytes of hi and 16 (16-bit) or 8 (e4m3) bytes of residuals stored per lane,
// instead of 8- and 4-byte stores (cdna_hip_programming.md Guideline 13).  HBM-bound: x fp32 in, [hi | lo] out.
template <typename OT, bool LAYERNORM>
__global__ __launch_bounds__(256) void norm_rows8_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, void* __restrict__ y, int rows,
                                                         int H, float eps, int prec, unsigned char* __restrict__ lo8) {
    constexpr int MAXI = NORM_MAXC / 2;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int n8 = H >> 3;
    const float4* xr = (const float4*)(x + (size_t)row * H);
    float v[MAXI][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = lane + 64 * i;
        if (c < n8) {
            const float4 a0 = xr[2 * c], a1 = xr[2 * c + 1];
            v[i][0] = a0.x; v[i][1] = a0.y; v[i][2] = a0.z; v[i][3] = a0.w; v[i][4] = a1.x; v[i][5] = a1.y; v[i][6] = a1.z; v[i][7] = a1.w;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += LAYERNORM ? v[i][j] : v[i][j] * v[i][j];
        }
    }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (LAYERNORM) {
        mean = s / H;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
            if (lane + 64 * i < n8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float a = v[i][j] - mean; q += a * a; }
            }
        q = wave_sum(q);
        rstd = rsqrtf(q / H + eps);
    } else {
        rstd = rsqrtf(s / H + eps);
    }
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = lane + 64 * i;
        if (c < n8) {
            const float4 w0 = ((const float4*)w)[2 * c], w1 = ((const float4*)w)[2 * c + 1];
            const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = (v[i][j] - mean) * rstd * ww[j];
            if (LAYERNORM) {
                const float4 b0 = ((const float4*)b)[2 * c], b1 = ((const float4*)b)[2 * c + 1];
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] += bb[j];
            }
        }
    }
    unsigned short* dst = (unsigned short*)y + (size_t)row * ((size_t)H << prec);
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = lane + 64 * i;
        unsigned short hb[8];
        float r[8];
        float m = 0.f;
        if (c < n8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { hb[j] = Op<OT>::from_f32(v[i][j]); r[j] = v[i][j] - Op<OT>::to_f32(hb[j]); m = fmaxf(m, fabsf(r[j])); }
            *(uint4*)(dst + 8 * c) = make_uint4(hb[0] | ((unsigned)hb[1] << 16), hb[2] | ((unsigned)hb[3] << 16),
                                                hb[4] | ((unsigned)hb[5] << 16), hb[6] | ((unsigned)hb[7] << 16));
        }
        if (lo8) {              // residual half as block-scaled e4m3 (common.h): 16 lanes = one 128-column block (H % 128 == 0)
#if LR_EMU_FP6
            {
                const float bm = quad_max(c < n8 ? m : 0.f);
                m = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { r[j] = c < n8 ? emu_e2m3(r[j], bm) : 0.f; m = fmaxf(m, fabsf(r[j])); }
            }
#endif
            const int E = e8m0_of_amax(row16_max(m));
            if (c < n8) {
                const float sc = e8m0_inv_scale(E);
                int p0 = 0, p1 = 0;
                p0 = __builtin_amdgcn_cvt_pk_fp8_f32(r[0] * sc, r[1] * sc, p0, false);
                p0 = __builtin_amdgcn_cvt_pk_fp8_f32(r[2] * sc, r[3] * sc, p0, true);
                p1 = __builtin_amdgcn_cvt_pk_fp8_f32(r[4] * sc, r[5] * sc, p1, false);
                p1 = __builtin_amdgcn_cvt_pk_fp8_f32(r[6] * sc, r[7] * sc, p1, true);
                *(uint2*)((unsigned char*)(dst + H) + 8 * c) = make_uint2((unsigned)p0, (unsigned)p1);
                if ((lane & 15) == 0) lo8[lo8_scale_at(row, c >> 4, rows)] = (unsigned char)E;
            }
        } else if (prec && c < n8) {
            *(uint4*)(dst + H + 8 * c) = make_uint4(pack2<OT>(r[0], r[1]), pack2<OT>(r[2], r[3]), pack2<OT>(r[4], r[5]), pack2<OT>(r[6], r[7]));
        }
    }
}

void launch_norm_rows(const float* x, const float* w, const float* b, void* y, int rows, int H, float eps,
                      int operand_dtype, hipStream_t st, int prec, int group, unsigned char* lo8) {
    if (rows <= 0) return;
    if (group < 1 || rows % group) throw std::runtime_error("norm_rows: rows must be a multiple of group");
    if (lo8 && (!prec || group != 1 || H % 128 || (((uintptr_t)y) & 15) || (((uintptr_t)w) & 15) || (b && (((uintptr_t)b) & 15))))
        throw std::runtime_error("norm_rows: the e4m3 residual form needs split-operand rows, group 1, H % 128 == 0 and 16-byte aligned pointers");
    if (H % 4 || H > NORM_MAXC * 256) throw std::runtime_error("norm_rows: H must be a multiple of 4 and <= 4096");
    dim3 g(cdiv(rows, 4)), t(256);
    const bool f16 = operand_dtype == DT_F16;
    if (group == 1 && H % 8 == 0 && (((uintptr_t)y) & 15) == 0 && (((uintptr_t)w) & 15) == 0 && (!b || (((uintptr_t)b) & 15) == 0)) {
        if (b) {
            if (f16) hipLaunchKernelGGL((norm_rows8_kernel<F16, true>), g, t, 0, st, x, w, b, y, rows, H, eps, prec, lo8);
            else hipLaunchKernelGGL((norm_rows8_kernel<BF16, true>), g, t, 0, st, x, w, b, y, rows, H, eps, prec, lo8);
        } else {
            if (f16) hipLaunchKernelGGL((norm_rows8_kernel<F16, false>), g, t, 0, st, x, w, b, y, rows, H, eps, prec, lo8);
            else hipLaunchKernelGGL((norm_rows8_kernel<BF16, false>), g, t, 0, st, x, w, b, y, rows, H, eps, prec, lo8);
        }
        return;
    }
    if (b) {
        if (f16) hipLaunchKernelGGL((norm_rows_kernel<F16, true>), g, t, 0, st, x, w, b, y, rows, H, eps, prec, group);
        else hipLaunchKernelGGL((norm_rows_kernel<BF16, true>), g, t, 0, st, x, w, b, y, rows, H, eps, prec, group);
    } else {
        if (f16) hipLaunchKernelGGL((norm_rows_kernel<F16, false>), g, t, 0, st, x, w, b, y, rows, H, eps, prec, group);
        else hipLaunchKernelGGL((norm_rows_kernel<BF16, false>), g, t, 0, st, x, w, b, y, rows, H, eps, prec, group);
    }
}

// ------------------------------------------------------------------------------------- patchify
// Conv2d(3,H,k=p,s=p,bias=False) as a GEMM: row = (crop, py, px), column k = c*p*p + ky*p + kx, zero
// padded to Kpad.  One block per (crop, py): reads p pixel rows of each channel, writes g patch rows.
template <typename OT, typename PT>
__global__ __launch_bounds__(256) void im2col_kernel(const PT* __restrict__ pix, const int* __restrict__ crop_src,
                                                     int img, int patch, int Kpad, void* __restrict__ out, int prec) {
    const int g = img / patch;
    const int crop = blockIdx.x / g, py = blockIdx.x % g;
    const PT* src = pix + (size_t)crop_src[crop] * 3 * img * img;
    const int K = 3 * patch * patch;
    const int ld = Kpad << prec;
    unsigned short* dst = (unsigned short*)out + ((size_t)crop * g * g + (size_t)py * g) * ld;
    for (int idx = threadIdx.x; idx < g * Kpad; idx += 256) {
        const int px = idx / Kpad, k = idx - px * Kpad;
        float v = 0.f;
        if (k < K) {
            const int c = k / (patch * patch), rem = k - c * patch * patch;
            const int ky = rem / patch, kx = rem - ky * patch;
            v = (float)src[((size_t)c * img + (py * patch + ky)) * img + px * patch + kx];
        }
        const unsigned short hv = Op<OT>::from_f32(v);
        dst[(size_t)px * ld + k] = hv;
        if (prec) dst[(size_t)px * ld + Kpad + k] = Op<OT>::from_f32(v - Op<OT>::to_f32(hv));
    }
}

void launch_im2col(const void* pixels, int pix_dtype, const int* crop_src, int ncrop, int img, int patch, int Kpad,
                   void* out, int operand_dtype, hipStream_t st, int prec) {
    if (ncrop <= 0) return;
    dim3 g(ncrop * (img / patch)), t(256);
    const bool f16 = operand_dtype == DT_F16;
    if (pix_dtype == DT_F32) {
        if (f16) hipLaunchKernelGGL((im2col_kernel<F16, float>), g, t, 0, st, (const float*)pixels, crop_src, img, patch, Kpad, out, prec);
        else hipLaunchKernelGGL((im2col_kernel<BF16, float>), g, t, 0, st, (const float*)pixels, crop_src, img, patch, Kpad, out, prec);
    } else if (pix_dtype == DT_BF16) {
        if (f16) hipLaunchKernelGGL((im2col_kernel<F16, __bf16>), g, t, 0, st, (const __bf16*)pixels, crop_src, img, patch, Kpad, out, prec);
        else hipLaunchKernelGGL((im2col_kernel<BF16, __bf16>), g, t, 0, st, (const __bf16*)pixels, crop_src, img, patch, Kpad, out, prec);
    } else {
        throw std::runtime_error("im2col: pixel dtype must be f32 or bf16");
    }
}

// --------------------------------------------------------------- CLIP embeddings + pre_layrnorm
__global__ __launch_bounds__(256) void clip_embed_kernel(const float* __restrict__ patch_out, const float* __restrict__ cls,
                                                         const float* __restrict__ pos, const float* __restrict__ lnw,
                                                         const float* __restrict__ lnb, float* __restrict__ x, int rows,
                                                         int T, int H, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int crop = row / T, t = row - crop * T;
    const int nch = H >> 2;
    const float4* src = t ? (const float4*)(patch_out + ((size_t)crop * (T - 1) + (t - 1)) * H) : (const float4*)cls;
    const float4* pp = (const float4*)(pos + (size_t)t * H);
    float4 v[NORM_MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NORM_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const float4 a = src[c], p = pp[c];
            v[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
            s += v[i].x + v[i].y + v[i].z + v[i].w;
        }
    }
    const float mean = wave_sum(s) / H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NORM_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / H + eps);
    float4* dst = (float4*)(x + (size_t)row * H);
#pragma unroll
    for (int i = 0; i < NORM_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const float4 ww = ((const float4*)lnw)[c], bb = ((const float4*)lnb)[c];
            dst[c] = make_float4((v[i].x - mean) * rstd * ww.x + bb.x, (v[i].y - mean) * rstd * ww.y + bb.y,
                                 (v[i].z - mean) * rstd * ww.z + bb.z, (v[i].w - mean) * rstd * ww.w + bb.w);
        }
    }
}

void launch_clip_embed(const float* patch_out, const float* cls, const float* pos, const float* lnw, const float* lnb,
                       float* x, int ncrop, int T, int H, float eps, hipStream_t st) {
    const int rows = ncrop * T;
    if (rows <= 0) return;
    hipLaunchKernelGGL(clip_embed_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, patch_out, cls, pos, lnw, lnb, x, rows, T, H, eps);
}

// ------------------------------------------------------------------------------------ token plan
// rw_model_general_preference.py:344-345 (position ids), modeling_phi3_v.py:228-249 (image slots in
// row-major order), rw_model:420/439 (last valid index).  tstat[b] = {last_valid, first_valid, n_img, n_valid}.
__global__ __launch_bounds__(256) void token_plan_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ mask,
                                                         int S, const int* __restrict__ voff, int* __restrict__ img_row,
                                                         int* __restrict__ pos, int* __restrict__ tstat, long image_token_id,
                                                         int pos_arange) {
    __shared__ int wsum[2][4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int run_m = 0, run_n = 0, last = -1, first = S;
    for (int base = 0; base < S; base += 256) {
        const int s = base + tid;
        bool m = false, n = false;
        if (s < S) {
            m = mask[(size_t)b * S + s] != 0;
            const int64_t id = ids[(size_t)b * S + s];
            n = image_token_id >= 0 ? (id == image_token_id) : (id < 0 && id > -1000000000LL);
        }
        const unsigned long long bm = __ballot(m), bn = __ballot(n);
        const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const int pm = __popcll(bm & lt), pn = __popcll(bn & lt);
        if (lane == 0) { wsum[0][wave] = __popcll(bm); wsum[1][wave] = __popcll(bn); }
        __syncthreads();
        int om = run_m, on = run_n, tm = 0, tn = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) { om += wsum[0][w]; on += wsum[1][w]; }
            tm += wsum[0][w]; tn += wsum[1][w];
        }
        if (s < S) {
            pos[(size_t)b * S + s] = pos_arange ? s : (m ? (om + pm) : 1);     // arange | cumsum(mask) - 1, pads -> 1
            // a slot beyond the image tokens this row's image_sizes produce would read another row's features (or past the buffer):
            // it keeps the text embedding instead, and slot_check_kernel turns the row's reward into NaN
            img_row[(size_t)b * S + s] = (n && on + pn < voff[b + 1] - voff[b]) ? (voff[b] + on + pn) : -1;
            if (m) { last = s; if (s < first) first = s; }
        }
        run_m += tm; run_n += tn;
        __syncthreads();
    }
    // block reduce last (max) / first (min)
    __shared__ int rl[256], rf[256];
    rl[tid] = last; rf[tid] = first;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { rl[tid] = max(rl[tid], rl[tid + o]); rf[tid] = min(rf[tid], rf[tid + o]); }
        __syncthreads();
    }
    if (tid == 0) {
        tstat[b * 4 + 0] = rl[0] >= 0 ? rl[0] : S - 1;   // S-1-argmax(flip(mask)); all-zero mask -> S-1
        tstat[b * 4 + 1] = rf[0];
        tstat[b * 4 + 2] = run_n;
        tstat[b * 4 + 3] = run_m;
    }
}

void launch_token_plan(const int64_t* ids, const int64_t* mask, int B, int S, const int* voff, int* img_row, int* pos,
                       int* tstat, hipStream_t st, long image_token_id, int pos_arange) {
    if (B <= 0) return;
    hipLaunchKernelGGL(token_plan_kernel, dim3(B), dim3(256), 0, st, ids, mask, S, voff, img_row, pos, tstat, image_token_id, pos_arange);
}

// The reference fails such a batch (index_put shape mismatch, modeling_phi3_v.py:247; "Image features and image tokens do not
// match", modeling_llava_next.py); the Python wrapper raises the same errors before the launch.  For direct C-ABI callers, who
// get no host-side check (it would cost a device sync per pass), a row whose image-slot count differs from the number of image
// tokens its image_sizes produce comes back as NaN instead of a silently wrong reward.
__global__ void slot_check_kernel(const int* __restrict__ tstat, const int* __restrict__ voff, float* __restrict__ rewards, int B, int d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * d) return;
    const int b = i / d;
    if (tstat[b * 4 + 2] != voff[b + 1] - voff[b]) rewards[i] = __builtin_nanf("");
}
void launch_slot_check(const int* tstat, const int* voff, float* rewards, int B, int d, hipStream_t st) {
    if (B <= 0) return;
    hipLaunchKernelGGL(slot_check_kernel, dim3(cdiv(B * d, 64)), dim3(64), 0, st, tstat, voff, rewards, B, d);
}

// --------------------------------------------------------------------------------------- embedding
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ ids, const int* __restrict__ img_row,
                                                    const unsigned short* __restrict__ wte, const float* __restrict__ ev,
                                                    float* __restrict__ x, int rows, int D, int vocab) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float4* dst = (float4*)(x + (size_t)row * D);
    const int ir = img_row[row];
    const int nch = D >> 2;
    if (ir >= 0) {
        const float4* src = (const float4*)(ev + (size_t)ir * D);
        for (int c = lane; c < nch; c += 64) dst[c] = src[c];
    } else {
        int64_t id = ids[row];
        id = id < 0 ? 0 : (id > vocab - 1 ? vocab - 1 : id);
        const uint2* src = (const uint2*)(wte + (size_t)id * D);
        for (int c = lane; c < nch; c += 64) {
            const uint2 u = src[c];
            dst[c] = make_float4(bf16_bits_to_f32(u.x & 0xFFFF), bf16_bits_to_f32(u.x >> 16),
                                 bf16_bits_to_f32(u.y & 0xFFFF), bf16_bits_to_f32(u.y >> 16));
        }
    }
}

void launch_embed(const int64_t* ids, const int* img_row, const unsigned short* wte_bf16, const float* ev, float* x,
                  int rows, int D, int vocab, hipStream_t st) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(embed_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, ids, img_row, wte_bf16, ev, x, rows, D, vocab);
}

// ------------------------------------------------------------------------------------------ RoPE
// modeling_phi3_v.py:446-476: freqs = pos * inv_freq; cos/sin scaled by sqrt(1 + ln(s)/ln(orig)); long factors when
// seq_len > original_max_position_embeddings.  The attention layers call it with seq_len = kv_seq_len = the PADDED length S
// (:673 eager, :1081 sdpa; no KV cache on this path), so the `seq_len or max(position_ids)+1` fallback of :448 is never taken:
// the switch depends on S alone, not on how many tokens of a row are valid.  This is the EAGER / SDPA convention (the oracle's and
// the goldens' attention implementation).  Phi3FlashAttention2 (:793-794; flash-attn is absent here, so it is unpinned) passes
// seq_len = max(kv_seq_len, max(position_ids)) + 1 and therefore switches one token earlier, at S >= original_max: at S ==
// original_max exactly (4096 for Phi-3.5-V) a flash-attention reference uses the long factors where eager / sdpa still use the
// short ones (golden ref_small_rope_at_orig_bt_ca pins the eager side of that boundary).  lr_model_desc.rope_flash_convention selects
// the flash side: the launcher then passes rotary_seq_len = S + 1.  cs layout: [row][half][2].
__global__ __launch_bounds__(256) void rope_table_kernel(const int* __restrict__ pos, int S,          // S = the seq_len the rotary module sees
                                                         int rows, const float* __restrict__ inv_s,
                                                         const float* __restrict__ inv_l, float scaling, int orig_max,
                                                         int half, float* __restrict__ cs) {
    const float* inv = (S > orig_max) ? inv_l : inv_s;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * half) return;
    const int row = (int)(i / half), k = (int)(i - (size_t)row * half);
    const float ang = (float)pos[row] * inv[k];
    cs[((size_t)row * half + k) * 2] = cosf(ang) * scaling;
    cs[((size_t)row * half + k) * 2 + 1] = sinf(ang) * scaling;
}

void launch_rope_table(const int* pos, const int* tstat, int B, int S, const float* inv_freq_short,
                       const float* inv_freq_long, float scaling, int orig_max_pos, int half, float* cs, hipStream_t st, int rotary_seq_len) {
    const int rows = B * S;
    (void)tstat;
    if (rows <= 0) return;
    hipLaunchKernelGGL(rope_table_kernel, dim3(cdiv((long)rows * half, 256)), dim3(256), 0, st, pos, rotary_seq_len > 0 ? rotary_seq_len : S, rows,
                       inv_freq_short, inv_freq_long, scaling, orig_max_pos, half, cs);
}

// modeling_phi3_v.py:521-553: q' = q*cos + rotate_half(q)*sin.  The engine stores q and k with their head
// dims PAIR-INTERLEAVED (weight rows permuted at upload: reference dims i and i+hd/2 sit at 2i, 2i+1), which
// leaves q.k unchanged; so here (x0, x1) = (q[i], q[i+hd/2]) -> (x0 c - x1 s, x1 c + x0 s).
// Fallback path for problems too small for the GEMM with the fused RoPE epilogue.
template <typename OT>
__global__ __launch_bounds__(256) void rope_split_kernel(const float* __restrict__ qkv, const float* __restrict__ cs,
                                                         void* __restrict__ out, int rope_cols, int v_cols, int hd, int prec) {
    const int row = blockIdx.x;
    const int half = hd >> 1, ld = rope_cols + v_cols;
    const float* src = qkv + (size_t)row * ld;
    const float4* t = (const float4*)(cs + (size_t)row * 2 * half);      // (c0,s0,c1,s1) per float4
    for (int it = threadIdx.x; it < (rope_cols >> 2); it += 256) {        // 4 columns = 2 pairs
        const int col = 4 * it;
        const int i0 = (col % hd) >> 1;
        const float4 x = *(const float4*)(src + col);
        const float4 c = t[i0 >> 1];
        float4 r;
        rope_pair(x.x, x.y, c.x, c.y, r.x, r.y);
        rope_pair(x.z, x.w, c.z, c.w, r.z, r.w);
        store4s<OT>(out, row, ld, prec, col, r.x, r.y, r.z, r.w);
    }
    for (int c = threadIdx.x; c < (v_cols >> 2); c += 256) {
        const float4 v = *(const float4*)(src + rope_cols + 4 * c);
        store4s<OT>(out, row, ld, prec, rope_cols + 4 * c, v.x, v.y, v.z, v.w);
    }
}

void launch_rope_split(const float* qkv32, const float* cs, void* out, int rows, int rope_cols, int v_cols, int hd,
                       int operand_dtype, hipStream_t st, int prec) {
    if (rows <= 0) return;
    if (hd % 8 || rope_cols % hd || v_cols % 4) throw std::runtime_error("rope_split: bad geometry");
    if (operand_dtype == DT_F16) hipLaunchKernelGGL((rope_split_kernel<F16>), dim3(rows), dim3(256), 0, st, qkv32, cs, out, rope_cols, v_cols, hd, prec);
    else hipLaunchKernelGGL((rope_split_kernel<BF16>), dim3(rows), dim3(256), 0, st, qkv32, cs, out, rope_cols, v_cols, hd, prec);
}

// ------------------------------------------------------------------------------------- HD gather
// modeling_phi3_v.py:254-362.  Output row layout per sample: (hc*g2) rows of (wc*g2 patches + sub_GN),
// then glb_GN, then g2 rows of (g2 patches + sub_GN) of the global crop; a patch row is the 2x2
// neighbourhood (2i+di, 2j+dj) concatenated as (di, dj, channel).
template <typename OT>
__global__ __launch_bounds__(256) void hd_gather_kernel(const float* __restrict__ clipx, const HdSample* __restrict__ smp,
                                                        int B, int total_rows, int T, int H, int g,
                                                        const float* __restrict__ sub_gn, const float* __restrict__ glb_gn,
                                                        void* __restrict__ out, int prec) {
    const int R = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (R >= total_rows) return;
    int b = 0;
    while (b + 1 < B && smp[b + 1].voff <= R) ++b;
    const HdSample sm = smp[b];
    const int g2 = g / 2;
    const int local = R - sm.voff;
    const int wrow = sm.wc * g2 + 1;
    const int nsub = sm.hc * g2 * wrow;
    const float* special = nullptr;
    int crop = 0, pi = 0, pj = 0;
    if (local < nsub) {
        const int i = local / wrow, j = local - i * wrow;
        if (j == sm.wc * g2) special = sub_gn;
        else { crop = sm.crop0 + 1 + (i / g2) * sm.wc + (j / g2); pi = i % g2; pj = j % g2; }
    } else if (local == nsub) {
        special = glb_gn;
    } else {
        const int q = local - nsub - 1;
        const int i = q / (g2 + 1), j = q - i * (g2 + 1);
        if (j == g2) special = sub_gn;
        else { crop = sm.crop0; pi = i; pj = j; }
    }
    const int nch = H;          // 4H values / 4 per chunk
    for (int c = lane; c < nch; c += 64) {
        float4 v;
        if (special) v = ((const float4*)special)[c];
        else {
            const int blk = (4 * c) / H, within = 4 * c - blk * H;
            const int t = 1 + (2 * pi + (blk >> 1)) * g + (2 * pj + (blk & 1));
            v = *(const float4*)(clipx + ((size_t)crop * T + t) * H + within);
        }
        store4s<OT>(out, R, 4 * H, prec, 4 * c, v.x, v.y, v.z, v.w);
    }
}

void launch_hd_gather(const float* clipx, const HdSample* samples, int B, int total_rows, int T, int H,
                      const float* sub_gn, const float* glb_gn, void* out, int operand_dtype, hipStream_t st, int prec) {
    if (total_rows <= 0) return;
    int g = 1;
    while (g * g + 1 < T) ++g;
    dim3 gr(cdiv(total_rows, 4)), t(256);
    if (operand_dtype == DT_F16) hipLaunchKernelGGL((hd_gather_kernel<F16>), gr, t, 0, st, clipx, samples, B, total_rows, T, H, g, sub_gn, glb_gn, out, prec);
    else hipLaunchKernelGGL((hd_gather_kernel<BF16>), gr, t, 0, st, clipx, samples, B, total_rows, T, H, g, sub_gn, glb_gn, out, prec);
}

// ------------------------------------------------------------------------------------------ LLaVA
template <typename OT>
__global__ __launch_bounds__(256) void clip_tokens_kernel(const float* __restrict__ clipx, void* __restrict__ out, int rows, int T, int H, int prec) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int crop = row / (T - 1), t = row - crop * (T - 1);
    const float4* src = (const float4*)(clipx + ((size_t)crop * T + 1 + t) * H);
    for (int c = lane; c < (H >> 2); c += 64) {
        const float4 v = src[c];
        store4s<OT>(out, row, H, prec, 4 * c, v.x, v.y, v.z, v.w);
    }
}

void launch_clip_tokens(const float* clipx, void* out, int ncrop, int T, int H, int operand_dtype, hipStream_t st, int prec) {
    const int rows = ncrop * (T - 1);
    if (rows <= 0) return;
    if (operand_dtype == DT_F16) hipLaunchKernelGGL((clip_tokens_kernel<F16>), dim3(cdiv(rows, 4)), dim3(256), 0, st, clipx, out, rows, T, H, prec);
    else hipLaunchKernelGGL((clip_tokens_kernel<BF16>), dim3(cdiv(rows, 4)), dim3(256), 0, st, clipx, out, rows, T, H, prec);
}

// modeling_llava_next.py:265-335 pack_image_features ("spatial_unpad"): per image [base crop; un-padded grid + newline column]
__global__ __launch_bounds__(256) void llava_pack_kernel(const float* __restrict__ proj, const LlavaSample* __restrict__ smp, int B,
                                                         int total_rows, int g, int D, const float* __restrict__ newline,
                                                         float* __restrict__ ev) {
    const int R = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (R >= total_rows) return;
    int b = 0;
    while (b + 1 < B && smp[b + 1].voff <= R) ++b;
    const LlavaSample sm = smp[b];
    const int local = R - sm.voff, gg = g * g;
    const float* src;
    if (local < gg) {
        src = proj + ((size_t)sm.crop0 * gg + local) * D;
    } else {
        const int q = local - gg, wrow = sm.c1 - sm.c0 + 1;
        const int y = sm.r0 + q / wrow, xx = q % wrow;
        if (xx == wrow - 1) src = newline;
        else {
            const int x = sm.c0 + xx;
            const int crop = sm.crop0 + 1 + (y / g) * sm.gw + x / g;
            src = proj + ((size_t)crop * gg + (y % g) * g + (x % g)) * D;
        }
    }
    float4* dst = (float4*)(ev + (size_t)R * D);
    for (int c = lane; c < (D >> 2); c += 64) dst[c] = ((const float4*)src)[c];
}

void launch_llava_pack(const float* proj, const LlavaSample* samples, int B, int total_rows, int g, int D,
                       const float* newline, float* ev, hipStream_t st) {
    if (total_rows <= 0) return;
    hipLaunchKernelGGL(llava_pack_kernel, dim3(cdiv(total_rows, 4)), dim3(256), 0, st, proj, samples, B, total_rows, g, D, newline, ev);
}

// ------------------------------------------------------------------------------------------- tail
// rw_model_general_preference.py:376-448 evaluated for the gathered (EOS) row only, all in fp32.
__global__ __launch_bounds__(256) void gather_norm_kernel(const float* __restrict__ x, const int* __restrict__ tstat, int S,
                                                          int use_last_pos, const float* __restrict__ w, float eps,
                                                          float* __restrict__ y, int B, int D) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const int idx = use_last_pos ? S - 1 : tstat[b * 4 + 0];
    const float* xr = x + ((size_t)b * S + idx) * D;
    if (!w) {          // hidden_states[layer_id] of an inner layer: the residual stream itself, no final norm (rw_model:351-352)
        for (int c = lane; c < D; c += 64) y[(size_t)b * D + c] = xr[c];
        return;
    }
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c] * xr[c];
    const float rstd = rsqrtf(wave_sum(s) / D + eps);
    for (int c = lane; c < D; c += 64) y[(size_t)b * D + c] = w[c] * (xr[c] * rstd);
}

void launch_gather_norm_rows(const float* x, const int* tstat, int S, int use_last_pos, const float* w, float eps,
                             float* y, int B, int D, hipStream_t st) {
    if (B <= 0) return;
    hipLaunchKernelGGL(gather_norm_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, x, tstat, S, use_last_pos, w, eps, y, B, D);
}

__global__ __launch_bounds__(256) void rowvec_linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                            float* __restrict__ y, int N, int K) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    const float4* xr = (const float4*)(x + (size_t)b * K);
    const float4* wr = (const float4*)(W + (size_t)n * K);
    float s = 0.f;
    for (int c = lane; c < (K >> 2); c += 64) {
        const float4 a = xr[c], w = wr[c];
        s += a.x * w.x + a.y * w.y + a.z * w.z + a.w * w.w;
    }
    s = wave_sum(s);
    if (lane == 0) y[(size_t)b * N + n] = s;
}

void launch_rowvec_linear(const float* x, const float* W, float* y, int B, int N, int K, hipStream_t st) {
    if (B <= 0) return;
    if (K % 4) throw std::runtime_error("rowvec_linear: K must be a multiple of 4");
    hipLaunchKernelGGL(rowvec_linear_kernel, dim3(cdiv(N, 4), B), dim3(256), 0, st, x, W, y, N, K);
}

// scores over the zero-padded vision rows: rows j >= V_b are all-zero keys -> score 0 (rw_model:381-385
// with modeling_phi3_v.py:245); they take part in the softmax un-masked.
__global__ __launch_bounds__(256) void ca_scores_kernel(const float* __restrict__ ev, const float* __restrict__ kq,
                                                        const int* __restrict__ voff, int Vmax, int D, float scale,
                                                        float* __restrict__ sc) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    if (j >= Vmax) return;
    const int vb = voff[b + 1] - voff[b];
    float s = 0.f;
    if (j < vb) {
        const float4* e = (const float4*)(ev + ((size_t)voff[b] + j) * D);
        const float4* q = (const float4*)(kq + (size_t)b * D);
        for (int c = lane; c < (D >> 2); c += 64) {
            const float4 a = e[c], w = q[c];
            s += a.x * w.x + a.y * w.y + a.z * w.z + a.w * w.w;
        }
        s = wave_sum(s) * scale;
    }
    if (lane == 0) sc[(size_t)b * Vmax + j] = s;
}

void launch_ca_scores(const float* ev, const float* kq, const int* voff, int B, int Vmax, int D, float scale, float* sc,
                      hipStream_t st) {
    if (B <= 0 || Vmax <= 0) return;
    hipLaunchKernelGGL(ca_scores_kernel, dim3(cdiv(Vmax, 4), B), dim3(256), 0, st, ev, kq, voff, Vmax, D, scale, sc);
}

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
    v = is_max ? wave_max(v) : wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float r = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = is_max ? fmaxf(r, sh[w]) : r + sh[w];
    return r;
}

__global__ __launch_bounds__(256) void ca_softmax_kernel(float* __restrict__ sc, int Vmax) {
    __shared__ float sh[4];
    float* r = sc + (size_t)blockIdx.x * Vmax;
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < Vmax; j += 256) mx = fmaxf(mx, r[j]);
    mx = block_reduce(mx, true, sh);
    float s = 0.f;
    for (int j = threadIdx.x; j < Vmax; j += 256) s += expf(r[j] - mx);
    s = block_reduce(s, false, sh);
    const float inv = 1.f / s;
    for (int j = threadIdx.x; j < Vmax; j += 256) r[j] = expf(r[j] - mx) * inv;
}

void launch_ca_softmax(float* sc, int B, int Vmax, hipStream_t st) {
    if (B <= 0 || Vmax <= 0) return;
    hipLaunchKernelGGL(ca_softmax_kernel, dim3(B), dim3(256), 0, st, sc, Vmax);
}

__global__ __launch_bounds__(256) void ca_context_kernel(const float* __restrict__ ev, const float* __restrict__ pr,
                                                         const int* __restrict__ voff, int Vmax, int D,
                                                         float* __restrict__ ctx) {
    const int d = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (d >= D) return;
    const int vb = voff[b + 1] - voff[b];
    const float* e = ev + (size_t)voff[b] * D + d;
    const float* p = pr + (size_t)b * Vmax;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int j = 0;
    for (; j + 4 <= vb; j += 4) {
        a0 += p[j] * e[(size_t)j * D];
        a1 += p[j + 1] * e[(size_t)(j + 1) * D];
        a2 += p[j + 2] * e[(size_t)(j + 2) * D];
        a3 += p[j + 3] * e[(size_t)(j + 3) * D];
    }
    for (; j < vb; ++j) a0 += p[j] * e[(size_t)j * D];
    ctx[(size_t)b * D + d] = (a0 + a1) + (a2 + a3);
}

void launch_ca_context(const float* ev, const float* pr, const int* voff, int B, int Vmax, int D, float* ctx,
                       hipStream_t st) {
    if (B <= 0) return;
    hipLaunchKernelGGL(ca_context_kernel, dim3(cdiv(D, 256), B), dim3(256), 0, st, ev, pr, voff, Vmax, D, ctx);
}

// reward[b][k] = value_head[k] . (attn_o ? RMSNorm_ca(hL + attn_o) : hL)       (rw_model:386,408/427)
__global__ __launch_bounds__(256) void reward_head_kernel(const float* __restrict__ hL, const float* __restrict__ attn_o,
                                                          const float* __restrict__ ca_w, float ca_eps,
                                                          const float* __restrict__ vh, int d, float* __restrict__ out,
                                                          int D) {
    __shared__ float sh[4];
    const int b = blockIdx.x;
    const float* h = hL + (size_t)b * D;
    float rstd = 1.f;
    if (attn_o) {
        const float* o = attn_o + (size_t)b * D;
        float s = 0.f;
        for (int c = threadIdx.x; c < D; c += 256) { const float v = h[c] + o[c]; s += v * v; }
        s = block_reduce(s, false, sh);
        rstd = rsqrtf(s / D + ca_eps);
    }
    for (int k = 0; k < d; ++k) {
        float s = 0.f;
        for (int c = threadIdx.x; c < D; c += 256) {
            float v = h[c];
            if (attn_o) v = ca_w[c] * ((v + attn_o[(size_t)b * D + c]) * rstd);
            s += v * vh[(size_t)k * D + c];
        }
        s = block_reduce(s, false, sh);
        if (threadIdx.x == 0) out[(size_t)b * d + k] = s;
    }
}

void launch_reward_head(const float* hL, const float* attn_o, const float* ca_w, float ca_eps, const float* vh, int d,
                        float* out, int B, int D, hipStream_t st) {
    if (B <= 0) return;
    hipLaunchKernelGGL(reward_head_kernel, dim3(B), dim3(256), 0, st, hL, attn_o, ca_w, ca_eps, vh, d, out, D);
}

// ------------------------------------------------------------------------- mean-pooling reward head
// rw_model_general_preference.py:398-406 (`mean_hidden_state`): the SkipCA block and its RMSNorm are applied to EVERY token
// and the value head reads the mask-weighted mean.  All fp32; fixed summation order (no atomics).
// y[r] = w * x[r] * rsqrt(mean(x[r]^2) + eps), fp32 in and out (hidden_states[-1] for every token)
__global__ __launch_bounds__(256) void rms_rows_f32_kernel(const float* __restrict__ x, const float* __restrict__ w, float eps,
                                                           float* __restrict__ y, int rows, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c] * xr[c];
    const float rstd = rsqrtf(wave_sum(s) / D + eps);
    for (int c = lane; c < D; c += 64) y[(size_t)row * D + c] = w[c] * (xr[c] * rstd);
}

void launch_rms_rows_f32(const float* x, const float* w, float eps, float* y, int rows, int D, hipStream_t st) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(rms_rows_f32_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, w, eps, y, rows, D);
}

// softmax over Vmax columns of which only the first vb hold scores; the other Vmax - vb are the zero-padded image rows of
// the batch (score exactly 0, value rows 0: they only enter the denominator).  In place, one block per row.
__global__ __launch_bounds__(256) void ca_softmax_pad_kernel(float* __restrict__ sc, int ld, int vb, int Vmax) {
    __shared__ float sh[4];
    float* r = sc + (size_t)blockIdx.x * ld;
    const int npad = Vmax - vb;
    float mx = npad > 0 ? 0.f : -INFINITY;
    for (int j = threadIdx.x; j < vb; j += 256) mx = fmaxf(mx, r[j]);
    mx = block_reduce(mx, true, sh);
    float s = 0.f;
    for (int j = threadIdx.x; j < vb; j += 256) s += expf(r[j] - mx);
    s = block_reduce(s, false, sh) + (float)npad * expf(-mx);
    const float inv = 1.f / s;
    for (int j = threadIdx.x; j < vb; j += 256) r[j] = expf(r[j] - mx) * inv;
}

void launch_ca_softmax_pad(float* sc, int rows, int ld, int vb, int Vmax, hipStream_t st) {
    if (rows <= 0 || vb <= 0) return;
    hipLaunchKernelGGL(ca_softmax_pad_kernel, dim3(rows), dim3(256), 0, st, sc, ld, vb, Vmax);
}

// pooled[b] = sum_s mask[b,s] * f(h[b,s]) / max(sum_s mask[b,s], 1e-8),  f(h) = ca_w * rmsnorm(h + o[b,s] + u[b]) when a SkipCA
// term is present (o per token: Phi-3-V; u per sample: Qwen2.5-VL's as-written block), else f(h) = h.  One block per sample.
__global__ __launch_bounds__(256) void ca_pool_kernel(const float* __restrict__ h, const float* __restrict__ o, const float* __restrict__ u,
                                                      const float* __restrict__ ca_w, float ca_eps, const int64_t* __restrict__ mask,
                                                      float* __restrict__ pooled, int S, int D) {
    constexpr int MAXC = 16;                 // D <= 4096
    __shared__ float sh[4];
    const int b = blockIdx.x;
    float acc[MAXC], uu[MAXC], ww[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = threadIdx.x + 256 * i;
        acc[i] = 0.f;
        uu[i] = (u && c < D) ? u[(size_t)b * D + c] : 0.f;
        ww[i] = (ca_w && c < D) ? ca_w[c] : 1.f;
    }
    const bool ca = o != nullptr || u != nullptr;
    float cnt = 0.f;
    for (int s = 0; s < S; ++s) {
        if (mask[(size_t)b * S + s] == 0) continue;          // block-uniform
        cnt += 1.f;
        const size_t base = ((size_t)b * S + s) * D;
        float v[MAXC], sq = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = threadIdx.x + 256 * i;
            v[i] = 0.f;
            if (c < D) {
                v[i] = h[base + c] + uu[i];
                if (o) v[i] += o[base + c];
                sq += v[i] * v[i];
            }
        }
        float rstd = 1.f;
        if (ca) rstd = rsqrtf(block_reduce(sq, false, sh) / D + ca_eps);
#pragma unroll
        for (int i = 0; i < MAXC; ++i) acc[i] += ca ? ww[i] * (v[i] * rstd) : v[i];
    }
    const float inv = 1.f / fmaxf(cnt, 1e-8f);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < D) pooled[(size_t)b * D + c] = acc[i] * inv;
    }
}

void launch_ca_pool(const float* h, const float* o, const float* u, const float* ca_w, float ca_eps, const int64_t* mask,
                    float* pooled, int B, int S, int D, hipStream_t st) {
    if (B <= 0) return;
    if (D > 4096) throw std::runtime_error("ca_pool: D above 4096 is not supported");
    hipLaunchKernelGGL(ca_pool_kernel, dim3(B), dim3(256), 0, st, h, o, u, ca_w, ca_eps, mask, pooled, S, D);
}

// --------------------------------------------------------------------------------------- weights
// Bit-identical to llava_reward_amd.synth.gen_tensor (splitmix64 counter hash, one fp32 multiply).
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_fill_kernel(float* __restrict__ out, size_t n, uint64_t tseed, float scale,
                                                         float offset, int has_offset, int bf16_round) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint64_t h = splitmix64(tseed + i);
        const int c = (int)(h >> 40) - (1 << 23);
        float v = __fmul_rn((float)c, scale);
        if (has_offset) v = __fadd_rn(v, offset);
        if (bf16_round) {
            unsigned u = __builtin_bit_cast(unsigned, v);
            u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
            v = __builtin_bit_cast(float, u);
        }
        out[i] = v;
    }
}

void launch_synth_fill(float* out, size_t n, uint64_t tseed, float scale, float offset, int bf16_round, hipStream_t st) {
    if (!n) return;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(synth_fill_kernel, dim3(grid), dim3(256), 0, st, out, n, tseed, scale, offset, offset != 0.f ? 1 : 0, bf16_round);
}

// Weight profiles of llava_reward_amd.synth (PROFILE_OUTLIER / PROFILE_E4M3), applied in place to the un-rounded values of
// synth_fill_kernel; bit-identical to synth._apply_outlier / synth.round_to_e4m3_np.
struct SynthProfile {
    int outlier, e4m3, bf16_round;
    int gain_vector;               // offset != 0: norm gain vectors take the gain rule, nothing else
    int matrix;                    // rows > 1 && cols > 1 && std > 0
    uint64_t tseed;
    int chan_axis;                 // 0 rows / 1 cols / -1
    int chan[3];
    size_t tiny[4], spike;
    float spike_val;
    int e4m3_exp;
    int cols;
};
__global__ __launch_bounds__(256) void synth_profile_kernel(float* __restrict__ out, size_t n, SynthProfile p) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = out[i];
        if (p.outlier) {
            if (p.gain_vector) {
                const uint64_t h = splitmix64((p.tseed ^ 0xA5A5A5A55A5A5A5Aull) + i);
                if ((h & 63) == 0) v = __fmul_rn(v, (float)(2 + (int)((h >> 8) % 29)));
            } else if (p.matrix) {
                if (p.chan_axis >= 0) {
                    const int c = p.chan_axis == 1 ? (int)(i % (size_t)p.cols) : (int)(i / (size_t)p.cols);
                    if (c == p.chan[0] || c == p.chan[1] || c == p.chan[2]) v = __fmul_rn(v, 200.f);
                }
                if (i == p.tiny[0] || i == p.tiny[1] || i == p.tiny[2] || i == p.tiny[3]) v = __fmul_rn(v, 0x1p-12f);
                if (i == p.spike) v = p.spike_val;
            }
        }
        if (p.e4m3 && p.matrix && !p.gain_vector) {
            float x = fminf(fmaxf(ldexpf(v, -p.e4m3_exp), -448.f), 448.f);
            int ex;
            (void)frexpf(x, &ex);
            const int qe = max(ex - 4, -9);
            v = ldexpf(ldexpf(rintf(ldexpf(x, -qe)), qe), p.e4m3_exp);
        }
        if (p.bf16_round) {
            unsigned u = __builtin_bit_cast(unsigned, v);
            u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
            v = __builtin_bit_cast(float, u);
        }
        out[i] = v;
    }
}
static uint64_t splitmix64_h(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void launch_synth_profile(float* out, int rows, int cols, uint64_t base_seed, uint64_t tseed, const char* name, double std_, double offset,
                          int flags, hipStream_t st) {
    const size_t n = (size_t)rows * cols;
    if (!n) return;
    SynthProfile p{};
    p.outlier = (flags & 2) ? 1 : 0;
    p.e4m3 = (flags & 4) ? 1 : 0;
    p.bf16_round = (flags & 1) ? 0 : 1;
    p.gain_vector = offset != 0.0;
    p.matrix = rows > 1 && cols > 1 && std_ > 0.0;
    p.tseed = tseed;
    p.cols = cols;
    const std::string nm(name);
    auto ends_with = [&](const char* suf) { const size_t l = strlen(suf); return nm.size() >= l && nm.compare(nm.size() - l, l, suf) == 0; };
    p.chan_axis = ends_with("embed_tokens.weight") ? 1 : (ends_with(".mlp.down_proj.weight") && nm.rfind("visual.", 0) != 0) ? 0 : -1;
    if (p.chan_axis >= 0) {
        const uint64_t dim = p.chan_axis == 1 ? (uint64_t)cols : (uint64_t)rows;
        for (int j = 0; j < 3; ++j) p.chan[j] = (int)(splitmix64_h((base_seed ^ 0x5851F42D4C957F2Dull) + (uint64_t)j) % dim);
    }
    for (int k = 0; k < 4; ++k) p.tiny[k] = (size_t)(splitmix64_h((tseed ^ 0x0F1E2D3C4B5A6978ull) + (uint64_t)k) % n);
    const uint64_t hs = splitmix64_h(tseed ^ 0x0123456789ABCDEFull);
    p.spike = (size_t)(hs % n);
    p.spike_val = ((hs >> 63) ? -1.f : 1.f) * (50.f * (float)std_);
    if (p.e4m3) {            // smallest e with 448 * 2^e >= 2^23 * scale (synth.e4m3_tensor_exponent)
        const float scale = (float)(std_ * std::sqrt(12.0) / 16777216.0);
        const double bound = (double)(8388608.f * scale);
        int ex;
        const double m = std::frexp(bound / 448.0, &ex);
        p.e4m3_exp = m == 0.5 ? ex - 1 : ex;
    }
    const int grid = (int)std::min<size_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(synth_profile_kernel, dim3(grid), dim3(256), 0, st, out, n, p);
}

__device__ __forceinline__ void store_elem(void* dst, size_t i, float v, int dt) {
    if (dt == DT_F32) ((float*)dst)[i] = v;
    else if (dt == DT_F16) ((unsigned short*)dst)[i] = Op<F16>::from_f32(v);
    else ((unsigned short*)dst)[i] = Op<BF16>::from_f32(v);
}

__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ src, void* __restrict__ dst, int rows, int cols,
                                                   int ld_dst, int cols_dst, int dst_dtype, int mode, int aux_d, int aux_hd,
                                                   int aux_hdp, void* __restrict__ lo_dst, int* __restrict__ inexact) {
    const size_t total = (size_t)rows * cols_dst;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / cols_dst), c = (int)(i - (size_t)r * cols_dst);
        const float v = c < cols ? src[(size_t)r * cols + c] : 0.f;
        size_t o;
        if (mode == PACK_SWIGLU) {
            const int I = rows >> 1;
            const int g = r < I ? r : r - I;
            const int rr = (g >> 5) * 64 + (r < I ? 0 : 32) + (g & 31);
            o = (size_t)rr * ld_dst + c;
        } else if (mode == PACK_SWIGLU_GATE || mode == PACK_SWIGLU_UP) {
            // separate gate_proj / up_proj tensors written into one interleaved [2I, K] buffer
            o = (size_t)((r >> 5) * 64 + (mode == PACK_SWIGLU_UP ? 32 : 0) + (r & 31)) * ld_dst + c;
        } else if (mode == PACK_ROPE_QKV) {
            // sections q | k | v of aux_d source rows each (heads of aux_hd dims, stored aux_hdp wide: the pad dims of the
            // destination were zeroed at allocation); q and k dims pair-interleaved, v dims in place
            const int sec = min(r / aux_d, 2), w = r - sec * aux_d;
            const int hh = w / aux_hd, d = w - hh * aux_hd, half = aux_hd >> 1;
            const int dd = sec < 2 ? 2 * (d % half) + d / half : d;
            o = (size_t)(sec * (aux_d / aux_hd) * aux_hdp + hh * aux_hdp + dd) * ld_dst + c;
        } else if (mode == PACK_HEADPAD_COLS) {
            // columns are heads of aux_hd dims, stored aux_hdp wide (K operand of the projection after padded-head attention)
            o = (size_t)r * ld_dst + (c / aux_hd) * aux_hdp + (c % aux_hd);
        } else if (mode == PACK_TRANSPOSE) {
            o = (size_t)c * ld_dst + r;
        } else {
            o = (size_t)r * ld_dst + c;
        }
        store_elem(dst, o, v, dst_dtype);
        if (lo_dst) {          // rounding residual of the weight in the operand type (same layout); non-zero anywhere = inexact weight
            float r;
            if (dst_dtype == DT_F16) r = v - Op<F16>::to_f32(Op<F16>::from_f32(v));
            else r = v - Op<BF16>::to_f32(Op<BF16>::from_f32(v));
            store_elem(lo_dst, o, r, dst_dtype);
            // inexact = a weight in the NORMAL range of the operand type with more mantissa bits than it holds.  (bf16 values
            // below 2^-14 land in f16's subnormal range: their residual, <= 3e-8 absolute, is not representable either and is
            // immaterial.)  Benign race: every writer stores 1.
            if (r != 0.f && fabsf(v) >= 6.103515625e-05f && inexact) *inexact = 1;
        }
    }
}

void launch_pack(const float* src, void* dst, int rows, int cols, int ld_dst, int cols_dst, int dst_dtype, int mode,
                 hipStream_t st, int aux_d, int aux_hd, int aux_hdp, void* lo_dst, int* inexact) {
    const size_t total = (size_t)rows * cols_dst;
    if (!total) return;
    if (mode == PACK_SWIGLU && ((rows >> 1) % 32)) throw std::runtime_error("pack: SwiGLU interleave needs I % 32 == 0");
    if (mode == PACK_HEADPAD_COLS && (cols_dst != cols || aux_hd < 1 || aux_hdp < aux_hd)) throw std::runtime_error("pack: bad head padding");
    if (mode == PACK_ROPE_QKV && (aux_hd < 2 || aux_d % aux_hd)) throw std::runtime_error("pack: bad RoPE section geometry");
    if (aux_hdp <= 0) aux_hdp = aux_hd;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(256), 0, st, src, dst, rows, cols, ld_dst, cols_dst, dst_dtype, mode, aux_d, aux_hd, aux_hdp, lo_dst, inexact);
}

__global__ __launch_bounds__(256) void cvt_to_f32_kernel(const void* __restrict__ src, int dt, float* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v;
        if (dt == DT_F32) v = ((const float*)src)[i];
        else if (dt == DT_F16) v = Op<F16>::to_f32(((const unsigned short*)src)[i]);
        else v = Op<BF16>::to_f32(((const unsigned short*)src)[i]);
        dst[i] = v;
    }
}

void launch_cvt_to_f32(const void* src, int src_dtype, float* dst, size_t n, hipStream_t st) {
    if (!n) return;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 16384);
    hipLaunchKernelGGL(cvt_to_f32_kernel, dim3(grid), dim3(256), 0, st, src, src_dtype, dst, n);
}

// ---- W8A8 mode: per-row dynamic quantisation to OCP e4m3 ----
// q[m][k] = e4m3(x[m][k] / s[m]), s[m] = max_k |x[m][k]| / 448 (1 when the row is zero).  One wave per row; the division is the
// correctly rounded fp32 one and the conversion rounds to nearest even, so the oracle reproduces every byte.
template <typename OT>
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const unsigned short* __restrict__ x, int ldx, int K, int rows,
                                                                unsigned char* __restrict__ q, int ldq, float* __restrict__ scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const unsigned short* xr = x + (size_t)row * ldx;
    float amax = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        const uint4 v = *(const uint4*)(xr + k);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            amax = fmaxf(amax, fabsf(Op<OT>::to_f32((unsigned short)(w[i] & 0xFFFF))));
            amax = fmaxf(amax, fabsf(Op<OT>::to_f32((unsigned short)(w[i] >> 16))));
        }
    }
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax / 448.0f : 1.0f;
    if (lane == 0) scale[row] = s;
    unsigned char* qr = q + (size_t)row * ldq;
    for (int k = lane * 8; k < K; k += 512) {
        const uint4 v = *(const uint4*)(xr + k);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        float f[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = Op<OT>::to_f32((unsigned short)(w[i] & 0xFFFF)) / s;
            f[2 * i + 1] = Op<OT>::to_f32((unsigned short)(w[i] >> 16)) / s;
        }
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
        *(uint2*)(qr + k) = make_uint2((unsigned)lo, (unsigned)hi);
    }
}

// ---- split-operand mode, e4m3 residual pass: block-scaled e4m3 (common.h lo8_scale_at) ----
// Rows [hi x K | lo x K] of 2-byte elements: the residual half is rewritten IN PLACE as K e4m3 bytes (the first half of its own
// space) + one E8M0 byte per 128-column block.  For producers that do not write that form themselves.  One wave per row, ONE sweep:
// 16 lanes hold a block, so its maximum is a DPP reduction; a sweep step reads bytes [1024 t, 1024 t + 1024) of the residual half
// and writes [512 t, 512 t + 512): only bytes that were already consumed.  K % 128 == 0.
// aexp2 != null (weights inexact in the operand type): the hi half is ALSO encoded, K more e4m3 bytes with ONE exponent per row, into
// the second half of the residual space (behind the residual bytes), after the residual sweep has consumed it.
template <typename OT>
__global__ __launch_bounds__(256) void quantize_lo_inplace_kernel(unsigned short* __restrict__ a, int ld, int K, int rows, unsigned char* __restrict__ scales,
                                                                  int* __restrict__ aexp2) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    unsigned short* lo = a + (size_t)row * ld + K;
    unsigned char* q = (unsigned char*)lo;
    for (int k0 = 0; k0 < K; k0 += 512) {
        const int k = k0 + lane * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (k < K) v = *(const uint4*)(lo + k);
        __builtin_amdgcn_s_waitcnt(0);                       // the whole wave has its 16 bytes before anyone overwrites them
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        float f[8];
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = Op<OT>::to_f32((unsigned short)(w[i] & 0xFFFF));
            f[2 * i + 1] = Op<OT>::to_f32((unsigned short)(w[i] >> 16));
            m = fmaxf(m, fmaxf(fabsf(f[2 * i]), fabsf(f[2 * i + 1])));
        }
#if LR_EMU_FP6
        {
            const float bm = quad_max(m);
            m = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { f[i] = emu_e2m3(f[i], bm); m = fmaxf(m, fabsf(f[i])); }
        }
#endif
        const int E = e8m0_of_amax(row16_max(m));
        if (k < K) {
            const float sc = e8m0_inv_scale(E);
            int l2 = 0, h2 = 0;
            l2 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * sc, f[1] * sc, l2, false);
            l2 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * sc, f[3] * sc, l2, true);
            h2 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * sc, f[5] * sc, h2, false);
            h2 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * sc, f[7] * sc, h2, true);
            *(uint2*)(q + k) = make_uint2((unsigned)l2, (unsigned)h2);
            if ((lane & 15) == 0) scales[lo8_scale_at(row, k >> 7, rows)] = (unsigned char)E;
        }
    }
    if (aexp2) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned short* hi = a + (size_t)row * ld;
        float hmax = 0.f;
        for (int k = lane * 8; k < K; k += 512) {
            const uint4 v = *(const uint4*)(hi + k);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                hmax = fmaxf(hmax, fabsf(Op<OT>::to_f32((unsigned short)(w[i] & 0xFFFF))));
                hmax = fmaxf(hmax, fabsf(Op<OT>::to_f32((unsigned short)(w[i] >> 16))));
            }
        }
        hmax = wave_max(hmax);
        const int E2 = e8m0_of_amax(hmax);
        if (lane == 0) aexp2[row] = E2;
        for (int k = lane * 8; k < K; k += 512) {
            const uint4 v = *(const uint4*)(hi + k);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
            float f[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f[2 * i] = ldexpf(Op<OT>::to_f32((unsigned short)(w[i] & 0xFFFF)), 127 - E2);
                f[2 * i + 1] = ldexpf(Op<OT>::to_f32((unsigned short)(w[i] >> 16)), 127 - E2);
            }
            int l2 = 0, h2 = 0;
            l2 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], l2, false);
            l2 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], l2, true);
            h2 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], h2, false);
            h2 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], h2, true);
            *(uint2*)(q + K + k) = make_uint2((unsigned)l2, (unsigned)h2);
        }
    }
}

void launch_quantize_lo_inplace(void* a, int ld, int K, int rows, unsigned char* scales, int operand_dtype, hipStream_t st, int* aexp2) {
    if (rows <= 0) return;
    if (K % 128 || ld % 8) throw std::runtime_error("quantize_lo_inplace: K must be a multiple of 128 and the row stride of 8");
    const dim3 grid((rows + 3) / 4), block(256);
    if (operand_dtype == DT_F16) hipLaunchKernelGGL(quantize_lo_inplace_kernel<F16>, grid, block, 0, st, (unsigned short*)a, ld, K, rows, scales, aexp2);
    else hipLaunchKernelGGL(quantize_lo_inplace_kernel<BF16>, grid, block, 0, st, (unsigned short*)a, ld, K, rows, scales, aexp2);
}


// W8 twin of a weight matrix [N, K] (2-byte elements, row stride ldw): e4m3(W * 2^(127 - E)) into the first K bytes of the rows of
// `dst` (same row stride, in 2-byte units), one exponent per TENSOR.  amax_bits: device word, max |w| as float bits.
template <typename OT>
__global__ __launch_bounds__(256) void weight_amax_kernel(const unsigned short* __restrict__ w, int ldw, int K, int N, unsigned* __restrict__ amax_bits) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    float amax = 0.f;
    for (int k = lane; k < K; k += 64) amax = fmaxf(amax, fabsf(Op<OT>::to_f32(w[(size_t)row * ldw + k])));
    amax = wave_max(amax);
    if (lane == 0) atomicMax(amax_bits, __float_as_uint(amax));
}
template <typename OT>
__global__ __launch_bounds__(256) void weight_to_e4m3_kernel(const unsigned short* __restrict__ w, int ldw, int K, int N, int E, unsigned char* __restrict__ dst) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const unsigned short* wr = w + (size_t)row * ldw;
    unsigned char* q = dst + (size_t)row * ldw * 2;
#if LR_EMU_FP6
    for (int k0 = 0; k0 < K; k0 += 256) {          // (uniform trip count: the 8 lanes of a 32-column block reduce together)
        const int k = k0 + lane * 4;
        float f[4] = {0.f, 0.f, 0.f, 0.f};
        float bm = 0.f;
        if (k < K) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { f[i] = Op<OT>::to_f32(wr[k + i]); bm = fmaxf(bm, fabsf(f[i])); }
        }
        bm = fmaxf(bm, __shfl_xor(bm, 1, 64)); bm = fmaxf(bm, __shfl_xor(bm, 2, 64)); bm = fmaxf(bm, __shfl_xor(bm, 4, 64));
        if (k < K) {
            int pk = 0;
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(ldexpf(emu_e2m3(f[0], bm), 127 - E), ldexpf(emu_e2m3(f[1], bm), 127 - E), pk, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(ldexpf(emu_e2m3(f[2], bm), 127 - E), ldexpf(emu_e2m3(f[3], bm), 127 - E), pk, true);
            *(unsigned*)(q + k) = (unsigned)pk;
        }
    }
#else
    for (int k = lane * 4; k < K; k += 256) {
        int pk = 0;
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(ldexpf(Op<OT>::to_f32(wr[k]), 127 - E), ldexpf(Op<OT>::to_f32(wr[k + 1]), 127 - E), pk, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(ldexpf(Op<OT>::to_f32(wr[k + 2]), 127 - E), ldexpf(Op<OT>::to_f32(wr[k + 3]), 127 - E), pk, true);
        *(unsigned*)(q + k) = (unsigned)pk;
    }
#endif
}

// Weights that are not exact in the operand type: `twin` holds their 16-bit residuals (same layout as W).  The residuals become
// e4m3 bytes K .. 2K-1 of their own rows and W's e4m3 twin bytes 0 .. K-1 (via `tmp`, a scratch of W's size, because both overwrite
// their source rows).  Returns the two tensor exponents (W8, Wlo8).  Synchronous, one-time.
void prepare_weight_e4m3_pair(const void* w, void* twin, int ldw, int K, int N, void* tmp, int operand_dtype, unsigned* scratch_word,
                              hipStream_t st, int* wexp, int* wexp2) {
    if (K % 4 || ldw % 2) throw std::runtime_error("prepare_weight_e4m3_pair: K must be a multiple of 4");
    const dim3 grid((N + 3) / 4), block(256);
    auto amax_of = [&](const void* src) {
        LR_HIP_CHECK(hipMemsetAsync(scratch_word, 0, 4, st));
        if (operand_dtype == DT_F16) hipLaunchKernelGGL(weight_amax_kernel<F16>, grid, block, 0, st, (const unsigned short*)src, ldw, K, N, scratch_word);
        else hipLaunchKernelGGL(weight_amax_kernel<BF16>, grid, block, 0, st, (const unsigned short*)src, ldw, K, N, scratch_word);
        unsigned bits = 0;
        LR_HIP_CHECK(hipMemcpyAsync(&bits, scratch_word, 4, hipMemcpyDeviceToHost, st));
        LR_HIP_CHECK(hipStreamSynchronize(st));
        float amax;
        memcpy(&amax, &bits, 4);
        return amax > 0.f ? 127 + (std::ilogb(amax) - 7) : 127;
    };
    *wexp2 = amax_of(twin);
    // e4m3(W_lo) -> bytes 0 .. K-1 of tmp's rows (tmp has W's layout and size)
    if (operand_dtype == DT_F16) hipLaunchKernelGGL(weight_to_e4m3_kernel<F16>, grid, block, 0, st, (const unsigned short*)twin, ldw, K, N, *wexp2, (unsigned char*)tmp);
    else hipLaunchKernelGGL(weight_to_e4m3_kernel<BF16>, grid, block, 0, st, (const unsigned short*)twin, ldw, K, N, *wexp2, (unsigned char*)tmp);
    *wexp = prepare_weight_e4m3(w, ldw, K, N, twin, operand_dtype, scratch_word, st);          // bytes 0 .. K-1 of the twin's rows
    LR_HIP_CHECK(hipMemcpy2DAsync((char*)twin + K, (size_t)ldw * 2, tmp, (size_t)ldw * 2, K, N, hipMemcpyDeviceToDevice, st));
    LR_HIP_CHECK(hipStreamSynchronize(st));
}

// Synchronous (one-time weight preparation): returns the E8M0 exponent of the tensor.
int prepare_weight_e4m3(const void* w, int ldw, int K, int N, void* dst, int operand_dtype, unsigned* scratch_word, hipStream_t st) {
    if (K % 4 || ldw % 2) throw std::runtime_error("prepare_weight_e4m3: K must be a multiple of 4");
    LR_HIP_CHECK(hipMemsetAsync(scratch_word, 0, 4, st));
    const dim3 grid((N + 3) / 4), block(256);
    if (operand_dtype == DT_F16) hipLaunchKernelGGL(weight_amax_kernel<F16>, grid, block, 0, st, (const unsigned short*)w, ldw, K, N, scratch_word);
    else hipLaunchKernelGGL(weight_amax_kernel<BF16>, grid, block, 0, st, (const unsigned short*)w, ldw, K, N, scratch_word);
    unsigned bits = 0;
    LR_HIP_CHECK(hipMemcpyAsync(&bits, scratch_word, 4, hipMemcpyDeviceToHost, st));
    LR_HIP_CHECK(hipStreamSynchronize(st));
    float amax;
    memcpy(&amax, &bits, 4);
    const int E = amax > 0.f ? 127 + (std::ilogb(amax) - 7) : 127;
    if (operand_dtype == DT_F16) hipLaunchKernelGGL(weight_to_e4m3_kernel<F16>, grid, block, 0, st, (const unsigned short*)w, ldw, K, N, E, (unsigned char*)dst);
    else hipLaunchKernelGGL(weight_to_e4m3_kernel<BF16>, grid, block, 0, st, (const unsigned short*)w, ldw, K, N, E, (unsigned char*)dst);
    return E;
}

void launch_quantize_rows_fp8(const void* x, int ldx, int K, int rows, void* q, int ldq, float* scale, int operand_dtype, hipStream_t st) {
    if (rows <= 0) return;
    if (K % 8 || ldx % 8 || ldq % 8) throw std::runtime_error("quantize_rows_fp8: K and the row strides must be multiples of 8");
    const dim3 grid((rows + 3) / 4), block(256);
    if (operand_dtype == DT_F16) hipLaunchKernelGGL(quantize_rows_fp8_kernel<F16>, grid, block, 0, st, (const unsigned short*)x, ldx, K, rows, (unsigned char*)q, ldq, scale);
    else hipLaunchKernelGGL(quantize_rows_fp8_kernel<BF16>, grid, block, 0, st, (const unsigned short*)x, ldx, K, rows, (unsigned char*)q, ldq, scale);
}


// DIAGNOSTIC ONLY (LR_ATT_EMU_LO8=1, tools/dbg/attn_lo8_probe.py): what the rewards would lose if attention's K_lo / V_lo residuals
// were carried as e4m3 with one power-of-two scale per (token, head) -- the operand format of an e4m3 cross-term attention kernel
// that does not exist yet.  Rounds the 16-bit residuals of the given column range to that grid IN PLACE (they stay 16-bit values).
template <typename OT>
__global__ void emulate_lo8_kernel(unsigned short* qkv, size_t rows, int ld, int col0, int heads, int hd) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * (size_t)heads) return;
    unsigned short* p = qkv + (i / heads) * (size_t)ld + col0 + (int)(i % heads) * hd;
    float m = 0.f;
    for (int d = 0; d < hd; ++d) m = fmaxf(m, fabsf(Op<OT>::to_f32(p[d])));
    const int E = e8m0_of_amax(m);
    const float inv = e8m0_inv_scale(E), sc = __builtin_bit_cast(float, (unsigned)E << 23);
    for (int d = 0; d < hd; d += 2) {
        const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(Op<OT>::to_f32(p[d]) * inv, Op<OT>::to_f32(p[d + 1]) * inv, 0, false);
        p[d] = Op<OT>::from_f32(__builtin_amdgcn_cvt_f32_fp8(pk, 0) * sc);
        p[d + 1] = Op<OT>::from_f32(__builtin_amdgcn_cvt_f32_fp8(pk, 1) * sc);
    }
}
void launch_emulate_lo8(void* qkv, size_t rows, int ld, int col0, int heads, int hd, int operand_dtype, hipStream_t st) {
    const size_t n = rows * (size_t)heads;
    if (!n) return;
    if (operand_dtype == DT_F16) hipLaunchKernelGGL(emulate_lo8_kernel<F16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (unsigned short*)qkv, rows, ld, col0, heads, hd);
    else hipLaunchKernelGGL(emulate_lo8_kernel<BF16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (unsigned short*)qkv, rows, ld, col0, heads, hd);
}

}  // namespace lr
