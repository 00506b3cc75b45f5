// HBM-bound row kernels of the Qwen2.5-VL branch (rw_model_general_preference.py:354-371, :387-397; backbone =
// transformers modeling_qwen2_5_vl.py, third party): patch gather in window order, the ViT's 2-D rotary table, the
// 3-D (temporal, height, width) position plan of get_rope_index, the multimodal RoPE table and the as-written SkipCA.
#include "common.h"
#include "kernels.h"

namespace lr {

static inline int cdivq(long a, int b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------------------- patch gather
// Qwen2_5_VisionPatchEmbed is a conv3d whose kernel equals its stride: a linear map over the 1176 values of a patch.
// Row i of the GEMM operand = patch src[i] (window order, Qwen2_5_VisionTransformerPretrainedModel.forward
// `hidden_states[window_index]`; the permutation commutes with the per-row linear map), zero-padded to Kpad.
template <typename OT, typename PT>
__global__ __launch_bounds__(256) void qwen_patch_gather_kernel(const PT* __restrict__ pix, const int* __restrict__ src, int K,
                                                                int Kpad, void* __restrict__ out, int prec) {
    const int row = blockIdx.x;
    const PT* p = pix + (size_t)src[row] * K;
    for (int c = threadIdx.x * 4; c < Kpad; c += 1024) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = 0.f;
            if (c + e < K) {
                if constexpr (sizeof(PT) == 4) x = (float)p[c + e];
                else x = bf16_bits_to_f32(p[c + e]);
            }
            v[e] = x;
        }
        unsigned short* dst = (unsigned short*)out + (size_t)row * (Kpad << prec) + c;
        uint2 h, l;
        split2<OT>(v[0], v[1], h.x, l.x);
        split2<OT>(v[2], v[3], h.y, l.y);
        *(uint2*)dst = h;
        if (prec) *(uint2*)(dst + Kpad) = l;
    }
}

void launch_qwen_patch_gather(const void* pixels, int pix_dtype, const int* src, int rows, int K, int Kpad, void* out,
                              int operand_dtype, hipStream_t st, int prec) {
    if (rows <= 0) return;
    if (Kpad % 4 || Kpad < K) throw std::runtime_error("qwen_patch_gather: bad padding");
    const bool f16 = operand_dtype == DT_F16;
    if (pix_dtype == DT_F32) {
        if (f16) hipLaunchKernelGGL((qwen_patch_gather_kernel<F16, float>), dim3(rows), dim3(256), 0, st, (const float*)pixels, src, K, Kpad, out, prec);
        else hipLaunchKernelGGL((qwen_patch_gather_kernel<BF16, float>), dim3(rows), dim3(256), 0, st, (const float*)pixels, src, K, Kpad, out, prec);
    } else if (pix_dtype == DT_BF16) {
        if (f16) hipLaunchKernelGGL((qwen_patch_gather_kernel<F16, unsigned short>), dim3(rows), dim3(256), 0, st, (const unsigned short*)pixels, src, K, Kpad, out, prec);
        else hipLaunchKernelGGL((qwen_patch_gather_kernel<BF16, unsigned short>), dim3(rows), dim3(256), 0, st, (const unsigned short*)pixels, src, K, Kpad, out, prec);
    } else {
        throw std::runtime_error("qwen_patch_gather: pixel dtype must be f32 or bf16");
    }
}

// ------------------------------------------------------------------------------- ViT rotary table
// Qwen2_5_VisionRotaryEmbedding + apply_rotary_pos_emb_vision: head dim i and i + hd/2 rotate by the same angle;
// angle = h * f[i] for i < hd/4, w * f[i - hd/4] for hd/4 <= i < hd/2, f[j] = theta^(-2j / (hd/2)).
// cs[row][i] = (cos, sin) for i < half_pad; pairs beyond hd/2 (zero pad dims of the stored head) get (1, 0).
__global__ __launch_bounds__(256) void vit_rope_table_kernel(const int2* __restrict__ hw, int rows, const float* __restrict__ inv,
                                                             int quarter, int half_pad, float* __restrict__ cs) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * half_pad) return;
    const int row = (int)(i / half_pad), k = (int)(i - (size_t)row * half_pad);
    float c = 1.f, s = 0.f;
    if (k < 2 * quarter) {
        const int2 p = hw[row];
        const float ang = (float)(k < quarter ? p.x : p.y) * inv[k < quarter ? k : k - quarter];
        c = cosf(ang); s = sinf(ang);
    }
    cs[i * 2] = c;
    cs[i * 2 + 1] = s;
}

void launch_vit_rope_table(const int2* hw, int rows, const float* inv_freq, int quarter, int half_pad, float* cs, hipStream_t st) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(vit_rope_table_kernel, dim3(cdivq((long)rows * half_pad, 256)), dim3(256), 0, st, hw, rows, inv_freq, quarter, half_pad, cs);
}

// ------------------------------------------------------------------------------- token runs
// One wave per row.  A "run" is a maximal stretch of image-slot tokens among the row's un-masked tokens
// (get_rope_index groups the masked-out row by token type); every run consumes one image of image_grid_thw.
// rstat[b] = {number of runs, number of tokens equal to ca_token (masked or not, rw_model:358), 0, 0}.
struct RowScan {
    unsigned long long vb, ib, sb;      // ballots of the current 64-token chunk: valid, valid image slot, run start
    int prev_img;                       // type of the last valid token before the chunk
};

__device__ __forceinline__ unsigned long long lt_mask(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }

__device__ __forceinline__ void scan_chunk(RowScan& r, bool valid, bool img, int lane) {
    r.vb = __ballot(valid);
    r.ib = __ballot(valid && img);
    const unsigned long long before = r.vb & lt_mask(lane);
    bool prev_img = r.prev_img != 0;
    if (before) prev_img = ((r.ib >> (63 - __clzll((long long)before))) & 1ull) != 0;
    r.sb = __ballot(valid && img && !prev_img);
}

__device__ __forceinline__ void scan_advance(RowScan& r) {
    if (r.vb) r.prev_img = (int)((r.ib >> (63 - __clzll((long long)r.vb))) & 1ull);
}

__global__ __launch_bounds__(64) void qwen_runs_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ mask, int S,
                                                       long image_token, long ca_token, int* __restrict__ rstat) {
    const int b = blockIdx.x, lane = threadIdx.x;
    RowScan r{0, 0, 0, 0};
    int nruns = 0, nca = 0;
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        bool valid = false, img = false, ca = false;
        if (s < S) {
            const int64_t id = ids[(size_t)b * S + s];
            valid = mask[(size_t)b * S + s] != 0;
            img = id == image_token;
            ca = id == ca_token;
        }
        scan_chunk(r, valid, img, lane);
        nruns += __popcll(r.sb);
        nca += __popcll(__ballot(ca));
        scan_advance(r);
    }
    if (lane == 0) { rstat[b * 4 + 0] = nruns; rstat[b * 4 + 1] = nca; rstat[b * 4 + 2] = 0; rstat[b * 4 + 3] = 0; }
}

// Qwen2_5_VLModel.get_rope_index for still images.  imgs[k] = {gh, gw, advance = max(h, w) / merge, first slot}
// (merged grid of image k, in slot order); slot2row maps a global image-slot index to its row of the merger output
// (window order).  pos3 = [3][B*S] (temporal, height, width); masked positions keep 0.  img_row[b*S+s] = merger row
// of an image slot, -1 for text.  Runs beyond n_images or longer than their image are clamped (the host wrapper
// checks the totals before the launch, as transformers does: "Image features and image tokens do not match").
__global__ __launch_bounds__(64) void qwen_plan_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ mask, int B, int S,
                                                       long image_token, const int* __restrict__ rstat, const int4* __restrict__ imgs,
                                                       int n_images, const int* __restrict__ slot2row, int n_slots,
                                                       int* __restrict__ pos3, int* __restrict__ img_row) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int img_base = 0;
    for (int i = 0; i < b; ++i) img_base += rstat[i * 4];
    const size_t plane = (size_t)B * S;
    RowScan r{0, 0, 0, 0};
    int text_seen = 0;      // un-masked text tokens before the chunk
    int img_seen = 0;       // un-masked image slots before the chunk
    int runs_seen = 0;      // runs started before the chunk
    int run_base = 0;       // img_seen at the start of the run that is open when the chunk begins
    int run_text = 0;       // text_seen at the start of that run
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        bool valid = false, img = false;
        if (s < S) {
            valid = mask[(size_t)b * S + s] != 0;
            img = ids[(size_t)b * S + s] == image_token;
        }
        scan_chunk(r, valid, img, lane);
        const unsigned long long lt = lt_mask(lane), le = lt | (1ull << lane);
        const unsigned long long tb = r.vb & ~r.ib;
        const int my_text = text_seen + __popcll(tb & lt);
        const int my_img = img_seen + __popcll(r.ib & lt);
        const int starts_le = runs_seen + __popcll(r.sb & le);
        int p0 = 0, p1 = 0, p2 = 0, irow = -1;
        if (valid) {
            // advance of the runs that precede this token's run (image token) or this token (text token)
            const int nprev = img ? starts_le - 1 : starts_le;
            int adv = 0;
            for (int k = 0; k < nprev; ++k) adv += imgs[min(img_base + k, n_images - 1)].z;
            if (!img) {
                p0 = p1 = p2 = my_text + adv;
            } else {
                const unsigned long long mine = r.sb & le;
                int rb = run_base, rt = run_text;
                if (mine) {
                    const int ls = 63 - __clzll((long long)mine);
                    rb = img_seen + __popcll(r.ib & lt_mask(ls));
                    rt = text_seen + __popcll(tb & lt_mask(ls));
                }
                const int4 g = imgs[min(img_base + starts_le - 1, n_images - 1)];
                const int j = min(my_img - rb, g.x * g.y - 1);
                const int start = rt + adv;
                p0 = start; p1 = start + j / g.y; p2 = start + j % g.y;
                irow = slot2row[min(g.w + j, n_slots - 1)];
            }
        }
        if (s < S) {
            const size_t o = (size_t)b * S + s;
            pos3[o] = p0; pos3[plane + o] = p1; pos3[2 * plane + o] = p2;
            img_row[o] = irow;
        }
        // carry: state of the run that is open at the end of the chunk
        if (r.sb) {
            const int ls = 63 - __clzll((long long)r.sb);
            run_base = img_seen + __popcll(r.ib & lt_mask(ls));
            run_text = text_seen + __popcll(tb & lt_mask(ls));
        }
        text_seen += __popcll(tb);
        img_seen += __popcll(r.ib);
        runs_seen += __popcll(r.sb);
        scan_advance(r);
    }
}

void launch_qwen_plan(const int64_t* ids, const int64_t* mask, int B, int S, long image_token, long ca_token, const int4* imgs,
                      int n_images, const int* slot2row, int n_slots, int* rstat, int* pos3, int* img_row, hipStream_t st) {
    if (B <= 0) return;
    hipLaunchKernelGGL(qwen_runs_kernel, dim3(B), dim3(64), 0, st, ids, mask, S, image_token, ca_token, rstat);
    hipLaunchKernelGGL(qwen_plan_kernel, dim3(B), dim3(64), 0, st, ids, mask, B, S, image_token, rstat, imgs, n_images, slot2row,
                       n_slots, pos3, img_row);
}

// ------------------------------------------------------------------------------- multimodal RoPE table
// Qwen2_5_VLRotaryEmbedding + apply_multimodal_rotary_pos_emb: frequency k of the half head takes the temporal
// position for k < s0, the height position for k < s0 + s1, the width position otherwise.  cs[row][k] = (cos, sin).
__global__ __launch_bounds__(256) void mrope_table_kernel(const int* __restrict__ pos3, int rows, const float* __restrict__ inv,
                                                          int s0, int s1, int half, float* __restrict__ cs) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * half) return;
    const int row = (int)(i / half), k = (int)(i - (size_t)row * half);
    const int stream = k < s0 ? 0 : (k < s0 + s1 ? 1 : 2);
    const float ang = (float)pos3[(size_t)stream * rows + row] * inv[k];
    cs[i * 2] = cosf(ang);
    cs[i * 2 + 1] = sinf(ang);
}

void launch_mrope_table(const int* pos3, int rows, const float* inv_freq, int s0, int s1, int half, float* cs, hipStream_t st) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(mrope_table_kernel, dim3(cdivq((long)rows * half, 256)), dim3(256), 0, st, pos3, rows, inv_freq, s0, s1, half, cs);
}

// ------------------------------------------------------------------------------- SkipCA as written
// rw_model_general_preference.py:358-371,387-395: the K/V rows of sample b are hidden_states[0] at the positions
// where input_ids == 151643, i.e. n copies of the embedding row wte[151643]; the masked softmax over n identical
// scores is uniform, so attn_o = W_v wte[151643] for every query when n > 0 and 0 when n == 0 (then every column is
// masked to -1e4 and every V row is zero).  u = W_v wte[ca_token] is computed once at lr_finalize.
__global__ __launch_bounds__(256) void qwen_ca_vec_kernel(const int* __restrict__ rstat, const float* __restrict__ u, int D,
                                                          float* __restrict__ out) {
    const int b = blockIdx.x;
    const bool has = rstat[b * 4 + 1] > 0;
    for (int c = threadIdx.x; c < D; c += 256) out[(size_t)b * D + c] = has ? u[c] : 0.f;
}

void launch_qwen_ca_vec(const int* rstat, const float* u, int B, int D, float* out, hipStream_t st) {
    if (B <= 0) return;
    hipLaunchKernelGGL(qwen_ca_vec_kernel, dim3(B), dim3(256), 0, st, rstat, u, D, out);
}

}  // namespace lr
