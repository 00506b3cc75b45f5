// Host-side launchers of the gfx950 kernels (definitions in gemm.hip / attention.hip / rowops.hip).
#pragma once
#include "common.h"
#include <stdexcept>

namespace lr {

// gemm.hip
void launch_gemm_bt(const GemmParams& p, int operand_dtype, int tile, hipStream_t st);
// true if launch_gemm_bt would run the deep-pipelined kernel (the only one with the fused RoPE epilogue)
bool gemm_bt_is_deep(const GemmParams& p, int tile);
// gemm8.hip (deep-pipelined 256x256 variant, tile == 3)
void launch_gemm_bt8(const GemmParams& p, int operand_dtype, int variant, hipStream_t st);
// W8A8: e4m3 operands (K, lda, ldw in bytes), per-row / per-channel fp32 scales in p.ascale / p.wscale
void launch_gemm_bt8_fp8(GemmParams p, int operand_dtype, hipStream_t st);
void launch_gemm_bt8_mixed(const GemmParams& p, int operand_dtype, hipStream_t st, int dbg = 0);
#ifdef LR_FP6_AB
void launch_gemm_bt8_fp6ab(const GemmParams& p, hipStream_t st, int noepi);      // round 6 A/B builds only (tools/fp6/)
#endif
void launch_emulate_lo8(void* qkv, size_t rows, int ld, int col0, int heads, int hd, int operand_dtype, hipStream_t st);      // diagnostic only
void launch_quantize_lo_inplace(void* a, int ld, int K, int rows, unsigned char* scales, int operand_dtype, hipStream_t st, int* aexp2 = nullptr);
void prepare_weight_e4m3_pair(const void* w, void* twin, int ldw, int K, int N, void* tmp, int operand_dtype, unsigned* scratch_word,
                              hipStream_t st, int* wexp, int* wexp2);
int prepare_weight_e4m3(const void* w, int ldw, int K, int N, void* dst, int operand_dtype, unsigned* scratch_word, hipStream_t st);
void launch_quantize_rows_fp8(const void* x, int ldx, int K, int rows, void* q, int ldq, float* scale, int operand_dtype, hipStream_t st);
// attention.hip
void launch_attention(const AttnParams& p, int batch, int head_dim, bool causal, int operand_dtype, hipStream_t st);

// rowops.hip ------------------------------------------------------------------------------------
// y_op[r][:] = LN(x[r][:]) * w + b   (b == null -> RMSNorm: x * rsqrt(mean x^2 + eps) * w)
// prec = 1 (every operand producer below): split-operand mode, rows are stored [hi | lo], twice as wide (common.h split2)
void launch_norm_rows(const float* x, const float* w, const float* b, void* y, int rows, int H, float eps,
                      int operand_dtype, hipStream_t st, int prec = 0, int group = 1, unsigned char* lo8 = nullptr);   // group g: g consecutive rows form one output row
// lo8: write the residual half as block-scaled e4m3 (common.h lo8_scale_at; lo8 = the scale array) instead of 16-bit residuals
// pixels [B, C, 3, img, img] (fp32/bf16) -> patch matrix [ncrop*g*g, Kpad] operand dtype; crop_src[i] = b*C + c
void launch_im2col(const void* pixels, int pix_dtype, const int* crop_src, int ncrop, int img, int patch, int Kpad,
                   void* out, int operand_dtype, hipStream_t st, int prec = 0);
// x[crop*T + t] = pre_LN( (t ? patch_out[crop*(T-1) + t-1] : cls) + pos[t] )
void launch_clip_embed(const float* patch_out, const float* cls, const float* pos, const float* lnw, const float* lnb,
                       float* x, int ncrop, int T, int H, float eps, hipStream_t st);
// per-sample token plan: positions, image-slot ranks, last valid index, first valid index
// tstat[b] = {last valid index, first valid index, #image slots, #valid tokens}
// image slots: ids < 0 (Phi-3-V) when image_token_id < 0, else ids == image_token_id (LLaVA);
// positions: cumsum(mask)-1 with pads -> 1 (rw_model:344-345) or arange(S) when pos_arange (no position_ids passed)
void launch_token_plan(const int64_t* ids, const int64_t* mask, int B, int S, const int* voff, int* img_row, int* pos,
                       int* tstat, hipStream_t st, long image_token_id = -1, int pos_arange = 0);
// rewards[b][:] = NaN where the image-slot count of row b (tstat) differs from its image-token count voff[b+1] - voff[b]
void launch_slot_check(const int* tstat, const int* voff, float* rewards, int B, int d, hipStream_t st);
// x[b*S+s] = img_row >= 0 ? ev[img_row] : wte[clamp(id)]
void launch_embed(const int64_t* ids, const int* img_row, const unsigned short* wte_bf16, const float* ev, float* x,
                  int rows, int D, int vocab, hipStream_t st);
// cos/sin table [rows][hd/2][2] = (cos, sin) pairs from positions (su-scaled RoPE); long factors iff S > orig_max (the reference
// passes seq_len = the padded length, modeling_phi3_v.py:673)
void launch_rope_table(const int* pos, const int* tstat, int B, int S, const float* inv_freq_short,
                       const float* inv_freq_long, float scaling, int orig_max_pos, int half, float* cs, hipStream_t st,
                       int rotary_seq_len = 0);     // the seq_len the rotary module is called with: S (eager / sdpa, default) or S + 1 (flash)
// qkv32 [rows, 3D] fp32 -> qkv operand dtype with RoPE applied to q and k heads (pair-interleaved head dims)
void launch_rope_split(const float* qkv32, const float* cs, void* out, int rows, int rope_cols, int v_cols, int hd,
                       int operand_dtype, hipStream_t st, int prec = 0);
// HD transform gather (modeling_phi3_v.py:254-362): rows of [sum V, 4H] from CLIP features x [ncrop*T, H]
struct HdSample { int hc, wc, crop0, voff; };
void launch_hd_gather(const float* clipx, const HdSample* samples, int B, int total_rows, int T, int H,
                      const float* sub_gn, const float* glb_gn, void* out, int operand_dtype, hipStream_t st, int prec = 0);

// LLaVA-1.6 (modeling_llava_next.py get_image_features / pack_image_features) -------------------
// patch tokens of every crop (CLS dropped) as GEMM operand rows: out[crop*(T-1) + t] = clipx[crop*T + 1 + t]
void launch_clip_tokens(const float* clipx, void* out, int ncrop, int T, int H, int operand_dtype, hipStream_t st, int prec = 0);
struct LlavaSample { int gh, gw, r0, r1, c0, c1, crop0, voff; };
// ev rows of sample b: [base crop tokens; rows r0..r1 x cols c0..c1 of the hi-res grid, image_newline after each row]
void launch_llava_pack(const float* proj, const LlavaSample* samples, int B, int total_rows, int g, int D,
                       const float* newline, float* ev, hipStream_t st);

// Qwen2.5-VL (rowops_qwen.hip; transformers modeling_qwen2_5_vl.py) -------------------------------
// GEMM operand rows of the patch embedding: out[i] = pixels[src[i]][0..K) zero-padded to Kpad (window order)
void launch_qwen_patch_gather(const void* pixels, int pix_dtype, const int* src, int rows, int K, int Kpad, void* out,
                              int operand_dtype, hipStream_t st, int prec = 0);
// ViT 2-D rotary table cs[row][half_pad][2]: pair k < quarter rotates by h*inv[k], k < 2*quarter by w*inv[k-quarter], else (1,0)
void launch_vit_rope_table(const int2* hw, int rows, const float* inv_freq, int quarter, int half_pad, float* cs, hipStream_t st);
// get_rope_index (still images): rstat[b] = {image runs, tokens == ca_token, 0, 0}; pos3 [3][B*S]; img_row = merger row or -1.
// imgs[k] = {merged grid h, merged grid w, position advance, first slot}; slot2row[n_slots]: slot -> merger output row.
void launch_qwen_plan(const int64_t* ids, const int64_t* mask, int B, int S, long image_token, long ca_token, const int4* imgs,
                      int n_images, const int* slot2row, int n_slots, int* rstat, int* pos3, int* img_row, hipStream_t st);
// multimodal RoPE table cs[row][half][2] from pos3 [3][rows]; frequency k uses stream 0 / 1 / 2 for k < s0 / < s0+s1 / else
void launch_mrope_table(const int* pos3, int rows, const float* inv_freq, int s0, int s1, int half, float* cs, hipStream_t st);
// as-written SkipCA of the qwen branch: out[b] = rstat[b].n_ca > 0 ? u : 0
void launch_qwen_ca_vec(const int* rstat, const float* u, int B, int D, float* out, hipStream_t st);

// tail (fp32) -----------------------------------------------------------------------------------
// y[b] = RMSNorm(x[b*S + (use_last_pos ? S-1 : tstat[b].last_valid)])
void launch_gather_norm_rows(const float* x, const int* tstat, int S, int use_last_pos, const float* w, float eps,
                             float* y, int B, int D, hipStream_t st);
void launch_rowvec_linear(const float* x, const float* W, float* y, int B, int N, int K, hipStream_t st);      // y[b] = W x[b], fp32
void launch_ca_scores(const float* ev, const float* kq, const int* voff, int B, int Vmax, int D, float scale, float* sc,
                      hipStream_t st);
void launch_ca_softmax(float* sc, int B, int Vmax, hipStream_t st);
void launch_ca_context(const float* ev, const float* pr, const int* voff, int B, int Vmax, int D, float* ctx,
                       hipStream_t st);
void launch_reward_head(const float* hL, const float* attn_o, const float* ca_w, float ca_eps, const float* vh, int d,
                        float* out, int B, int D, hipStream_t st);

// mean-pooling reward head (rw_model:398-406), fp32 -------------------------------------------------
// gemm_f32.hip: C = alpha * A op(B), fp32 MFMA (32x32x2); B [N,K] (b_nt) or [K,N]; B elements fp32 or bf16 bits
void launch_gemm_f32(const float* A, const void* B, int b_is_bf16, int b_nt, float* C, int M, int N, int K, int lda, int ldb,
                     int ldc, float alpha, hipStream_t st);
void launch_rms_rows_f32(const float* x, const float* w, float eps, float* y, int rows, int D, hipStream_t st);
// in-place softmax of rows of `ld` floats: vb real scores + (Vmax - vb) implicit zero scores in the denominator
void launch_ca_softmax_pad(float* sc, int rows, int ld, int vb, int Vmax, hipStream_t st);
// pooled[b] = masked mean over tokens of (ca ? ca_w * rmsnorm(h + o + u[b]) : h); o, u, ca_w may be null
void launch_ca_pool(const float* h, const float* o, const float* u, const float* ca_w, float ca_eps, const int64_t* mask,
                    float* pooled, int B, int S, int D, hipStream_t st);

// weights ---------------------------------------------------------------------------------------
void launch_synth_fill(float* out, size_t n, uint64_t tseed, float scale, float offset, int bf16_round, hipStream_t st);
// flags as lr_synth_weights_ex: 1 = no bf16 rounding, 2 = outlier profile, 4 = e4m3-valued matrices; `out` holds the UN-ROUNDED fill
void launch_synth_profile(float* out, int rows, int cols, uint64_t base_seed, uint64_t tseed, const char* name, double std_, double offset,
                          int flags, hipStream_t st);
enum : int { PACK_PLAIN = 0, PACK_SWIGLU = 1, PACK_TRANSPOSE = 2, PACK_ROPE_QKV = 3, PACK_SWIGLU_GATE = 4, PACK_SWIGLU_UP = 5,
             PACK_HEADPAD_COLS = 6 };
// dst[f(r)][c] (ld_dst elements, zero-padded columns up to cols_dst) = convert(src[r][c])
// PACK_ROPE_QKV: rows [0, 2*aux_d) (q and k sections of a fused qkv weight, heads of aux_hd) get their head dims
// pair-interleaved: dim i of the first half and dim i of the second half become neighbours (2i, 2i+1); with
// aux_hdp > aux_hd every head is stored aux_hdp wide (destination pre-zeroed).  PACK_SWIGLU_GATE/UP accept any row
// count (destination rows rounded up to 32 and pre-zeroed).  PACK_HEADPAD_COLS pads the heads along the columns.
void launch_pack(const float* src, void* dst, int rows, int cols, int ld_dst, int cols_dst, int dst_dtype, int mode,
                 hipStream_t st, int aux_d = 0, int aux_hd = 0, int aux_hdp = 0, void* lo_dst = nullptr, int* inexact = nullptr);
// lo_dst (operand-typed destinations only): also store round(w - round(w)) in the same layout and set *inexact if any is non-zero
void launch_cvt_to_f32(const void* src, int src_dtype, float* dst, size_t n, hipStream_t st);

}  // namespace lr
