/* C ABI of the MI355X (gfx950) reward-scoring engine -- libllava_reward_hip.so
 *
 * The reference (sjz5202/LLaVA-Reward) has no FFI for this path: its boundary is four Python
 * callables.  This header is the seam a maintainer binds instead of the torch
 * modules behind them; INTEGRATION.md shows the ctypes stub.  Entry point -> what it replaces:
 *
 *   lr_create / lr_upload_weight / lr_finalize
 *        eval/reward_adaptor_loader.py:32-60   building CustomRewardModel and loading base weights,
 *                                              the (merged) LoRA adapter and the reward heads
 *   lr_forward
 *        llava_reward/models/rw_model_general_preference.py:334-448  CustomRewardModel.custom_forward
 *        (phi3v branch), i.e. modeling_phi3_v.py:1376-1516 Phi3VModel.forward with :221-362
 *        Phi3ImageEmbedding, the CLIP tower (utils/utils.py:264-282), the decoder stack
 *        (:1144-1205), SkipCA (rw_model:376-386) and the value head + EOS gather (:407-448)
 *   lr_forward_qwen
 *        the same function's qwen branch (rw_model:354-371, :387-397) = transformers Qwen2_5_VLForConditionalGeneration
 *        .forward (ViT, merger, mRoPE decoder) + the as-written pad-token SkipCA + value head
 *   lr_all_gather_plan: none (the reference scores on one GPU, eval/batch_inference_rm_phi.py:50-57)
 *
 * Conventions: every function returns 0 on success and a non-zero LR_E* code on failure; nothing
 * throws across the ABI; lr_last_error() returns a message for the last failing call on that
 * handle (or for lr_create when handle is NULL).  No torch types appear here: tensors are plain
 * device pointers owned by the caller.  A handle is bound to one HIP device and is not
 * thread-safe.  lr_forward only enqueues work on `stream` and never synchronises; the caller
 * synchronises before reading `rewards_out`.
 */
#ifndef LLAVA_REWARD_HIP_H
#define LLAVA_REWARD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LR_ABI_VERSION 9

enum { LR_OK = 0, LR_EINVAL = 1, LR_EHIP = 2, LR_ESTATE = 3, LR_ENOTFOUND = 4, LR_ENOMEM = 5 };
enum { LR_DT_BF16 = 0, LR_DT_F16 = 1, LR_DT_F32 = 2 };
/* lr_forward flags */
enum { LR_FWD_TRAINING_LAST_TOKEN = 1,     /* self.training reward selection (rw_model:410-415,429-434) */
       LR_FWD_NO_FINAL_NORM = 2,           /* with lr_set_layer_limits(-1, k), k < layers: hidden_states[k] = the residual stream
                                              entering layer k, which the reference takes when layer_id != 32 (rw_model:349-352) */
       LR_FWD_KEEP_HIDDEN_STATES = 4 };    /* keep every token's residual stream through the LAST decoder layer.  Without it that layer
                                              runs its attention, o_proj and MLP only for the one row per sample the reward is read from
                                              (rw_model:408-421; same arithmetic, bit-identical rewards, ~1/33 of the decoder's work less);
                                              lr_last_hidden_state and the "x" tap need a forward that carried this flag */

#define LR_MAX_HALF_HEAD 64
#define LR_MAX_PINPOINTS 8
#define LR_MAX_FULLATT 8
enum { LR_BACKBONE_PHI3V = 0, LR_BACKBONE_LLAVA_NEXT = 1, LR_BACKBONE_QWEN2_5_VL = 2 };

typedef struct lr_engine* lr_handle;

typedef struct lr_model_desc {
    int32_t struct_size;          /* sizeof(lr_model_desc), ABI guard */
    /* Phi-3 decoder (configuration_phi3_v.py:119-145) */
    int32_t vocab_size, hidden, intermediate, layers, heads;
    float rms_eps;
    int32_t orig_max_pos;         /* original_max_position_embeddings */
    float rope_scaling;           /* sqrt(1 + ln(max_pos/orig)/ln(orig)), modeling_phi3_v.py:468-472 */
    float inv_freq_short[LR_MAX_HALF_HEAD];   /* 1/(short_factor * theta^(2i/hd)), modeling_phi3_v.py:455 */
    float inv_freq_long[LR_MAX_HALF_HEAD];
    /* CLIP tower (modeling_phi3_v.py:68-83; layers = encoder layers actually used) */
    int32_t clip_hidden, clip_heads, clip_mlp, clip_layers, clip_image, clip_patch;
    float clip_ln_eps;
    /* reward head (reward_config.yaml, eval/reward_adaptor_loader.py:25-30) */
    int32_t value_head_dim, add_cross_attention;
    float ca_eps;
    /* capacity: workspace is sized for these at lr_finalize */
    int32_t max_batch, max_seq, max_crops;    /* max_crops counts the global crop */
    int32_t operand_dtype;        /* LR_DT_BF16 or LR_DT_F16: element type of MFMA operands */
    /* backbone (ABI 2).  LR_BACKBONE_PHI3V: fields above as in Phi3VConfig, kv_heads = heads, head_dim = hidden/heads.
     * LR_BACKBONE_LLAVA_NEXT (rw_model_general_preference.py:372-375; transformers LlavaNext + Mistral): GQA decoder
     * with explicit head_dim, image slots marked by image_token_id, positions = arange(S), anyres grid pinpoints
     * (h, w) pairs, no SkipCA.  Weight names are those of the llava-v1.6-*-hf checkpoints. */
    int32_t backbone, kv_heads, head_dim, image_token_id, n_pinpoints;
    int32_t pinpoints[2 * LR_MAX_PINPOINTS];
    /* LR_BACKBONE_QWEN2_5_VL (ABI 3; rw_model_general_preference.py:354-371,387-397; transformers Qwen2_5_VL*):
     * the clip_* fields are ignored.  ViT: `vit_depth` blocks of width vit_hidden (vit_heads heads, SwiGLU MLP of
     * vit_intermediate with biases), patches of vit_in_ch x vit_temporal_patch x vit_patch^2 values, window attention
     * over vit_window pixels except in the blocks listed in vit_fullatt, 2-D rotary (vit_rope_theta), vit_merge^2
     * patches merged per LLM token.  Decoder: GQA with q/k/v bias, multimodal RoPE (inv_freq_short = 1/theta^(2i/hd);
     * frequency i follows the temporal / height / width position for i < s0, < s0+s1, < s0+s1+s2).  SkipCA as written
     * in the reference: K/V rows are the embedding rows of the tokens equal to ca_token_id (151643, rw_model:358).
     * Capacity: max_patches = ViT tokens (rows of pixel_values) per lr_forward_qwen call, at most 4 * max_batch images. */
    int32_t vit_depth, vit_hidden, vit_heads, vit_intermediate, vit_patch, vit_temporal_patch, vit_merge, vit_window, vit_in_ch;
    int32_t vit_n_fullatt, vit_fullatt[LR_MAX_FULLATT];
    float vit_rope_theta, vit_eps;
    int32_t mrope_section[3];
    int32_t ca_token_id, max_patches;
    /* Split-operand ("precise") mode, any backbone: every activation that feeds an MFMA is carried as hi + lo in the
     * operand type (f16: 22 mantissa bits), weights stay single (bf16 checkpoints are exact in f16); every contraction
     * costs 2x (linears) / 3x (attention) the MFMA work.  0 = single-pass operands (default).
     * 2 = the same, with the residual pass of every GEMM that runs on the deep-pipelined kernel made in e4m3 (F16 operands only):
     * the residual half of A is block-scaled e4m3 (one power-of-two scale per row and 128 columns), written in that form by the
     * kernel that produces A where it can (norms, SwiGLU / operand-out GEMM epilogues) and re-encoded in place otherwise; W gets
     * an e4m3 twin on first use; the scaled MFMA accumulates both passes in the same registers; 1.5x instead of 2x, ~8e-6 instead
     * of ~1e-6 per GEMM. */
    int32_t precise;
    /* rw_model_general_preference.py:398-406 `mean_hidden_state`: the SkipCA block + its RMSNorm are applied to every token and
     * the value head reads the attention-mask-weighted mean (fp32 path; rewards are [B, value_head_dim] in train and eval). */
    int32_t mean_hidden_state;
    /* W8A8 mode ("fp8 MFMA weight path", BASELINE configs[4]); needs precise == 0 and F16 operands.  Activations stay f16 in HBM;
     * in front of every GEMM with K % 128 == 0 the rows of A are quantised to OCP e4m3 with one fp32 scale per row (max|x| / 448),
     * each weight gets an e4m3 twin with one scale per output channel on first use, and the GEMM runs on
     * v_mfma_scale_f32_16x16x128_f8f6f4 (1.9x the f16 GEMM rate).  Attention, norms and the fp32 tail are unchanged.  Rewards move
     * by ~1e-2: NOT a parity mode. */
    int32_t w8a8;
    /* ABI 6.  Un-merged LoRA adapter on the decoder linears, rank `lora_rank` (0 = none).  The reference loads its adapter
     * un-merged (eval/reward_adaptor_loader.py:44-45, targets llava_reward/utils/utils.py:194-262: qkv_proj / o_proj /
     * gate_up_proj / down_proj for Phi-3-V, q/k/v/o/gate/up/down_proj for LLaVA and Qwen) and peft evaluates
     * y = W x + (lora_alpha / r) B (A x).  Here every such linear gets two more weight tensors, `<module>.lora_A.weight` [r, in]
     * and `<module>.lora_B.weight` [out, r], the latter uploaded PRE-SCALED by lora_alpha / r; t = x A^T is a small GEMM and
     * t B^T rides in the K loop of the base GEMM (r rounded up to 64 extra columns), so the base weights stay bf16-exact and
     * the layer costs ~6 % more instead of the 33-50 % of merged (inexact) weights.  Not available with w8a8. */
    int32_t lora_rank;
    /* ABI 8.  Which sequence length switches the su-RoPE tables from the short to the long factors (Phi-3-V only).  0 (default): the
     * eager / sdpa attention classes pass seq_len = the padded length S (modeling_phi3_v.py:673, :1081) -> long factors iff
     * S > original_max_position_embeddings.  1: Phi3FlashAttention2 (what every --flash_attn script of the reference runs) passes
     * max(S, position_ids[:, -1].max()) + 1 (:793-794) = S + 1 -> long factors iff S >= original_max_position_embeddings.  The two
     * differ at S == original_max (4096 for Phi-3.5-V) only. */
    int32_t rope_flash_convention;
} lr_model_desc;

int lr_abi_version(void);
int lr_create(const lr_model_desc* desc, int device, lr_handle* out);
int lr_destroy(lr_handle h);
const char* lr_last_error(lr_handle h);

/* Copy one tensor under its reference state_dict name (e.g. "model.layers.0.mlp.down_proj.weight").
 * `data` is host memory unless is_device != 0; dtype is LR_DT_*; shape is checked against the
 * model description.  The engine converts to its packed operand layouts immediately. */
int lr_upload_weight(lr_handle h, const char* name, const void* data, const int64_t* shape, int ndim, int dtype,
                     int is_device);
/* Fill every tensor with the deterministic synthetic weights of llava_reward_amd.synth (same
 * integer hash, generated directly in HBM).  Used by bench.py and the full-size parity test. */
int lr_synth_weights(lr_handle h, uint64_t seed);
/* The same with flags: 1 = do NOT round the synthetic values to bf16, i.e. fp32-valued weights that are inexact in the operand
 * type, as the merged LoRA weights of a real LLaVA-Reward checkpoint are (bench.py --merged-weights);
 * 2 = outlier profile (llava_reward_amd.synth.PROFILE_OUTLIER: three massive residual-stream channels -- embedding columns and
 * decoder down_proj rows x 200 --, norm gains of 2..30 on ~1.6 % of the channels, one 50-sigma element and four elements below
 * f16's normal range per matrix: what trained checkpoints show and N(0, 0.02) init does not);
 * 4 = every matrix rounded to the OCP e4m3 grid under one power-of-two scale per tensor (the de-quantised weights of an
 * fp8-weight checkpoint, BASELINE configs[4]). */
int lr_synth_weights_ex(lr_handle h, uint64_t seed, int flags);
/* Number of expected weight tensors / name of the i-th one (for loaders and tests). */
int lr_num_weights(lr_handle h);
const char* lr_weight_name(lr_handle h, int i);
/* Checks that every tensor was provided and allocates the activation workspace. */
int lr_finalize(lr_handle h);
size_t lr_workspace_bytes(lr_handle h);

/* One scoring pass.  input_ids/attention_mask: device int64 [B,S] (image slots are negative ids);
 * pixel_values: device [B, n_crops, 3, img, img] of pix_dtype (F32 or BF16); image_sizes: HOST int64
 * [B,2] = HD-transformed (h, w) for Phi-3-V, ORIGINAL (h, w) for LLaVA; rewards_out: device fp32 [B, value_head_dim].
 * Every row must hold exactly as many image slots as its image_sizes produce image tokens (the reference fails such a batch,
 * modeling_phi3_v.py:247).  The call cannot know without synchronising, so it never reads out of bounds (surplus slots keep
 * their text embedding) and returns NaN rewards for a row whose counts differ; callers that want the reference's exception
 * compare the counts on the host first, as the Python wrapper does. */
int lr_forward(lr_handle h, const int64_t* input_ids, const int64_t* attention_mask, const void* pixel_values,
               int pix_dtype, const int64_t* image_sizes_host, int B, int S, int n_crops, int flags, float* rewards_out,
               void* hip_stream);

/* One scoring pass of the Qwen2.5-VL branch: the reference's `inputs_batch` (rw_model:354-357).  input_ids /
 * attention_mask: device int64 [B,S], image slots = image_token_id, already expanded by the processor to
 * t*h*w/merge^2 tokens per image; pixel_values: device [sum t*h*w, vit_in_ch*vit_temporal_patch*vit_patch^2] of
 * pix_dtype, patches in the processor's merge-block order; image_grid_thw: HOST int64 [n_images,3] (t must be 1),
 * images in the order their slots appear in input_ids (row-major).  rewards_out: device fp32 [B, value_head_dim]. */
int lr_forward_qwen(lr_handle h, const int64_t* input_ids, const int64_t* attention_mask, const void* pixel_values,
                    int pix_dtype, const int64_t* image_grid_thw_host, int n_images, int B, int S, int flags,
                    float* rewards_out, void* hip_stream);

/* `outputs["last_hidden_state"]` of the last forward on this handle (rw_model_general_preference.py:347-353, returned by
 * custom_forward(return_output=True) and read by the trainer's evaluate, rm_trainer_general_preference.py:414-418): the final
 * RMSNorm of every token's residual stream (or the stream itself with no_final_norm, i.e. hidden_states[layer_id]), fp32
 * [B*S, hidden] written to DEVICE memory `out_dev` on `stream`.  Off the scoring path: the forward itself only norms the row
 * it gathers.  Valid until the next forward on the handle. */
int lr_last_hidden_state(lr_handle h, float* out_dev, size_t capacity, int no_final_norm, void* hip_stream);
/* The zero-padded projected image tokens of the last lr_forward, [B, V_max, hidden] fp32 on the device: the tensor the reference's
 * backbone appends as the LAST entry of `hidden_states` (modeling_phi3_v.py:242-245 img_token_batch_embedding, :1505; read back as the
 * SkipCA key/value source at rw_model_general_preference.py:353).  *vmax receives V_max of that forward; out_dev == NULL only reports
 * it (size the buffer, then call again).  Stream-ordered; Phi-3-V / LLaVA handles (ABI 9). */
int lr_vision_embeds(lr_handle h, float* out_dev, size_t capacity, int* vmax, void* hip_stream);
/* Debug taps: copy an internal fp32 buffer of the last forward to host (synchronises).  Names:
 * "clip_x" [crops*T, Hc], "ev" [sumV, D], "x" [B*S, D] (residual stream after the last layer),
 * "hL" [B, D]; Qwen: "vit_x" [patches, vit_hidden] (window order), "ev" [patches/merge^2, D] (window order),
 * "pos3" [3, B*S] (as floats).  Returns the number of floats copied through *n. */
int lr_read_tap(lr_handle h, const char* name, float* host_out, size_t capacity, size_t* n);
/* Stop after `n_clip_layers` / `n_layers` (-1 = all); for stage-wise parity tests. */
int lr_set_layer_limits(lr_handle h, int n_clip_layers, int n_layers);
/* Operand form per stage (default: lr_model_desc.precise everywhere).  Forms: -1 = the descriptor's, 0 = single-pass operands,
 * 1 = split operands with 16-bit residual passes (the strict form), 2 = split operands with e4m3 residual passes; a stage can only
 * take a split form the handle was created with.  clip_form: the vision tower (CLIP / the Qwen ViT, up to the projector / merger).
 * Decoder layers [decoder_first, layers - decoder_last) take decoder_mid_form, the first / last ones the descriptor's.  Two uses:
 * (1) a handle created with precise == 2 can run the strict form (1, 1, 0, 0) without being rebuilt -- what the Python layer's
 * operand-form probe (RewardModel.to) locks in when, on the loaded weights, the default form's rewards sit further from the strict
 * form's than its parity budget; (2) measurements (tools/prec_map_probe.py).  The map survives weight uploads; callers that
 * derive it from the weights re-derive it when lr_weights_epoch has moved. */
int lr_set_precision_map(lr_handle h, int clip_form, int decoder_mid_form, int decoder_first, int decoder_last);
/* Operand form per SITE of the decoder layers the map above covers (ABI 9): -1 = the layer's form (decoder_mid_form), 1 = 16-bit
 * residual passes, 2 = e4m3 residual passes.  Sites: qkv = input norm + qkv projection; attention (both forms are three-pass split
 * operands; 1 = the exact softmax maximum, 2 = the lazy one); o_proj; gate_up = post-attention norm + gate_up projection; down.  A
 * site is an operand's producer together with the launch that reads it, so the forms of neighbouring sites are independent
 * (modeling_phi3_v.py:1144-1205 runs them all in one dtype; which of them a weight set that amplifies operand rounding needs strict
 * was measured with tools/prec_map_probe.py sites: DESIGN.md 4c).  Layers outside the map's range keep the descriptor's form. */
int lr_set_precision_sites(lr_handle h, int qkv_form, int attention_form, int o_proj_form, int gate_up_form, int down_form);
/* Thresholds of the attention kernels' lazy softmax reference maximum (ABI 9; log2 units, 0 .. 15): a row's reference moves only when
 * its new maximum exceeds it by more than the threshold, which spares most rescales of the output accumulators (-14 % per launch) at the
 * price of fp32-level noise on the softmax weights (exponent arguments rounded at ulp(threshold) instead of ulp(0)) -- invisible on
 * benign weights, up to 3e-4 on a weight set that amplifies rounding (profiles/r6_outlier_fp64.log).  0 = the exact running maximum of
 * the reference's softmax (modeling_phi3_v.py:685-701).  default_stages (8 at lr_create): launches of stages in the e4m3-residual
 * default form; strict_stages (0 at lr_create): launches of stages in the strict form -- the yardstick keeps the reference's own
 * arithmetic; the Python layer raises it to 8 for ONE pass of its operand-form probe, as a numerically equivalent re-statement of the
 * strict form whose distance to the exact one is the weight set's own fp32 noise floor (model.py _compare_forms). */
int lr_set_attention_lazy_threshold(lr_handle h, float default_stages, float strict_stages);
/* Counts the calls that changed this handle's weights (lr_upload_weight, lr_synth_weights*): anything derived from the weights --
 * the operand form the Python layer locks at .to('cuda') -- is stale once it has moved. */
uint64_t lr_weights_epoch(lr_handle h);
/* GEMM tile selection: -1 heuristic, 0 = 128x128, 1 = 256x128, 2 = 256x256. */
int lr_set_gemm_tile(lr_handle h, int tile);

/* ---- single-kernel entry points (per-kernel parity tests and microbenchmarks) ---- */
int lr_op_gemm_bt(const void* A, const void* W, void* C, const float* bias, int M, int N, int K, int lda, int ldw,
                  int ldc, int epi, int act, int operand_dtype, int tile, void* hip_stream);
/* Split-operand form (lr_model_desc.precise): A is [M, 2K] = [A_hi | A_lo] in the operand type, W stays [N, K]; operand-typed
 * outputs (EPI_OUT_OP, EPI_SWIGLU_OP) come back as [C_hi | C_lo], twice as wide; fp32 outputs are unchanged. */
int lr_op_gemm_bt_split(const void* A, const void* W, void* C, const float* bias, int M, int N, int K, int epi, int act,
                        int operand_dtype, int tile, void* hip_stream);
/* K-extension (an un-merged LoRA adapter inside the base GEMM, lr_model_desc.lora_rank): C = epi(A W^T + T B^T), W [N, K], B [N, k2]
 * (Blo: its rounding residuals when B is not exact in the operand type, else NULL), k2 % 64 == 0.  split = 0: A [M, K], T [M, k2];
 * split = 1: A [M, 2K] = [hi | lo], T [M, 2 k2] = [hi | lo], operand-typed outputs [C_hi | C_lo].  Deep-pipelined kernel only. */
int lr_op_gemm_bt_ext(const void* A, const void* W, const void* T, const void* B, const void* Blo, void* C, const float* bias, int M,
                      int N, int K, int k2, int split, int epi, int act, int operand_dtype, void* hip_stream);
/* QKV projection with the fused RoPE epilogue: C_op[m][n] = rotate(A W^T) for n < rope_cols, pairs (2i, 2i+1) of each
 * rope_hd-wide head rotated by cs[m][i] = (cos, sin); weight rows must already be pair-interleaved. */
int lr_op_gemm_rope(const void* A, const void* W, void* C, const float* bias, const float* cs, int M, int N, int K,
                    int rope_cols, int rope_hd, int operand_dtype, int tile, void* hip_stream);   /* bias: [N] packed like W's rows, or NULL */
int lr_op_attention(const void* Q, const void* K, const void* V, void* O, const int64_t* mask, const int* kmin, int ldq,
                    int ldo, int qoff, int koff, int voff, int batch, int S, int heads, int head_dim, int causal,
                    int kv_group, float scale, int operand_dtype, void* hip_stream);   /* kv_group = query heads per K/V head */
/* Split-operand form: Q/K/V rows carry their residuals lo_off columns to the right (ldq counts both halves); both contractions
 * are evaluated as hi.hi + hi.lo + lo.hi; O is stored [O_hi | O_lo] when o_split > 0 (ldo counts both halves). */
int lr_op_attention_split(const void* Q, const void* K, const void* V, void* O, const int64_t* mask, const int* kmin, int ldq,
                          int ldo, int qoff, int koff, int voff, int lo_off, int o_split, int batch, int S, int heads,
                          int head_dim, int causal, int kv_group, float scale, int operand_dtype, void* hip_stream);
/* ... with the softmax reference-maximum threshold spelled out (ABI 9): a row's running reference moves only when its new maximum
 * exceeds it by more than lazy_threshold (log2 units, 0 .. 15).  0 = the exact running maximum of the reference's softmax
 * (modeling_phi3_v.py:685-701): what the engine runs in strict-form stages; 8 = what it runs in default-form stages and what
 * lr_op_attention_split runs. */
int lr_op_attention_split_ex(const void* Q, const void* K, const void* V, void* O, const int64_t* mask, const int* kmin, int ldq,
                             int ldo, int qoff, int koff, int voff, int lo_off, int o_split, int batch, int S, int heads,
                             int head_dim, int causal, int kv_group, float scale, float lazy_threshold, int operand_dtype,
                             void* hip_stream);
/* Block-diagonal (ragged) dense attention: rows [cu[i], cu[i+1]) attend to each other only (the ViT's windows /
 * images, transformers Qwen2_5_VLVisionAttention over cu_seqlens).  cu_seqlens: HOST int32 [n_seg + 1]. */
int lr_op_attention_segments(const void* Q, const void* K, const void* V, void* O, const int32_t* cu_seqlens_host, int n_seg,
                             int ldq, int ldo, int qoff, int koff, int voff, int heads, int head_dim, float scale,
                             int operand_dtype, void* hip_stream);
/* W8A8 mode: rows of an operand-typed matrix -> OCP e4m3 bytes q [rows, K] + one fp32 scale per row (max|x| / 448), and
 * C = epilogue((A8 W8^T) * ascale[m] * wscale[n]) on v_mfma_scale_f32_16x16x128_f8f6f4; K multiple of 128. */
int lr_op_quantize_rows_fp8(const void* x, int rows, int K, int ldx, void* q, float* scale, int operand_dtype, void* hip_stream);
int lr_op_gemm_fp8(const void* A8, const float* ascale, const void* W8, const float* wscale, void* C, const float* bias, int M, int N,
                   int K, int ldc, int epi, int act, int operand_dtype, void* hip_stream);
/* Split-operand GEMM with the e4m3 residual pass (lr_model_desc.precise == 2): A = [A_hi | A_lo] (2-byte elements, 2K per row),
 * W [N, K], W8 = DEVICE scratch of the size of W (the e4m3 twin), scratch = DEVICE bytes, lr_op_lo8_scratch_bytes(M, K) of them
 * (block scales of A's residuals: one E8M0 byte per row and 128 columns, one KB per 4 K-tiles and 256-row tile in the consuming
 * kernel's lane order, csrc/common.h lo8_scale_at; then one int32 per row for flag 8), wexp = HOST int (in/out).
 * flags: 1 = prepare W8 from W and store its exponent in *wexp (synchronous), 2 = re-encode the residual half of A in place
 * (e4m3 bytes + the block scales), 4 = stop there (no GEMM), 8 = W is NOT exact in the operand type: on entry W8 holds
 * its 16-bit residuals (W's layout); a third segment A_hi(e4m3) x e4m3(W_lo)^T is added (one exponent per row; wexp: int [2]),
 * 16 = diagnostic, 32 = operand-typed output (EPI_OUT_OP / EPI_SWIGLU_OP, columns % 128 == 0) with ITS residual half in the one-byte
 * form as well, written by the epilogue: e4m3 bytes in C + the block scales of C at scratch + lr_op_lo8_scratch_bytes(M, K)
 * (lr_op_lo8_scratch_bytes(M, columns) more bytes) -- the next lr_op_gemm_bt_mixed takes C as its A with that pointer as its scratch
 * and without flag 2.
 * Output as lr_op_gemm_bt_split.  K multiple of 128. */
size_t lr_op_lo8_scratch_bytes(int M, int K);
int lr_op_gemm_bt_mixed(void* A, const void* W, void* W8, void* scratch, void* C, const float* bias, int M, int N, int K, int epi, int act,
                        int operand_dtype, int flags, int* wexp, void* hip_stream);
int lr_op_norm_rows(const float* x, const float* w, const float* b, void* y, int rows, int H, float eps,
                    int operand_dtype, void* hip_stream);
int lr_op_synth_fill(float* out, size_t n, uint64_t seed, const char* name, float std, float offset, int bf16_round,
                     void* hip_stream);

/* ---- input hand-over: the Phi-3.5-V image processor on the GPU ----
 * Replaces Phi3VImageProcessor.preprocess (llava_reward/models/base_mllm/phi3_v/processing_phi3_v.py:262-288 with
 * HD_transform :85-107 and padding_336 :62-72) for ONE image: `rgb` = DEVICE uint8 [height, width, 3] (what
 * PIL's Image.convert('RGB') holds), `pixel_values` = DEVICE fp32 [num_crops + 1, 3, 336, 336] (global view first, then the
 * local crops row-major, zero crops behind them), `image_size` = HOST int64 [2] = padded (h, w) (may be NULL),
 * `num_img_tokens` = HOST int32 (:269; may be NULL).  Local crops are bit-exact with the reference (Pillow's 8-bit
 * resampler); the bicubic global view is fp32 (<= 1e-5 of torch's).  `workspace` = DEVICE scratch of at least
 * lr_hd_transform_workspace(height, width, num_crops) bytes (0 = invalid arguments).  Enqueues on `hip_stream` and returns. */
size_t lr_hd_transform_workspace(int height, int width, int num_crops);
int lr_hd_transform(const uint8_t* rgb, int height, int width, int num_crops, float* pixel_values, int64_t* image_size,
                    int32_t* num_img_tokens, void* workspace, size_t workspace_bytes, void* hip_stream);

/* The Qwen2-VL image processor (transformers image_processing_qwen2_vl, the PIL / "slow" arithmetic of the pinned 4.50;
 * call site: the processor built in llava_reward/utils/utils.py:34-44 with min_pixels = 256*28*28, max_pixels = 1280*28*28)
 * for ONE image: smart_resize -> Pillow BICUBIC on uint8 -> * 1/255, (x - mean) / std -> patchify.
 * `pixel_values` = DEVICE fp32 [grid_h * grid_w, 1176] (rows in 2x2 merge-block order, the single frame repeated for the two
 * temporal slots), `grid_thw` = HOST int64 [3] = (1, grid_h, grid_w).  lr_qwen_image_grid is host-only: call it first to size
 * `pixel_values`.  Results are bit-exact with the processor. */
int lr_qwen_image_grid(int height, int width, int64_t min_pixels, int64_t max_pixels, int64_t* grid_thw);
size_t lr_qwen_image_workspace(int height, int width, int64_t min_pixels, int64_t max_pixels);
int lr_qwen_image_transform(const uint8_t* rgb, int height, int width, int64_t min_pixels, int64_t max_pixels, float* pixel_values,
                            int64_t* grid_thw, void* workspace, size_t workspace_bytes, void* hip_stream);

/* The LLaVA-NeXT image processor (transformers image_processing_llava_next, PIL arithmetic; call site: the processor built in
 * llava_reward/utils/utils.py:46-55, used by eval/batch_inference_rm_llava.py) for ONE image: select_best_resolution over
 * `pinpoints` (HOST int32 [n, 2] = (h, w), multiples of 336), crop 0 = the image resized to 336x336, crops 1.. = the
 * aspect-preserving BICUBIC resize centred on a zero canvas and cut into 336x336 tiles, * 1/255, (x - mean) / std; crops up to
 * `max_crops` zero-filled (the processor's _pad_for_batching).  `pixel_values` = DEVICE fp32 [max_crops, 3, 336, 336],
 * `image_size` = HOST int64 [2] = the ORIGINAL (h, w).  lr_llava_image_geometry (host-only) returns
 * (best_h, best_w, resized_h, resized_w, n_crops).  Results are bit-exact with the processor. */
int lr_llava_image_geometry(int height, int width, const int32_t* pinpoints, int n_pinpoints, int32_t* out5);
size_t lr_llava_image_workspace(int height, int width, const int32_t* pinpoints, int n_pinpoints);
int lr_llava_image_transform(const uint8_t* rgb, int height, int width, const int32_t* pinpoints, int n_pinpoints, int max_crops,
                             float* pixel_values, int64_t* image_size, void* workspace, size_t workspace_bytes, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* LLAVA_REWARD_HIP_H */
