// C-ABI engine: weight store, workspace and the launch sequence of one scoring pass.
// See include/llava_reward_hip.h for the contract and the reference lines each entry replaces.
#include "engine.h"


namespace {

// Phi-3.5-V names: modeling_phi3_v.py:1332-1374, :118-207; reward heads rw_model_general_preference.py:314-326
void build_weight_table_phi(lr_engine* e) {
    const lr_model_desc& d = e->d;
    const int Hc = d.clip_hidden, D = d.hidden, I = d.intermediate;
    const int od = e->op_dt;
    const std::string ep = "model.vision_embed_tokens.";
    e->wte = (unsigned short*)e->dalloc((size_t)d.vocab_size * D * 2, true);      // bf16 embedding table (not an MFMA operand)
    add_slot(e, "model.embed_tokens.weight", {d.vocab_size, D}, e->wte, D, D, DT_BF16, PACK_PLAIN, 0.02, 0);
    register_clip(e, "model.vision_embed_tokens.img_processor.vision_model.");
    e->glb_gn = falloc(e, 4 * Hc); e->sub_gn = falloc(e, 4 * Hc);
    vec_slot(e, ep + "glb_GN", {1, 1, 4 * Hc}, e->glb_gn, 0.02, 0);
    vec_slot(e, ep + "sub_GN", {1, 1, 1, 4 * Hc}, e->sub_gn, 0.02, 0);
    e->p0_w = oalloc(e, (size_t)D * 4 * Hc); e->p0_b = falloc(e, D);
    e->p2_w = oalloc(e, (size_t)D * D); e->p2_b = falloc(e, D);
    add_slot(e, ep + "img_projection.0.weight", {D, 4 * Hc}, e->p0_w, 4 * Hc, 4 * Hc, od, PACK_PLAIN, 0.02, 0);
    vec_slot(e, ep + "img_projection.0.bias", {D}, e->p0_b, 0.02, 0);
    add_slot(e, ep + "img_projection.2.weight", {D, D}, e->p2_w, D, D, od, PACK_PLAIN, 0.02, 0);
    vec_slot(e, ep + "img_projection.2.bias", {D}, e->p2_b, 0.02, 0);
    e->dl.resize(d.layers);
    for (int l = 0; l < d.layers; ++l) {
        DecLayer& L = e->dl[l];
        const std::string p = "model.layers." + std::to_string(l) + ".";
        L.ln1 = falloc(e, D); L.ln2 = falloc(e, D);
        L.qkv_w = oalloc(e, (size_t)3 * D * D); L.o_w = oalloc(e, (size_t)D * D);
        L.gu_w = oalloc(e, (size_t)2 * I * D); L.down_w = oalloc(e, (size_t)D * I);
        vec_slot(e, p + "input_layernorm.weight", {D}, L.ln1, 0.05, 1.0);
        add_slot(e, p + "self_attn.qkv_proj.weight", {3 * D, D}, L.qkv_w, D, D, od, PACK_ROPE_QKV, 0.02, 0);
        e->slots.back().aux_d = D; e->slots.back().aux_hd = e->hd;
        add_slot(e, p + "self_attn.o_proj.weight", {D, D}, L.o_w, D, D, od, PACK_PLAIN, 0.02, 0);
        vec_slot(e, p + "post_attention_layernorm.weight", {D}, L.ln2, 0.05, 1.0);
        add_slot(e, p + "mlp.gate_up_proj.weight", {2 * I, D}, L.gu_w, D, D, od, PACK_SWIGLU, 0.02, 0);
        add_slot(e, p + "mlp.down_proj.weight", {D, I}, L.down_w, I, I, od, PACK_PLAIN, 0.02, 0);
        if (d.lora_rank > 0) {      // utils/utils.py:194-222 create_lora_config: qkv_proj, o_proj, gate_up_proj, down_proj
            register_lora(e, L.lqkv, p + "self_attn.qkv_proj", 3 * D, D, 0, 1, 3 * D, 0, PACK_ROPE_QKV, D, e->hd);
            register_lora(e, L.lo, p + "self_attn.o_proj", D, D, 0, 1, D, 0, PACK_PLAIN);
            register_lora(e, L.lgu, p + "mlp.gate_up_proj", 2 * I, D, 0, 1, 2 * I, 0, PACK_SWIGLU);
            register_lora(e, L.ldown, p + "mlp.down_proj", D, I, 0, 1, D, 0, PACK_PLAIN);
        }
    }
    e->norm_w = falloc(e, D);
    vec_slot(e, "model.norm.weight", {D}, e->norm_w, 0.05, 1.0);
    if (d.add_cross_attention) {
        e->Wq = falloc(e, (size_t)D * D);
        e->WkT = falloc(e, (size_t)D * D);
        e->Wv = falloc(e, (size_t)D * D);
        e->ca_w = falloc(e, D);
        add_slot(e, "W_q.weight", {D, D}, e->Wq, D, D, DT_F32, PACK_PLAIN, 0.02, 0);
        add_slot(e, "W_k.weight", {D, D}, e->WkT, D, D, DT_F32, PACK_TRANSPOSE, 0.02, 0);
        add_slot(e, "W_v.weight", {D, D}, e->Wv, D, D, DT_F32, PACK_PLAIN, 0.02, 0);
        vec_slot(e, "ca_layernorm.weight", {D}, e->ca_w, 0.05, 1.0);
    }
    e->vh = falloc(e, (size_t)d.value_head_dim * D);
    add_slot(e, "value_head.weight", {d.value_head_dim, D}, e->vh, D, D, DT_F32, PACK_PLAIN, 1.0 / std::sqrt((double)D), 0);
    upload_rope_tables(e);
}

// llava-v1.6-mistral-7b-hf checkpoint names (transformers LlavaNextForConditionalGeneration, 4.50 layout)
void build_weight_table_llava(lr_engine* e) {
    const lr_model_desc& d = e->d;
    const int Hc = d.clip_hidden, D = d.hidden, I = d.intermediate, hd = e->hd, Hq = e->Hq, Hkv = e->Hkv;
    const int od = e->op_dt;
    e->wte = (unsigned short*)e->dalloc((size_t)d.vocab_size * D * 2, true);      // bf16 embedding table (not an MFMA operand)
    add_slot(e, "language_model.model.embed_tokens.weight", {d.vocab_size, D}, e->wte, D, D, DT_BF16, PACK_PLAIN, 0.02, 0);
    register_clip(e, "vision_tower.vision_model.");
    e->p0_w = oalloc(e, (size_t)D * Hc); e->p0_b = falloc(e, D);
    e->p2_w = oalloc(e, (size_t)D * D); e->p2_b = falloc(e, D);
    add_slot(e, "multi_modal_projector.linear_1.weight", {D, Hc}, e->p0_w, Hc, Hc, od, PACK_PLAIN, 0.02, 0);
    vec_slot(e, "multi_modal_projector.linear_1.bias", {D}, e->p0_b, 0.02, 0);
    add_slot(e, "multi_modal_projector.linear_2.weight", {D, D}, e->p2_w, D, D, od, PACK_PLAIN, 0.02, 0);
    vec_slot(e, "multi_modal_projector.linear_2.bias", {D}, e->p2_b, 0.02, 0);
    e->newline = falloc(e, D);
    vec_slot(e, "image_newline", {D}, e->newline, 0.02, 0);
    e->dl.resize(d.layers);
    for (int l = 0; l < d.layers; ++l) {
        DecLayer& L = e->dl[l];
        const std::string p = "language_model.model.layers." + std::to_string(l) + ".";
        L.ln1 = falloc(e, D); L.ln2 = falloc(e, D);
        L.qkv_w = oalloc(e, (size_t)e->Nqkv * D); L.o_w = oalloc(e, (size_t)D * Hq);
        L.gu_w = oalloc(e, (size_t)2 * I * D); L.down_w = oalloc(e, (size_t)D * I);
        vec_slot(e, p + "input_layernorm.weight", {D}, L.ln1, 0.05, 1.0);
        // q, k, v land in one fused [Hq + 2 Hkv, D] matrix; q and k rows pair-interleaved per head for the RoPE epilogue
        add_slot(e, p + "self_attn.q_proj.weight", {Hq, D}, L.qkv_w, D, D, od, PACK_ROPE_QKV, 0.02, 0);
        e->slots.back().aux_d = Hq; e->slots.back().aux_hd = hd;
        add_slot(e, p + "self_attn.k_proj.weight", {Hkv, D}, (char*)L.qkv_w + (size_t)Hq * D * 2, D, D, od, PACK_ROPE_QKV, 0.02, 0);
        e->slots.back().aux_d = Hkv; e->slots.back().aux_hd = hd;
        add_slot(e, p + "self_attn.v_proj.weight", {Hkv, D}, (char*)L.qkv_w + (size_t)(Hq + Hkv) * D * 2, D, D, od, PACK_PLAIN, 0.02, 0);
        add_slot(e, p + "self_attn.o_proj.weight", {D, Hq}, L.o_w, Hq, Hq, od, PACK_PLAIN, 0.02, 0);
        vec_slot(e, p + "post_attention_layernorm.weight", {D}, L.ln2, 0.05, 1.0);
        add_slot(e, p + "mlp.gate_proj.weight", {I, D}, L.gu_w, D, D, od, PACK_SWIGLU_GATE, 0.02, 0);
        add_slot(e, p + "mlp.up_proj.weight", {I, D}, L.gu_w, D, D, od, PACK_SWIGLU_UP, 0.02, 0);
        add_slot(e, p + "mlp.down_proj.weight", {D, I}, L.down_w, I, I, od, PACK_PLAIN, 0.02, 0);
        if (d.lora_rank > 0) {      // utils/utils.py:243-262 create_lora_config_llava16_vicuna: q, k, v, o, gate, up, down of every layer
            register_lora(e, L.lqkv, p + "self_attn.q_proj", e->Nqkv, D, 0, 3, Hq, 0, PACK_ROPE_QKV, Hq, hd);
            register_lora(e, L.lqkv, p + "self_attn.k_proj", e->Nqkv, D, 1, 3, Hkv, Hq, PACK_ROPE_QKV, Hkv, hd);
            register_lora(e, L.lqkv, p + "self_attn.v_proj", e->Nqkv, D, 2, 3, Hkv, Hq + Hkv, PACK_PLAIN);
            register_lora(e, L.lo, p + "self_attn.o_proj", D, Hq, 0, 1, D, 0, PACK_PLAIN);
            register_lora(e, L.lgu, p + "mlp.gate_proj", 2 * I, D, 0, 2, I, 0, PACK_SWIGLU_GATE);
            register_lora(e, L.lgu, p + "mlp.up_proj", 2 * I, D, 1, 2, I, 0, PACK_SWIGLU_UP);
            register_lora(e, L.ldown, p + "mlp.down_proj", D, I, 0, 1, D, 0, PACK_PLAIN);
        }
    }
    e->norm_w = falloc(e, D);
    vec_slot(e, "language_model.model.norm.weight", {D}, e->norm_w, 0.05, 1.0);
    e->vh = falloc(e, (size_t)d.value_head_dim * D);
    add_slot(e, "value_head.weight", {d.value_head_dim, D}, e->vh, D, D, DT_F32, PACK_PLAIN, 1.0 / std::sqrt((double)D), 0);
    upload_rope_tables(e);
}

void ensure_stage(lr_engine* e, size_t raw_bytes, size_t n_f32) {
    if (raw_bytes > e->stage_raw_cap) {
        if (e->stage_raw) LR_HIP_CHECK(hipFree(e->stage_raw));
        LR_HIP_CHECK(hipMalloc(&e->stage_raw, raw_bytes));
        e->stage_raw_cap = raw_bytes;
    }
    if (n_f32 > e->stage_f32_cap) {
        if (e->stage_f32) LR_HIP_CHECK(hipFree(e->stage_f32));
        LR_HIP_CHECK(hipMalloc((void**)&e->stage_f32, n_f32 * 4));
        e->stage_f32_cap = n_f32;
    }
}

void pack_slot(lr_engine* e, Slot& s, const float* src_f32) {
    if (s.lo_dst && !e->inexact_dev) {
        LR_HIP_CHECK(hipMalloc((void**)&e->inexact_dev, e->wbufs.size() * sizeof(int)));
        LR_HIP_CHECK(hipMemset(e->inexact_dev, 0, e->wbufs.size() * sizeof(int)));
    }
    launch_pack(src_f32, s.dst, s.rows, s.cols, s.ld_dst, s.mode == PACK_TRANSPOSE ? s.cols : s.cols_dst, s.dst_dtype, s.mode, 0,
                s.aux_d, s.aux_hd, s.aux_hdp, s.lo_dst, s.lo_dst ? e->inexact_dev + s.wid : nullptr);
    s.provided = true;
}

void validate_desc(const lr_model_desc& d) {
    auto bad = [](const char* m) { throw std::invalid_argument(m); };
    if (d.struct_size != (int)sizeof(lr_model_desc)) bad("lr_model_desc.struct_size mismatch (ABI)");
    if (d.hidden <= 0 || d.heads <= 0) bad("hidden and heads must be positive");
    if (d.backbone != LR_BACKBONE_PHI3V && d.backbone != LR_BACKBONE_LLAVA_NEXT && d.backbone != LR_BACKBONE_QWEN2_5_VL) bad("unknown backbone");
    if (d.backbone == LR_BACKBONE_PHI3V) {
        if (d.hidden % d.heads) bad("hidden must be divisible by heads");
        const int hd = d.hidden / d.heads;
        if (hd != 96 && hd != 64) bad("Phi-3-V decoder head_dim must be 96 or 64");
        if (d.kv_heads != d.heads || d.head_dim != hd) bad("Phi-3-V: kv_heads must equal heads and head_dim hidden/heads");
    } else if (d.backbone == LR_BACKBONE_QWEN2_5_VL) {
        validate_desc_qwen(d);
    } else {
        if (d.head_dim != 128) bad("LLaVA decoder head_dim must be 128");
        if (d.kv_heads < 1 || d.heads % d.kv_heads) bad("heads must be a multiple of kv_heads");
        if (d.add_cross_attention) bad("the llava branch has no SkipCA (rw_model_general_preference.py:376-397)");
        if (d.n_pinpoints < 1 || d.n_pinpoints > LR_MAX_PINPOINTS) bad("n_pinpoints out of range");
        if (d.image_token_id < 0 || d.image_token_id >= d.vocab_size) bad("image_token_id out of range");
        if (((d.heads + d.kv_heads) * d.head_dim) % 256) bad("(heads + kv_heads) * head_dim must be a multiple of 256");
    }
    const bool qwen = d.backbone == LR_BACKBONE_QWEN2_5_VL;
    if (!qwen) {
        if (d.clip_heads <= 0 || d.clip_patch <= 0) bad("CLIP geometry must be positive");
        if (d.clip_hidden % d.clip_heads || d.clip_hidden / d.clip_heads != 64) bad("CLIP head_dim must be 64");
        if (d.clip_hidden % 64 || d.clip_mlp % 64) bad("widths must be multiples of 64");
        if (d.clip_hidden > 4096) bad("hidden sizes above 4096 are not supported by the norm kernels");
        if (d.clip_image % d.clip_patch || (d.clip_image / d.clip_patch) % 2) bad("CLIP grid must be even");
        if (d.max_crops < 2) bad("capacity fields must be positive (max_crops >= 2)");
    }
    if (d.hidden % 64 || d.intermediate % 64) bad("widths must be multiples of 64");
    if (d.hidden > 4096) bad("hidden sizes above 4096 are not supported by the norm kernels");
    if (d.value_head_dim < 1 || d.value_head_dim > 64) bad("value_head_dim out of range");
    if (d.max_batch < 1 || d.max_seq < 1) bad("capacity fields must be positive");
    if (d.operand_dtype != LR_DT_BF16 && d.operand_dtype != LR_DT_F16) bad("operand_dtype must be BF16 or F16");
    if (d.precise < 0 || d.precise > 2) bad("precise must be 0, 1 or 2");
    if (d.precise == 2 && d.operand_dtype != LR_DT_F16) bad("precise == 2 (e4m3 residual pass) needs F16 operands");
    if (d.w8a8 != 0 && d.w8a8 != 1) bad("w8a8 must be 0 or 1");
    if (d.w8a8 && (d.precise || d.operand_dtype != LR_DT_F16)) bad("w8a8 needs precise == 0 and F16 operands");
    if (d.layers < 0 || d.clip_layers < 0) bad("layer counts must be non-negative");
    if (d.lora_rank < 0 || d.lora_rank > 1024) bad("lora_rank out of range");
    if (d.lora_rank > 0 && d.w8a8) bad("w8a8 runs merged weights only: merge the adapter on the host (lora_rank = 0)");
    if (d.rope_flash_convention != 0 && d.rope_flash_convention != 1) bad("rope_flash_convention must be 0 or 1");
}

}  // namespace

// Pre-norm decoder stack: modeling_phi3_v.py:1144-1205 | modeling_mistral.py MistralDecoderLayer | modeling_qwen2_5_vl.py
// Qwen2_5_VLDecoderLayer (q/k/v bias).  RoPE comes from the per-token (cos, sin) table h->cs.
void alloc_gather_ws(lr_engine* h) {
    const lr_model_desc& d = h->d;
    const size_t R = (size_t)d.max_batch + 256, ob = 2 * (size_t)(1 + h->prec);
    h->xg = (float*)h->dalloc(R * d.hidden * 4, false);
    h->hg = h->dalloc(R * d.hidden * ob, false);
    h->attg = h->dalloc(R * h->Hq * ob, false);
    h->ffg = h->dalloc(R * d.intermediate * ob, false);
}

// DIAGNOSTIC ONLY (LR_ATT_EMU_LO8=1; tools/dbg/attn_lo8_probe.py): round the K / V residuals the attention kernel is about to read to
// e4m3 with one scale per (token, head) -- see launch_emulate_lo8.  No effect unless the variable is set; default form only.
static void emulate_attention_lo8(lr_engine* h, const AttnParams& ap, size_t rows, int kv_heads, int hd, hipStream_t st) {
    static const bool on = [] { const char* e = getenv("LR_ATT_EMU_LO8"); return e && atoi(e) != 0; }();
    if (!on || !ap.lo_off || !h->lo8) return;          // (stages in the default form only: lo8 = the e4m3-residual form is in force)
    launch_emulate_lo8((void*)ap.K, rows, ap.ldq, ap.lo_off + ap.koff, kv_heads, hd, h->op_dt, st);
    launch_emulate_lo8((void*)ap.V, rows, ap.ldq, ap.lo_off + ap.voff, kv_heads, hd, h->op_dt, st);
}

bool run_decoder_stack(lr_engine* h, hipStream_t st, const int64_t* attention_mask, int B, int S, int gather) {
    const lr_model_desc& d = h->d;
    const int D = d.hidden, I = d.intermediate, Rl = B * S;
    const int nl = h->lim_layers >= 0 && h->lim_layers < d.layers ? h->lim_layers : d.layers;
    const float ascale = 1.0f / std::sqrt((float)h->hd);
    const int Hq = h->Hq, Hkv = h->Hkv, Nqkv = h->Nqkv;
    bool pruned = false;
    for (int l = 0; l < nl; ++l) {
        const DecLayer& L = h->dl[l];
        // the operand form of every site of this layer (lr_set_precision_map / lr_set_precision_sites); set right in front of the
        // site's producer, in force until its consumer has been launched
        auto site = [&](int s) { h->set_form(h->site_form(l, s)); };
        site(lr_engine::SITE_QKV);
        {   // qkv projection with RoPE on q,k: fused in the GEMM epilogue when the deep-pipelined kernel runs
            GemmParams gp{h->h, L.qkv_w, h->qkv, L.qkv_b, Rl, Nqkv, D, D, D, Nqkv, EPI_ROPE_OP, ACT_NONE, h->cs, Hq + Hkv, h->hd};
            const bool tiles_ok = (Hq + Hkv) % 256 == 0;
            launch_norm_rows(h->x, L.ln1, nullptr, h->h, Rl, D, d.rms_eps, h->op_dt, st, h->prec, 1, tiles_ok ? lo8_norm_target(h, gp) : nullptr);
            GemmParams probe = gp;
            apply_prec_base(h, probe);
            if (tiles_ok && (L.lqkv.k2 > 0 || w8a8_eligible(h, probe) || lo8_eligible(h, probe) || gemm_bt_is_deep(probe, h->gemm_tile))) {
                gemm_p(h, st, gp, &L.lqkv);
            } else {
                // (pre-RoPE fp32 projections: only this fallback needs them, so the 3.1 GB at B=32 are allocated on its first use)
                if (!h->qkv32) h->qkv32 = (float*)h->dalloc(((size_t)d.max_batch * d.max_seq + 256) * Nqkv * 4, false);
                gemm(h, st, h->h, L.qkv_w, h->qkv32, L.qkv_b, Rl, Nqkv, D, D, D, Nqkv, EPI_OUT_F32, ACT_NONE, &L.lqkv);
                launch_rope_split(h->qkv32, h->cs, h->qkv, Rl, Hq + Hkv, Hkv, h->hd, h->op_dt, st, h->prec);
            }
        }
        AttnParams ap{h->qkv, h->qkv, h->qkv, h->att, attention_mask, h->tstat + 1, 4, Nqkv, Hq, 0, Hq, Hq + Hkv, S, d.heads, ascale,
                      d.heads / d.kv_heads};
        if (gather && l == nl - 1) {
            // Last layer: every row still feeds K and V, but only one row per sequence is read afterwards (rw_model:408-421).  Its
            // query sits in the qkv rows as usual; the attention kernel runs the one query tile that holds it (same arithmetic as
            // in the full launch) and stores B rows; the residual rows are gathered and o_proj / norm / MLP run on M = B rows --
            // the GEMM kernels never look at M when they pick their form, so these rows come out bit-identical to a full layer.
            ap.O = h->attg;
            ap.qsel = h->tstat; ap.qsel_stride = 4; ap.qsel_last = gather == 2 ? 1 : 0;
            site(lr_engine::SITE_ATTN);
            apply_prec(h, ap);
            emulate_attention_lo8(h, ap, Rl, d.kv_heads, h->hd, st);
            launch_attention(ap, B, h->hd, true, h->op_dt, st);
            launch_gather_norm_rows(h->x, h->tstat, S, gather == 2 ? 1 : 0, nullptr, 0.f, h->xg, B, D, st);      // (no weight: a plain row gather)
            site(lr_engine::SITE_O);
            gemm(h, st, h->attg, L.o_w, h->xg, nullptr, B, D, Hq, Hq, Hq, D, EPI_RESADD_F32, ACT_NONE, &L.lo);
            site(lr_engine::SITE_GATE_UP);
            launch_norm_rows(h->xg, L.ln2, nullptr, h->hg, B, D, d.rms_eps, h->op_dt, st, h->prec, 1,
                             lo8_norm_target(h, GemmParams{h->hg, L.gu_w, h->ffg, nullptr, B, 2 * I, D, D, D, I, EPI_SWIGLU_OP, ACT_NONE, nullptr, 0, 0}));
            gemm(h, st, h->hg, L.gu_w, h->ffg, nullptr, B, 2 * I, D, D, D, I, EPI_SWIGLU_OP, ACT_NONE, &L.lgu, L.down_w, D, h->site_form(l, lr_engine::SITE_DOWN));
            { const void* enc = h->pre_enc; const unsigned char* esc = h->pre_enc_sc; const bool e8 = h->pre_enc_hi8;
              site(lr_engine::SITE_DOWN); h->pre_enc = enc; h->pre_enc_sc = esc; h->pre_enc_hi8 = e8; }      // (ffg's residual format was chosen for THIS site: keep its mark)
            gemm(h, st, h->ffg, L.down_w, h->xg, nullptr, B, D, I, I, I, D, EPI_RESADD_F32, ACT_NONE, &L.ldown);
            pruned = true;
            break;
        }
        site(lr_engine::SITE_ATTN);
        apply_prec(h, ap);
        emulate_attention_lo8(h, ap, Rl, d.kv_heads, h->hd, st);
        launch_attention(ap, B, h->hd, true, h->op_dt, st);
        site(lr_engine::SITE_O);
        gemm(h, st, h->att, L.o_w, h->x, nullptr, Rl, D, Hq, Hq, Hq, D, EPI_RESADD_F32, ACT_NONE, &L.lo);
        site(lr_engine::SITE_GATE_UP);
        launch_norm_rows(h->x, L.ln2, nullptr, h->h, Rl, D, d.rms_eps, h->op_dt, st, h->prec, 1,
                         lo8_norm_target(h, GemmParams{h->h, L.gu_w, h->ff, nullptr, Rl, 2 * I, D, D, D, I, EPI_SWIGLU_OP, ACT_NONE, nullptr, 0, 0}));
        gemm(h, st, h->h, L.gu_w, h->ff, nullptr, Rl, 2 * I, D, D, D, I, EPI_SWIGLU_OP, ACT_NONE, &L.lgu, L.down_w, D, h->site_form(l, lr_engine::SITE_DOWN));      // ff is down's operand
        { const void* enc = h->pre_enc; const unsigned char* esc = h->pre_enc_sc; const bool e8 = h->pre_enc_hi8;
          site(lr_engine::SITE_DOWN); h->pre_enc = enc; h->pre_enc_sc = esc; h->pre_enc_hi8 = e8; }
        gemm(h, st, h->ff, L.down_w, h->x, nullptr, Rl, D, I, I, I, D, EPI_RESADD_F32, ACT_NONE, &L.ldown);
    }
    h->set_form(-1);
    return pruned;
}

void alloc_mean_pool(lr_engine* h, size_t rows, size_t vcap) {
    const size_t D = h->d.hidden;
    h->mh_h = (float*)h->dalloc(rows * D * 4, false);
    if (h->d.add_cross_attention && !h->qwen) {
        h->mh_t1 = (float*)h->dalloc(rows * D * 4, false);
        h->mh_t2 = (float*)h->dalloc(rows * D * 4, false);
        h->mh_sc = (float*)h->dalloc(((size_t)h->d.max_seq + 256) * vcap * 4, false);
    }
}

void run_mean_pool_head(lr_engine* h, hipStream_t st, const int64_t* attention_mask, int B, int S, const int* voff_host, int Vmax,
                        const float* u, bool final_norm, float* rewards_out) {
    const lr_model_desc& d = h->d;
    const int D = d.hidden, Rl = B * S;
    if (final_norm) launch_rms_rows_f32(h->x, h->norm_w, d.rms_eps, h->mh_h, Rl, D, st);
    else LR_HIP_CHECK(hipMemcpyAsync(h->mh_h, h->x, (size_t)Rl * D * 4, hipMemcpyDeviceToDevice, st));
    const float* o = nullptr;
    if (d.add_cross_attention && !h->qwen) {
        // SkipCA for every token (rw_model:376-386), folded as in the gathered-row tail: score = (W_k^T W_q h) . e_j, out = W_v sum_j p_j e_j
        launch_gemm_f32(h->mh_h, h->Wq, 0, 1, h->mh_t1, Rl, D, D, D, D, D, 1.f, st);
        launch_gemm_f32(h->mh_t1, h->WkT, 0, 1, h->mh_t2, Rl, D, D, D, D, D, 1.f, st);
        const int ldsc = h->Vcap;
        for (int b = 0; b < B; ++b) {
            const int vb = voff_host[b + 1] - voff_host[b];
            const float* evb = h->ev + (size_t)voff_host[b] * D;
            float* ctx = h->mh_t1 + (size_t)b * S * D;
            launch_gemm_f32(h->mh_t2 + (size_t)b * S * D, evb, 0, 1, h->mh_sc, S, vb, D, D, D, ldsc, 1.0f / std::sqrt((float)D), st);
            launch_ca_softmax_pad(h->mh_sc, S, ldsc, vb, Vmax, st);
            launch_gemm_f32(h->mh_sc, evb, 0, 0, ctx, S, D, vb, ldsc, D, D, 1.f, st);
        }
        launch_gemm_f32(h->mh_t1, h->Wv, 0, 1, h->mh_t2, Rl, D, D, D, D, D, 1.f, st);
        o = h->mh_t2;
    }
    const bool ca = d.add_cross_attention != 0;
    launch_ca_pool(h->mh_h, o, ca ? u : nullptr, ca ? h->ca_w : nullptr, d.ca_eps, attention_mask, h->hL, B, S, D, st);
    launch_reward_head(h->hL, nullptr, nullptr, 0.f, h->vh, d.value_head_dim, rewards_out, B, D, st);
}

extern "C" {

int lr_abi_version(void) { return LR_ABI_VERSION; }

const char* lr_last_error(lr_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int lr_create(const lr_model_desc* desc, int device, lr_handle* out) {
    if (!desc || !out) { g_create_error = "lr_create: null argument"; return LR_EINVAL; }
    lr_engine* e = nullptr;
    int rc = guarded(nullptr, [&] {
        validate_desc(*desc);
        int ndev = 0;
        LR_HIP_CHECK(hipGetDeviceCount(&ndev));
        if (device < 0 || device >= ndev) throw std::invalid_argument("lr_create: no such HIP device");
        LR_HIP_CHECK(hipSetDevice(device));
        e = new lr_engine();
        e->d = *desc;
        e->device = device;
        e->op_dt = desc->operand_dtype == LR_DT_F16 ? DT_F16 : DT_BF16;
        e->prec = e->prec0 = desc->precise ? 1 : 0;
        e->lo8 = e->lo8_0 = desc->precise == 2 ? 1 : 0;
        e->sched_mem = (int*)e->dalloc(64, true);
        e->w8a8 = desc->w8a8 ? 1 : 0;
        e->llava = desc->backbone == LR_BACKBONE_LLAVA_NEXT;
        e->qwen = desc->backbone == LR_BACKBONE_QWEN2_5_VL;
        if (!e->qwen) {
            e->G = desc->clip_image / desc->clip_patch;
            e->T = e->G * e->G + 1;
            e->Kpatch = 3 * desc->clip_patch * desc->clip_patch;
            e->Kpad = (e->Kpatch + 63) / 64 * 64;
        }
        e->hd = desc->head_dim;
        e->half = e->hd / 2;
        e->lora_rp = (desc->lora_rank + 63) / 64 * 64;
        e->Hq = desc->heads * e->hd; e->Hkv = desc->kv_heads * e->hd; e->Nqkv = e->Hq + 2 * e->Hkv;
        if (e->qwen) {
            build_weight_table_qwen(e);
        } else if (e->llava) {
            int vmax = 0, cmax = 0;
            for (int i = 0; i < desc->n_pinpoints; ++i) {
                const int gh = desc->pinpoints[2 * i] / desc->clip_image, gw = desc->pinpoints[2 * i + 1] / desc->clip_image;
                if (gh < 1 || gw < 1) throw std::invalid_argument("pinpoints must be positive multiples of the crop size");
                vmax = std::max(vmax, gh * e->G * (gw * e->G + 1));
                cmax = std::max(cmax, gh * gw + 1);
            }
            if (desc->max_crops < cmax) throw std::invalid_argument("max_crops is smaller than the largest anyres grid + 1");
            e->Vcap = e->G * e->G + vmax;
            build_weight_table_llava(e);
        } else {
            const int g2 = e->G / 2;
            e->Vcap = desc->max_crops * g2 * g2 + 1 + desc->max_crops * g2;
            build_weight_table_phi(e);
        }
    });
    if (rc != LR_OK) {
        if (e) { g_create_error = g_create_error.empty() ? e->err : g_create_error; lr_destroy(e); }
        return rc;
    }
    *out = e;
    return LR_OK;
}

int lr_destroy(lr_handle h) {
    if (!h) return LR_OK;
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    for (void* p : h->allocs) hipFree(p);
    if (h->stage_raw) hipFree(h->stage_raw);
    if (h->stage_f32) hipFree(h->stage_f32);
    if (h->inexact_dev) hipFree(h->inexact_dev);
    if (h->tab_host) hipHostFree(h->tab_host);
    for (int i = 0; i < lr_engine::NSLOT; ++i) if (h->tab_ev[i]) hipEventDestroy(h->tab_ev[i]);
    delete h;
    return LR_OK;
}

int lr_num_weights(lr_handle h) { return h ? (int)h->slots.size() : 0; }
const char* lr_weight_name(lr_handle h, int i) {
    if (!h || i < 0 || i >= (int)h->slots.size()) return nullptr;
    return h->slots[i].name.c_str();
}

int lr_upload_weight(lr_handle h, const char* name, const void* data, const int64_t* shape, int ndim, int dtype, int is_device) {
    if (!h) return LR_EINVAL;
    return guarded(h, [&] {
        if (!name || !data || !shape) throw std::invalid_argument("lr_upload_weight: null argument");
        auto it = h->index.find(name);
        if (it == h->index.end()) throw std::invalid_argument(std::string("lr_upload_weight: unknown tensor ") + name);
        Slot& s = h->slots[it->second];
        if ((int)s.shape.size() != ndim) throw std::invalid_argument(std::string("lr_upload_weight: rank mismatch for ") + name);
        size_t n = 1;
        for (int i = 0; i < ndim; ++i) {
            if (shape[i] != s.shape[i]) throw std::invalid_argument(std::string("lr_upload_weight: shape mismatch for ") + name);
            n *= (size_t)shape[i];
        }
        const size_t esz = dtype == LR_DT_F32 ? 4 : 2;
        if (dtype != LR_DT_F32 && dtype != LR_DT_BF16 && dtype != LR_DT_F16) throw std::invalid_argument("lr_upload_weight: bad dtype");
        ensure_stage(h, is_device ? 0 : n * esz, n);
        const void* dev_src = data;
        if (!is_device) {
            LR_HIP_CHECK(hipMemcpy(h->stage_raw, data, n * esz, hipMemcpyHostToDevice));
            dev_src = h->stage_raw;
        }
        const float* f32 = (const float*)dev_src;
        if (dtype != LR_DT_F32) {
            launch_cvt_to_f32(dev_src, dtype == LR_DT_F16 ? DT_F16 : DT_BF16, h->stage_f32, n, 0);
            f32 = h->stage_f32;
        }
        if (s.wid >= 0 && s.lo_dst) {          // split-operand mode: this tensor's buffer has a residual twin
            const lr_engine::WBuf& wb = h->wbufs[s.wid];
            if (h->own8.count(wb.hi)) {        // adapter matrix: its e4m3 twin is a separate buffer, rebuilt at the next launch
                h->w8exp.erase(wb.hi); h->w8exp2.erase(wb.hi);
            } else if (h->w8exp.count(wb.hi)) {       // its e4m3 twin has been prepared (default parity mode, after lr_finalize)
                const bool whole = (size_t)s.rows * s.ld_dst * 2 == wb.bytes && s.dst == wb.hi;
                const bool was_inexact = !h->inexact.empty() && h->inexact[s.wid];
                // an exact buffer's residual twin holds e4m3(W) now (its 16-bit residuals were all zero); an inexact buffer's still
                // holds the 16-bit residuals (its e4m3 rows live in pair8)
                if (!whole && !was_inexact) LR_HIP_CHECK(hipMemset(wb.lo, 0, wb.bytes));
                h->w8exp.erase(wb.hi); h->w8exp2.erase(wb.hi);
            }
        }
        if (s.wid >= 0 && s.lo_dst && h->inexact_dev && (size_t)s.rows * s.ld_dst * 2 == h->wbufs[s.wid].bytes && s.dst == h->wbufs[s.wid].hi)
            LR_HIP_CHECK(hipMemset(h->inexact_dev + s.wid, 0, sizeof(int)));      // the whole buffer is replaced: it may be exact again
        pack_slot(h, s, f32);
        ++h->weights_epoch;
        h->w8.clear();             // W8A8 twins are separate buffers, rebuilt from the packed weights at the next launch
        LR_HIP_CHECK(hipStreamSynchronize(0));
        if (h->finalized && h->inexact_dev) {    // a re-upload after lr_finalize may have made a buffer inexact -- or exact again
            LR_HIP_CHECK(hipMemcpy(h->inexact.data(), h->inexact_dev, h->wbufs.size() * sizeof(int), hipMemcpyDeviceToHost));
            h->release_stale_pair8();
        }
    });
}

int lr_synth_weights(lr_handle h, uint64_t seed) { return lr_synth_weights_ex(h, seed, 0); }

int lr_synth_weights_ex(lr_handle h, uint64_t seed, int flags) {
    if (!h) return LR_EINVAL;
    return guarded(h, [&] {
        if (h->inexact_dev) LR_HIP_CHECK(hipMemset(h->inexact_dev, 0, h->wbufs.size() * sizeof(int)));   // every buffer is rewritten below
        for (Slot& s : h->slots) {
            const size_t n = (size_t)s.rows * s.cols;
            ensure_stage(h, 0, n);
            const uint64_t ts = tensor_seed(seed, s.name.c_str());
            if (flags & 6) {        // weight profiles (outliers / e4m3-valued): applied to the un-rounded fill, then rounded
                launch_synth_fill(h->stage_f32, n, ts, uniform_scale(s.std_), (float)s.offset, 0, 0);
                launch_synth_profile(h->stage_f32, s.rows, s.cols, seed, ts, s.name.c_str(), s.std_, s.offset, flags, 0);
            } else
                launch_synth_fill(h->stage_f32, n, ts, uniform_scale(s.std_), (float)s.offset, (flags & 1) ? 0 : 1, 0);
            pack_slot(h, s, h->stage_f32);
        }
        h->w8exp.clear(); h->w8exp2.clear(); h->w8.clear();
        ++h->weights_epoch;
        LR_HIP_CHECK(hipStreamSynchronize(0));
        if (h->finalized && h->inexact_dev) {    // re-synthesised after lr_finalize (fp32-valued profile): a buffer may have become inexact
            h->inexact.resize(h->wbufs.size(), 0);
            LR_HIP_CHECK(hipMemcpy(h->inexact.data(), h->inexact_dev, h->wbufs.size() * sizeof(int), hipMemcpyDeviceToHost));
            h->release_stale_pair8();
        }
    });
}

size_t lr_workspace_bytes(lr_handle h) { return h ? h->ws_bytes : 0; }

int lr_finalize(lr_handle h) {
    if (!h) return LR_EINVAL;
    return guarded(h, [&] {
        if (h->finalized) return;
        for (const Slot& s : h->slots)
            if (!s.provided) throw std::logic_error("lr_finalize: tensor never provided: " + s.name);
        if (h->inexact_dev) {           // which operand-weight buffers carry non-zero rounding residuals (split-operand mode)
            h->inexact.assign(h->wbufs.size(), 0);
            LR_HIP_CHECK(hipMemcpy(h->inexact.data(), h->inexact_dev, h->wbufs.size() * sizeof(int), hipMemcpyDeviceToHost));
        }
        if (h->stage_raw) { LR_HIP_CHECK(hipFree(h->stage_raw)); h->stage_raw = nullptr; h->stage_raw_cap = 0; }
        if (h->stage_f32) { LR_HIP_CHECK(hipFree(h->stage_f32)); h->stage_f32 = nullptr; h->stage_f32_cap = 0; }
        if (h->qwen) { finalize_qwen(h); h->finalized = true; return; }
        const lr_model_desc& d = h->d;
        const size_t B = d.max_batch, S = d.max_seq, C = d.max_crops;
        const size_t Hc = d.clip_hidden, Mc = d.clip_mlp, D = d.hidden, I = d.intermediate;
        const size_t PAD = 256;
        const size_t NC = B * C, Rc = NC * h->T + PAD, Rp = NC * (h->T - 1) + PAD, SV = B * (size_t)h->Vcap + PAD, Rl = B * S + PAD;
        auto W = [&](size_t bytes) { return h->dalloc(bytes, false); };
        const size_t ob = 2 * (size_t)(1 + h->prec);      // bytes per operand element (hi [+ lo])
        h->patchA = W(Rp * h->Kpad * ob); h->patch_out = (float*)W(Rp * Hc * 4);
        h->clip_x = (float*)W(Rc * Hc * 4); h->clip_h = W(Rc * Hc * ob); h->clip_qkv = W(Rc * 3 * Hc * ob);
        h->clip_att = W(Rc * Hc * ob); h->clip_ff = W(Rc * Mc * ob);
        if (h->llava) {       // projector runs on every crop token, packing afterwards
            h->hdA = W(Rp * Hc * ob); h->proj1 = W(Rp * D * ob); h->pf32 = (float*)W(Rp * D * 4);
        } else {              // HD-merged rows go through the projector
            h->hdA = W(SV * 4 * Hc * ob); h->proj1 = W(SV * D * ob);
        }
        h->ev = (float*)W(SV * D * 4);
        h->x = (float*)W(Rl * D * 4); h->h = W(Rl * D * ob); h->qkv = W(Rl * h->Nqkv * ob);
        h->att = W(Rl * h->Hq * ob); h->ff = W(Rl * I * ob); h->cs = (float*)W(Rl * h->hd * 4);
        if (h->lora_k2max > 0) h->lt = W(Rl * (size_t)h->lora_k2max * ob);
        h->pos_ids = (int*)W(Rl * 4); h->img_row = (int*)W(Rl * 4); h->tstat = (int*)W(B * 16);
        h->hL = (float*)W(B * D * 4); h->tq = (float*)W(B * D * 4); h->tkq = (float*)W(B * D * 4);
        h->tsc = (float*)W(B * (size_t)h->Vcap * 4); h->tctx = (float*)W(B * D * 4); h->tao = (float*)W(B * D * 4);
        if (d.mean_hidden_state) alloc_mean_pool(h, Rl, (size_t)h->Vcap);
        alloc_gather_ws(h);
        // tables: crop_src[NC] | per-sample geometry [B] (HdSample or LlavaSample) | voff[B+1]
        h->tab_bytes = ((NC * 4 + B * sizeof(LlavaSample) + (B + 1) * 4) + 255) & ~(size_t)255;
        LR_HIP_CHECK(hipHostMalloc((void**)&h->tab_host, h->tab_bytes * lr_engine::NSLOT));
        h->tab_dev = (char*)W(h->tab_bytes * lr_engine::NSLOT);
        for (int i = 0; i < lr_engine::NSLOT; ++i) LR_HIP_CHECK(hipEventCreateWithFlags(&h->tab_ev[i], hipEventDisableTiming));
        LR_HIP_CHECK(hipMemset(h->tstat, 0, B * 16));
        {   // e4m3 twins of the GEMM weights (default parity mode / W8A8 mode)
            std::vector<GemmWeight> ws;
            ws.push_back({h->patch_w, (int)Hc, h->Kpad, Rp});
            for (const ClipLayer& c : h->cl) {
                ws.push_back({c.qkv_w, (int)(3 * Hc), (int)Hc, Rc}); ws.push_back({c.out_w, (int)Hc, (int)Hc, Rc});
                ws.push_back({c.fc1_w, (int)Mc, (int)Hc, Rc}); ws.push_back({c.fc2_w, (int)Hc, (int)Mc, Rc});
            }
            if (h->llava) { ws.push_back({h->p0_w, (int)D, (int)Hc, Rp}); ws.push_back({h->p2_w, (int)D, (int)D, Rp}); }
            else { ws.push_back({h->p0_w, (int)D, (int)(4 * Hc), SV}); ws.push_back({h->p2_w, (int)D, (int)D, SV}); }
            for (const DecLayer& L : h->dl) {
                ws.push_back({L.qkv_w, h->Nqkv, (int)D, Rl}); ws.push_back({L.o_w, (int)D, h->Hq, Rl});
                ws.push_back({L.gu_w, (int)(2 * I), (int)D, Rl}); ws.push_back({L.down_w, (int)D, (int)I, Rl});
                if (L.lqkv.k2) {
                    ws.push_back({L.lqkv.A, L.lqkv.k2, (int)D, Rl}); ws.push_back({L.lo.A, L.lo.k2, h->Hq, Rl});
                    ws.push_back({L.lgu.A, L.lgu.k2, (int)D, Rl}); ws.push_back({L.ldown.A, L.ldown.k2, (int)I, Rl});
                }
            }
            prepare_twins(h, ws);
        }
        h->finalized = true;
    });
}

int lr_set_layer_limits(lr_handle h, int n_clip_layers, int n_layers) {
    if (!h) return LR_EINVAL;
    h->lim_clip = n_clip_layers; h->lim_layers = n_layers;
    return LR_OK;
}
int lr_set_precision_map(lr_handle h, int clip_form, int decoder_mid_form, int decoder_first, int decoder_last) {
    if (!h) return LR_EINVAL;
    return guarded(h, [&] {
        auto ok = [&](int m) { return m >= -1 && m <= 2 && (m <= 0 || (h->prec0 && h->op_dt == DT_F16) || (m == 1 && h->prec0)) && (m != 2 || h->lo8_0); };
        if (!ok(clip_form) || !ok(decoder_mid_form) || decoder_first < 0 || decoder_last < 0)
            throw std::invalid_argument("lr_set_precision_map: a stage can take form 0, or a split form the handle was created with "
                                        "(precise >= 1 for form 1, precise == 2 for form 2)");
        h->pm_clip = clip_form; h->pm_mid = decoder_mid_form; h->pm_first = decoder_first; h->pm_last = decoder_last;
    });
}
int lr_set_precision_sites(lr_handle h, int qkv_form, int attention_form, int o_proj_form, int gate_up_form, int down_form) {
    if (!h) return LR_EINVAL;
    return guarded(h, [&] {
        const int f[lr_engine::N_SITES] = {qkv_form, attention_form, o_proj_form, gate_up_form, down_form};
        for (int v : f)
            if (v != -1 && !(v == 1 && h->prec0) && !(v == 2 && h->lo8_0))
                throw std::invalid_argument("lr_set_precision_sites: a site takes -1 (the layer's form), 1 (16-bit residual passes; handle created with "
                                            "precise >= 1) or 2 (e4m3 residual passes; precise == 2)");
        for (int i = 0; i < lr_engine::N_SITES; ++i) h->pm_site[i] = f[i];
    });
}
int lr_set_attention_lazy_threshold(lr_handle h, float default_stages, float strict_stages) {
    if (!h) return LR_EINVAL;
    return guarded(h, [&] {
        if (!(default_stages >= 0.f && default_stages <= 15.f && strict_stages >= 0.f && strict_stages <= 15.f))
            throw std::invalid_argument("lr_set_attention_lazy_threshold: 0 (exact maximum) .. 15 log2 units");
        h->att_lazy_t = default_stages;
        h->att_lazy_t_strict = strict_stages;
    });
}
uint64_t lr_weights_epoch(lr_handle h) { return h ? h->weights_epoch : 0; }
int lr_set_gemm_tile(lr_handle h, int tile) {
    if (!h || tile < -1 || tile > 15) return LR_EINVAL;
    h->gemm_tile = tile;
    return LR_OK;
}

int lr_forward(lr_handle h, const int64_t* input_ids, const int64_t* attention_mask, const void* pixel_values, int pix_dtype,
               const int64_t* image_sizes_host, int B, int S, int n_crops, int flags, float* rewards_out, void* hip_stream) {
    if (!h) return LR_EINVAL;
    return guarded(h, [&] {
        if (!h->finalized) throw std::logic_error("lr_forward: call lr_finalize first");
        if (h->qwen) throw std::logic_error("lr_forward: Qwen2.5-VL handles take lr_forward_qwen (inputs_batch carries image_grid_thw)");
        if (!input_ids || !attention_mask || !pixel_values || !image_sizes_host || !rewards_out)
            throw std::invalid_argument("lr_forward: null argument (every row must carry an image, modeling_phi3_v.py:252)");
        const lr_model_desc& d = h->d;
        if (B < 1 || B > d.max_batch) throw std::invalid_argument("lr_forward: batch exceeds max_batch");
        if (S < 1 || S > d.max_seq) throw std::invalid_argument("lr_forward: sequence exceeds max_seq");
        if (n_crops < 2 || n_crops > d.max_crops) throw std::invalid_argument("lr_forward: n_crops exceeds max_crops");
        if (pix_dtype != LR_DT_F32 && pix_dtype != LR_DT_BF16) throw std::invalid_argument("lr_forward: pixel dtype must be F32 or BF16");
        hipStream_t st = (hipStream_t)hip_stream;
        const int Hc = d.clip_hidden, Mc = d.clip_mlp, D = d.hidden, I = d.intermediate, T = h->T, g2 = h->G / 2;
        const int img = d.clip_image;

        // ---- host plan: active crops, per-sample vision geometry, image-token offsets ----
        // Phi-3-V: modeling_phi3_v.py:276-297 (HD crop grid);  LLaVA: modeling_llava_next.py:41-146,265-335 (anyres + unpad)
        const int slot = h->slot_i; h->slot_i = (h->slot_i + 1) % lr_engine::NSLOT;
        if (h->tab_used[slot]) LR_HIP_CHECK(hipEventSynchronize(h->tab_ev[slot]));   // bounds host run-ahead to NSLOT passes
        char* th = h->tab_host + (size_t)slot * h->tab_bytes;
        char* td = h->tab_dev + (size_t)slot * h->tab_bytes;
        const size_t NCcap = (size_t)d.max_batch * d.max_crops;
        const size_t smp_off = NCcap * 4, voff_off = smp_off + (size_t)d.max_batch * sizeof(LlavaSample);
        int* crop_src = (int*)th;
        HdSample* smp = (HdSample*)(th + smp_off);
        LlavaSample* lsmp = (LlavaSample*)(th + smp_off);
        int* voff = (int*)(th + voff_off);
        int NC = 0, SV = 0, Vmax = 0;
        for (int b = 0; b < B; ++b) {
            const int64_t hh = image_sizes_host[2 * b], ww = image_sizes_host[2 * b + 1];
            int ncr, V;
            if (h->llava) {
                if (hh <= 0 || ww <= 0) throw std::invalid_argument("lr_forward: image_sizes must be positive");
                // transformers select_best_resolution: max effective resolution, then min waste
                long best_eff = 0, best_waste = 0; int bh = 0, bw = 0;
                for (int i = 0; i < d.n_pinpoints; ++i) {
                    const int ph = d.pinpoints[2 * i], pw = d.pinpoints[2 * i + 1];
                    const double sc = std::min((double)pw / (double)ww, (double)ph / (double)hh);
                    const long dw = (long)((double)ww * sc), dh = (long)((double)hh * sc);
                    const long eff = std::min(dw * dh, (long)(ww * hh)), waste = (long)pw * ph - eff;
                    if (eff > best_eff || (eff == best_eff && (bh == 0 || waste < best_waste))) { best_eff = eff; best_waste = waste; bh = ph; bw = pw; }
                }
                const int gh = bh / img, gw = bw / img, ch = gh * h->G, cw = gw * h->G;
                int r0 = 0, r1 = ch, c0 = 0, c1 = cw;
                // unpad_image (modeling_llava_next.py:109-146): new extent = int(round(x, 7)) -- 7 decimals, then truncation
                auto trunc7 = [](double x) { return (int)(std::round(x * 1e7) / 1e7); };
                if ((double)ww / (double)hh > (double)cw / (double)ch) {           // rows were padded
                    const int pad = (ch - trunc7((double)hh * ((double)cw / (double)ww))) / 2; r0 = pad; r1 = ch - pad;
                } else {                                                           // columns were padded
                    const int pad = (cw - trunc7((double)ww * ((double)ch / (double)hh))) / 2; c0 = pad; c1 = cw - pad;
                }
                ncr = 1 + gh * gw;
                V = h->G * h->G + (r1 - r0) * (c1 - c0 + 1);
                lsmp[b] = LlavaSample{gh, gw, r0, r1, c0, c1, NC, SV};
            } else {
                if (hh <= 0 || ww <= 0 || hh % img || ww % img) throw std::invalid_argument("lr_forward: image_sizes must be positive multiples of the crop size");
                const int hc = (int)(hh / img), wc = (int)(ww / img);
                ncr = hc * wc + 1;
                smp[b] = HdSample{hc, wc, NC, SV};
                V = hc * g2 * (wc * g2 + 1) + 1 + g2 * (g2 + 1);
            }
            if (ncr > n_crops) throw std::invalid_argument("lr_forward: image_sizes needs more crops than pixel_values holds");
            voff[b] = SV;
            for (int c = 0; c < ncr; ++c) crop_src[NC++] = b * n_crops + c;
            SV += V;
            Vmax = V > Vmax ? V : Vmax;
        }
        voff[B] = SV;
        LR_HIP_CHECK(hipMemcpyAsync(td, th, h->tab_bytes, hipMemcpyHostToDevice, st));
        LR_HIP_CHECK(hipEventRecord(h->tab_ev[slot], st));
        h->tab_used[slot] = true;
        const int* d_crop_src = (const int*)td;
        const HdSample* d_smp = (const HdSample*)(td + smp_off);
        const LlavaSample* d_lsmp = (const LlavaSample*)(td + smp_off);
        const int* d_voff = (const int*)(td + voff_off);
        h->lastB = B; h->lastS = S; h->lastNC = NC; h->lastSV = SV; h->lastVmax = Vmax;
        h->last_voff.assign(voff, voff + B + 1);

        // ---- CLIP tower (utils/utils.py:266-273) ----
        const int Rp = NC * (T - 1), Rc = NC * T;
        h->set_form(h->pm_clip);
        launch_im2col(pixel_values, pix_dtype == LR_DT_F32 ? DT_F32 : DT_BF16, d_crop_src, NC, img, d.clip_patch, h->Kpad, h->patchA, h->op_dt, st, h->prec);
        gemm(h, st, h->patchA, h->patch_w, h->patch_out, nullptr, Rp, Hc, h->Kpad, h->Kpad, h->Kpad, Hc, EPI_OUT_F32, ACT_NONE);
        launch_clip_embed(h->patch_out, h->cls, h->pos, h->pre_w, h->pre_b, h->clip_x, NC, T, Hc, d.clip_ln_eps, st);
        const int ncl = h->lim_clip >= 0 && h->lim_clip < d.clip_layers ? h->lim_clip : d.clip_layers;
        for (int l = 0; l < ncl; ++l) {
            const ClipLayer& c = h->cl[l];
            launch_norm_rows(h->clip_x, c.ln1_w, c.ln1_b, h->clip_h, Rc, Hc, d.clip_ln_eps, h->op_dt, st, h->prec, 1,
                             lo8_norm_target(h, GemmParams{h->clip_h, c.qkv_w, h->clip_qkv, c.qkv_b, Rc, 3 * Hc, Hc, Hc, Hc, 3 * Hc, EPI_OUT_OP, ACT_NONE, nullptr, 0, 0}));
            gemm(h, st, h->clip_h, c.qkv_w, h->clip_qkv, c.qkv_b, Rc, 3 * Hc, Hc, Hc, Hc, 3 * Hc, EPI_OUT_OP, ACT_NONE);
            AttnParams ap{h->clip_qkv, h->clip_qkv, h->clip_qkv, h->clip_att, nullptr, nullptr, 0, 3 * Hc, Hc, 0, Hc, 2 * Hc, T, d.clip_heads, 0.125f, 1};
            apply_prec(h, ap);
            emulate_attention_lo8(h, ap, Rc, d.clip_heads, 64, st);
            launch_attention(ap, NC, 64, false, h->op_dt, st);
            gemm(h, st, h->clip_att, c.out_w, h->clip_x, c.out_b, Rc, Hc, Hc, Hc, Hc, Hc, EPI_RESADD_F32, ACT_NONE);
            launch_norm_rows(h->clip_x, c.ln2_w, c.ln2_b, h->clip_h, Rc, Hc, d.clip_ln_eps, h->op_dt, st, h->prec, 1,
                             lo8_norm_target(h, GemmParams{h->clip_h, c.fc1_w, h->clip_ff, c.fc1_b, Rc, Mc, Hc, Hc, Hc, Mc, EPI_OUT_OP, ACT_QUICK_GELU, nullptr, 0, 0}));
            gemm(h, st, h->clip_h, c.fc1_w, h->clip_ff, c.fc1_b, Rc, Mc, Hc, Hc, Hc, Mc, EPI_OUT_OP, ACT_QUICK_GELU, nullptr, c.fc2_w, Hc);
            gemm(h, st, h->clip_ff, c.fc2_w, h->clip_x, c.fc2_b, Rc, Hc, Mc, Mc, Mc, Hc, EPI_RESADD_F32, ACT_NONE);
        }
        if (h->llava) {
            // ---- per-token projector, then anyres packing (modeling_llava_next.py get_image_features/pack_image_features) ----
            launch_clip_tokens(h->clip_x, h->hdA, NC, T, Hc, h->op_dt, st, h->prec);
            gemm(h, st, h->hdA, h->p0_w, h->proj1, h->p0_b, Rp, D, Hc, Hc, Hc, D, EPI_OUT_OP, ACT_GELU_ERF, nullptr, h->p2_w, D);
            gemm(h, st, h->proj1, h->p2_w, h->pf32, h->p2_b, Rp, D, D, D, D, D, EPI_OUT_F32, ACT_NONE);
            launch_llava_pack(h->pf32, d_lsmp, B, SV, h->G, D, h->newline, h->ev, st);
        } else {
            // ---- HD transform + projector (modeling_phi3_v.py:254-303) ----
            launch_hd_gather(h->clip_x, d_smp, B, SV, T, Hc, h->sub_gn, h->glb_gn, h->hdA, h->op_dt, st, h->prec);
            gemm(h, st, h->hdA, h->p0_w, h->proj1, h->p0_b, SV, D, 4 * Hc, 4 * Hc, 4 * Hc, D, EPI_OUT_OP, ACT_GELU_ERF, nullptr, h->p2_w, D);
            gemm(h, st, h->proj1, h->p2_w, h->ev, h->p2_b, SV, D, D, D, D, D, EPI_OUT_F32, ACT_NONE);
        }
        h->set_form(-1);          // (the vision stage's operand form, lr_set_precision_map, covers the projector)
        // ---- embeddings, positions (modeling_phi3_v.py:228-249, rw_model:344-345 | llava: masked_scatter, arange positions) ----
        const int Rl = B * S;
        launch_token_plan(input_ids, attention_mask, B, S, d_voff, h->img_row, h->pos_ids, h->tstat, st,
                          h->llava ? (long)d.image_token_id : -1L, h->llava ? 1 : 0);
        launch_embed(input_ids, h->img_row, h->wte, h->ev, h->x, Rl, D, d.vocab_size, st);
        launch_rope_table(h->pos_ids, h->tstat, B, S, h->inv_s, h->inv_l, d.rope_scaling, d.orig_max_pos, h->half, h->cs, st,
                          S + (d.rope_flash_convention ? 1 : 0));      // seq_len the rotary module is called with (lr_model_desc.rope_flash_convention)
        const int last_pos = (flags & LR_FWD_TRAINING_LAST_TOKEN) ? 1 : 0;
        const int gather = (d.mean_hidden_state || (flags & LR_FWD_KEEP_HIDDEN_STATES)) ? 0 : 1 + last_pos;
        h->last_pruned = run_decoder_stack(h, st, attention_mask, B, S, gather);
        if (d.mean_hidden_state) {      // rw_model:398-406: SkipCA + norm on every token, masked mean, value head
            run_mean_pool_head(h, st, attention_mask, B, S, voff, Vmax, nullptr, !(flags & LR_FWD_NO_FINAL_NORM), rewards_out);
            launch_slot_check(h->tstat, d_voff, rewards_out, B, d.value_head_dim, st);
            LR_HIP_CHECK(hipGetLastError());
            return;
        }
        // ---- tail: final norm of the gathered row, SkipCA, value head (rw_model:376-448) ----
        // (after a gathered last layer the B rows are already compact in xg: S = 1, position 0)
        launch_gather_norm_rows(h->last_pruned ? h->xg : h->x, h->tstat, h->last_pruned ? 1 : S, h->last_pruned ? 1 : last_pos,
                                (flags & LR_FWD_NO_FINAL_NORM) ? nullptr : h->norm_w, d.rms_eps, h->hL, B, D, st);
        const float* ao = nullptr;
        if (d.add_cross_attention) {
            launch_rowvec_linear(h->hL, h->Wq, h->tq, B, D, D, st);
            launch_rowvec_linear(h->tq, h->WkT, h->tkq, B, D, D, st);
            launch_ca_scores(h->ev, h->tkq, d_voff, B, Vmax, D, 1.0f / std::sqrt((float)D), h->tsc, st);
            launch_ca_softmax(h->tsc, B, Vmax, st);
            launch_ca_context(h->ev, h->tsc, d_voff, B, Vmax, D, h->tctx, st);
            launch_rowvec_linear(h->tctx, h->Wv, h->tao, B, D, D, st);
            ao = h->tao;
        }
        launch_reward_head(h->hL, ao, h->ca_w, d.ca_eps, h->vh, d.value_head_dim, rewards_out, B, D, st);
        launch_slot_check(h->tstat, d_voff, rewards_out, B, d.value_head_dim, st);
        LR_HIP_CHECK(hipGetLastError());
    });
}

int lr_last_hidden_state(lr_handle h, float* out_dev, size_t capacity, int no_final_norm, void* hip_stream) {
    if (!h || !out_dev) return LR_EINVAL;
    return guarded(h, [&] {
        if (!h->finalized || h->lastB <= 0) throw std::logic_error("lr_last_hidden_state: no forward has run on this handle");
        if (h->last_pruned) throw std::logic_error("lr_last_hidden_state: the last forward ran its final decoder layer for the reward rows only; "
                                                   "pass LR_FWD_KEEP_HIDDEN_STATES to the forward whose hidden states are wanted");
        const size_t rows = (size_t)h->lastB * h->lastS, D = (size_t)h->d.hidden;
        if (capacity < rows * D) throw std::invalid_argument("lr_last_hidden_state: buffer too small");
        hipStream_t st = (hipStream_t)hip_stream;
        if (no_final_norm) LR_HIP_CHECK(hipMemcpyAsync(out_dev, h->x, rows * D * 4, hipMemcpyDeviceToDevice, st));
        else launch_rms_rows_f32(h->x, h->norm_w, h->d.rms_eps, out_dev, (int)rows, (int)D, st);
        LR_HIP_CHECK(hipGetLastError());
    });
}

int lr_vision_embeds(lr_handle h, float* out_dev, size_t capacity, int* vmax, void* hip_stream) {
    if (!h || !vmax) return LR_EINVAL;
    return guarded(h, [&] {
        if (!h->finalized || h->lastB <= 0) throw std::logic_error("lr_vision_embeds: no forward has run on this handle");
        if (h->qwen) throw std::logic_error("lr_vision_embeds: the qwen branch keeps no padded vision rows (its SkipCA reads the embeddings, rw_model:359-371)");
        *vmax = h->lastVmax;
        if (!out_dev) return;
        const size_t D = (size_t)h->d.hidden, B = (size_t)h->lastB, V = (size_t)h->lastVmax;
        if (capacity < B * V * D) throw std::invalid_argument("lr_vision_embeds: buffer too small");
        hipStream_t st = (hipStream_t)hip_stream;
        LR_HIP_CHECK(hipMemsetAsync(out_dev, 0, B * V * D * 4, st));          // F.pad(..., "constant", 0), modeling_phi3_v.py:245
        for (size_t b = 0; b < B; ++b) {
            const size_t n = (size_t)(h->last_voff[b + 1] - h->last_voff[b]);
            if (n) LR_HIP_CHECK(hipMemcpyAsync(out_dev + b * V * D, h->ev + (size_t)h->last_voff[b] * D, n * D * 4, hipMemcpyDeviceToDevice, st));
        }
    });
}

int lr_read_tap(lr_handle h, const char* name, float* host_out, size_t capacity, size_t* n) {
    if (!h || !name || !host_out || !n) return LR_EINVAL;
    return guarded(h, [&] {
        const float* src = nullptr; size_t cnt = 0;
        const std::string nm = name;
        std::vector<float> conv;
        if (nm == "clip_x" && !h->qwen) { src = h->clip_x; cnt = (size_t)h->lastNC * h->T * h->d.clip_hidden; }
        else if (nm == "vit_x" && h->qwen) { src = h->vx; cnt = (size_t)h->lastP * h->vH; }
        else if (nm == "pos3" && h->qwen) {
            cnt = (size_t)3 * h->lastB * h->lastS;
            if (cnt > capacity) throw std::invalid_argument("lr_read_tap: buffer too small");
            std::vector<int> tmp(cnt);
            LR_HIP_CHECK(hipDeviceSynchronize());
            LR_HIP_CHECK(hipMemcpy(tmp.data(), h->pos3, cnt * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < cnt; ++i) host_out[i] = (float)tmp[i];
            *n = cnt;
            return;
        }
        else if (nm == "ev") { src = h->ev; cnt = (size_t)h->lastSV * h->d.hidden; }
        else if (nm == "x") {
            if (h->last_pruned) throw std::logic_error("lr_read_tap(x): the last forward ran its final decoder layer for the reward rows only; "
                                                       "pass LR_FWD_KEEP_HIDDEN_STATES to the forward whose residual stream is wanted");
            src = h->x; cnt = (size_t)h->lastB * h->lastS * h->d.hidden;
        }
        else if (nm == "hL") { src = h->hL; cnt = (size_t)h->lastB * h->d.hidden; }
        else throw std::invalid_argument("lr_read_tap: unknown tap");
        if (cnt > capacity) throw std::invalid_argument("lr_read_tap: buffer too small");
        LR_HIP_CHECK(hipDeviceSynchronize());
        LR_HIP_CHECK(hipMemcpy(host_out, src, cnt * 4, hipMemcpyDeviceToHost));
        *n = cnt;
    });
}

// ------------------------------------------------------------------------- single-kernel entries
static int op_guard(const std::function<void()>& f) {
    try { f(); LR_HIP_CHECK(hipGetLastError()); return LR_OK; }
    catch (const std::exception& ex) { g_create_error = ex.what(); return LR_EINVAL; }
}

int lr_op_gemm_bt(const void* A, const void* W, void* C, const float* bias, int M, int N, int K, int lda, int ldw, int ldc,
                  int epi, int act, int operand_dtype, int tile, void* hip_stream) {
    return op_guard([&] {
        GemmParams p{A, W, C, bias, M, N, K, lda, ldw, ldc, epi, act, nullptr, 0, 0};
        launch_gemm_bt(p, operand_dtype == LR_DT_F16 ? DT_F16 : DT_BF16, tile, (hipStream_t)hip_stream);
    });
}

int lr_op_gemm_bt_split(const void* A, const void* W, void* C, const float* bias, int M, int N, int K, int epi, int act,
                        int operand_dtype, int tile, void* hip_stream) {
    return op_guard([&] {
        const int nout = epi == EPI_SWIGLU_OP ? N / 2 : N;
        const bool op_out = epi == EPI_OUT_OP || epi == EPI_SWIGLU_OP;
        GemmParams p{A, W, C, bias, M, N, 2 * K, 2 * K, K, op_out ? 2 * nout : nout, epi, act, nullptr, 0, 0, K, op_out ? nout : 0};
        launch_gemm_bt(p, operand_dtype == LR_DT_F16 ? DT_F16 : DT_BF16, tile, (hipStream_t)hip_stream);
    });
}

int lr_op_gemm_bt_ext(const void* A, const void* W, const void* T, const void* Bm, const void* Blo, void* C, const float* bias, int M,
                      int N, int K, int k2, int split, int epi, int act, int operand_dtype, void* hip_stream) {
    return op_guard([&] {
        const int nout = epi == EPI_SWIGLU_OP ? N / 2 : N;
        const bool op_out = epi == EPI_OUT_OP || epi == EPI_SWIGLU_OP;
        GemmParams p{A, W, C, bias, M, N, K, K, K, nout, epi, act, nullptr, 0, 0};
        if (split) { p.K = 2 * K; p.lda = 2 * K; p.kw = K; if (op_out) { p.ldc = 2 * nout; p.split = nout; } }
        p.A2 = T; p.lda2 = (split ? 2 : 1) * k2; p.W2 = Bm; p.W2lo = Blo; p.ldw2 = k2; p.k2 = k2;
        launch_gemm_bt(p, operand_dtype == LR_DT_F16 ? DT_F16 : DT_BF16, 6, (hipStream_t)hip_stream);
    });
}

size_t lr_op_lo8_scratch_bytes(int M, int K) { return ((lo8_scale_bytes(M, K) + 255) & ~(size_t)255) + (((size_t)M * 4 + 255) & ~(size_t)255); }

int lr_op_gemm_bt_mixed(void* A, const void* W, void* W8, void* scratch, void* C, const float* bias, int M, int N, int K, int epi, int act,
                        int operand_dtype, int flags, int* wexp, void* hip_stream) {
    return op_guard([&] {
        const int dt = operand_dtype == LR_DT_F16 ? DT_F16 : DT_BF16;
        hipStream_t st = (hipStream_t)hip_stream;
        if (!wexp || !scratch || K % 128) throw std::runtime_error("lr_op_gemm_bt_mixed: wexp and scratch are required, K % 128 == 0");
        unsigned char* scales = (unsigned char*)scratch;                                                    // lr_op_lo8_scratch_bytes(M, K)
        int* aexp2 = (int*)(scales + ((lo8_scale_bytes(M, K) + 255) & ~(size_t)255));
        const bool inexact = (flags & 8) != 0;
        if (flags & 1) {                                   // prepare the e4m3 twin(s) of W (synchronous)
            unsigned* word = nullptr;
            LR_HIP_CHECK(hipMalloc((void**)&word, 4));
            if (inexact) {
                void* tmp = nullptr;
                LR_HIP_CHECK(hipMalloc(&tmp, (size_t)N * K * 2));
                prepare_weight_e4m3_pair(W, W8, K, K, N, tmp, dt, word, st, &wexp[0], &wexp[1]);
                LR_HIP_CHECK(hipFree(tmp));
            } else {
                wexp[0] = prepare_weight_e4m3(W, K, K, N, W8, dt, word, st);
            }
            LR_HIP_CHECK(hipFree(word));
        }
        if (flags & 2) launch_quantize_lo_inplace(A, 2 * K, K, M, scales, dt, st, inexact ? aexp2 : nullptr);   // re-encode A's residual half
        if (flags & 4) return;
        const int nout = epi == EPI_SWIGLU_OP ? N / 2 : N;
        const bool op_out = epi == EPI_OUT_OP || epi == EPI_SWIGLU_OP;
        GemmParams p{A, W, C, bias, M, N, inexact ? 2 * K : K + K / 2, 2 * K, K, op_out ? 2 * nout : nout, epi, act, nullptr, 0, 0, K, op_out ? nout : 0, W8};
        p.aexp = scales; p.wexp = wexp[0];
        if (inexact) { p.aexp2 = aexp2; p.wexp2 = wexp[1]; }
        if (flags & 32) {          // operand-typed output with one-byte residuals: C's block scales behind this call's own scratch
            if (!op_out || nout % 128) throw std::runtime_error("lr_op_gemm_bt_mixed: flag 32 needs an operand-typed output with columns % 128 == 0");
            p.oexp = scales + lr_op_lo8_scratch_bytes(M, K);
        }
        launch_gemm_bt8_mixed(p, dt, st, (flags & 16) ? 2 : (flags >> 8));          // flags & 16: no-epilogue diagnostic (tools/gemm_epi_probe.py)
    });
}

#ifdef LR_FP6_AB
// Round 6 A/B builds only (tools/fp6/build_ab.sh; not declared in include/llava_reward_hip.h, not in the product library): the
// split-operand GEMM with an FP6 residual pass.  A = [A_hi f16 x K | FP6 K-tiles, 96 bytes per 128 columns], W [N, K] f16,
// W6 = rows of ldw = K two-byte units holding W's FP6 K-tiles, ascales / wscales = per-(row, 32 elements) E8M0 bytes in the kernel's
// slice order ([K-tile][256-row tile] x 1 KB each).  flags & 16: the K loop alone (no epilogue, results not written).
int lr_op_gemm_bt_fp6ab(const void* A, const void* W, const void* W6, const void* ascales, const void* wscales, void* C, const float* bias,
                        int M, int N, int K, int epi, int act, int flags, void* hip_stream) {
    return op_guard([&] {
        const int nout = epi == EPI_SWIGLU_OP ? N / 2 : N;
        const bool op_out = epi == EPI_OUT_OP || epi == EPI_SWIGLU_OP;
        GemmParams p{A, W, C, bias, M, N, K + K / 2, 2 * K, K, op_out ? 2 * nout : nout, epi, act, nullptr, 0, 0, K, op_out ? nout : 0, W6};
        p.aexp = (const unsigned char*)ascales; p.wscale = (const float*)wscales; p.wexp = 127;
        // flags & 64: in-kernel stamps of the FP6 form, & 128: of the e4m3 form (W6 = the e4m3 twin, ascales = its block scales then), into `bias`
        launch_gemm_bt8_fp6ab(p, (hipStream_t)hip_stream, (flags & 16) ? 1 : (flags & 64) ? 2 : (flags & 128) ? 3 : 0);
    });
}
#endif

int lr_op_quantize_rows_fp8(const void* x, int rows, int K, int ldx, void* q, float* scale, int operand_dtype, void* hip_stream) {
    return op_guard([&] {
        launch_quantize_rows_fp8(x, ldx, K, rows, q, K, scale, operand_dtype == LR_DT_F16 ? DT_F16 : DT_BF16, (hipStream_t)hip_stream);
    });
}

int lr_op_gemm_fp8(const void* A8, const float* ascale, const void* W8, const float* wscale, void* C, const float* bias, int M, int N,
                   int K, int ldc, int epi, int act, int operand_dtype, void* hip_stream) {
    return op_guard([&] {
        GemmParams p{A8, W8, C, bias, M, N, K, K, K, ldc, epi, act, nullptr, 0, 0};
        p.ascale = ascale; p.wscale = wscale;
        launch_gemm_bt8_fp8(p, operand_dtype == LR_DT_F16 ? DT_F16 : DT_BF16, (hipStream_t)hip_stream);
    });
}

int lr_op_gemm_rope(const void* A, const void* W, void* C, const float* bias, const float* cs, int M, int N, int K, int rope_cols,
                    int rope_hd, int operand_dtype, int tile, void* hip_stream) {
    return op_guard([&] {
        GemmParams p{A, W, C, bias, M, N, K, K, K, N, EPI_ROPE_OP, ACT_NONE, cs, rope_cols, rope_hd};
        launch_gemm_bt(p, operand_dtype == LR_DT_F16 ? DT_F16 : DT_BF16, tile, (hipStream_t)hip_stream);
    });
}

int lr_op_attention(const void* Q, const void* K, const void* V, void* O, const int64_t* mask, const int* kmin, int ldq, int ldo,
                    int qoff, int koff, int voff, int batch, int S, int heads, int head_dim, int causal, int kv_group, float scale,
                    int operand_dtype, void* hip_stream) {
    return op_guard([&] {
        AttnParams p{Q, K, V, O, mask, kmin, 1, ldq, ldo, qoff, koff, voff, S, heads, scale, kv_group};
        launch_attention(p, batch, head_dim, causal != 0, operand_dtype == LR_DT_F16 ? DT_F16 : DT_BF16, (hipStream_t)hip_stream);
    });
}

int lr_op_attention_split(const void* Q, const void* K, const void* V, void* O, const int64_t* mask, const int* kmin, int ldq,
                          int ldo, int qoff, int koff, int voff, int lo_off, int o_split, int batch, int S, int heads, int head_dim,
                          int causal, int kv_group, float scale, int operand_dtype, void* hip_stream) {
    return op_guard([&] {
        AttnParams p{Q, K, V, O, mask, kmin, 1, ldq, ldo, qoff, koff, voff, S, heads, scale, kv_group, nullptr, 0, lo_off, o_split};
        p.lazy_t = ATT_LAZY_T_DEFAULT;
        launch_attention(p, batch, head_dim, causal != 0, operand_dtype == LR_DT_F16 ? DT_F16 : DT_BF16, (hipStream_t)hip_stream);
    });
}

int lr_op_attention_split_ex(const void* Q, const void* K, const void* V, void* O, const int64_t* mask, const int* kmin, int ldq,
                             int ldo, int qoff, int koff, int voff, int lo_off, int o_split, int batch, int S, int heads, int head_dim,
                             int causal, int kv_group, float scale, float lazy_threshold, int operand_dtype, void* hip_stream) {
    return op_guard([&] {
        AttnParams p{Q, K, V, O, mask, kmin, 1, ldq, ldo, qoff, koff, voff, S, heads, scale, kv_group, nullptr, 0, lo_off, o_split};
        p.lazy_t = lazy_threshold;
        launch_attention(p, batch, head_dim, causal != 0, operand_dtype == LR_DT_F16 ? DT_F16 : DT_BF16, (hipStream_t)hip_stream);
    });
}

int lr_op_attention_segments(const void* Q, const void* K, const void* V, void* O, const int32_t* cu, int n_seg, int ldq, int ldo,
                             int qoff, int koff, int voff, int heads, int head_dim, float scale, int operand_dtype, void* hip_stream) {
    return op_guard([&] {
        if (!cu || n_seg < 1) throw std::invalid_argument("attention_segments: need at least one segment");
        std::vector<int4> items;
        int max_len = 0;
        for (int i = 0; i < n_seg; ++i) {
            const int len = cu[i + 1] - cu[i];
            if (len < 0) throw std::invalid_argument("attention_segments: cu_seqlens must be non-decreasing");
            for (int q = 0; q * 128 < len; ++q) items.push_back(make_int4(cu[i], len, q, 0));
            max_len = std::max(max_len, len);
        }
        if (items.empty()) return;
        int4* d_items = nullptr;
        LR_HIP_CHECK(hipMalloc((void**)&d_items, items.size() * sizeof(int4)));
        hipStream_t st = (hipStream_t)hip_stream;
        LR_HIP_CHECK(hipMemcpyAsync(d_items, items.data(), items.size() * sizeof(int4), hipMemcpyHostToDevice, st));
        AttnParams p{Q, K, V, O, nullptr, nullptr, 0, ldq, ldo, qoff, koff, voff, max_len, heads, scale, 1, d_items, (int)items.size()};
        try { launch_attention(p, 1, head_dim, false, operand_dtype == LR_DT_F16 ? DT_F16 : DT_BF16, st); }
        catch (...) { hipStreamSynchronize(st); hipFree(d_items); throw; }
        LR_HIP_CHECK(hipStreamSynchronize(st));
        LR_HIP_CHECK(hipFree(d_items));
    });
}

int lr_op_norm_rows(const float* x, const float* w, const float* b, void* y, int rows, int H, float eps, int operand_dtype,
                    void* hip_stream) {
    return op_guard([&] { launch_norm_rows(x, w, b, y, rows, H, eps, operand_dtype == LR_DT_F16 ? DT_F16 : DT_BF16, (hipStream_t)hip_stream); });
}

int lr_op_synth_fill(float* out, size_t n, uint64_t seed, const char* name, float std_, float offset, int bf16_round,
                     void* hip_stream) {
    return op_guard([&] { launch_synth_fill(out, n, tensor_seed(seed, name), uniform_scale(std_), offset, bf16_round, (hipStream_t)hip_stream); });
}

}  // extern "C"
