// Qwen2.5-VL branch of the scoring path behind lr_forward_qwen (include/llava_reward_hip.h).
// Reference: rw_model_general_preference.py:354-371 (qwen branch of custom_forward), :387-397 (its SkipCA), :407-448
// (value head + EOS gather).  Backbone (third party, transformers modeling_qwen2_5_vl.py): ViT with window / full
// attention over cu_seqlens, 2-D rotary, RMSNorm, biased SwiGLU; 2x2 patch merger; decoder with q/k/v bias, GQA and
// multimodal RoPE; get_rope_index for the 3-D positions.
//
// Layout decisions (MI355X): the ViT runs in WINDOW ORDER end to end (the permutation is folded into the patch gather
// and into the image-slot -> merger-row table, so no activation is ever re-ordered); the 80-wide ViT heads are stored
// 96 wide (zero rows in the packed qkv weight / zero columns in the projection weight) so that the MFMA attention
// kernel and the fused-RoPE GEMM epilogue of the decoder serve the tower unchanged; the MLP width 3420 is stored 3456
// wide; window and full attention are one ragged launch each (a table of (row0, length, tile) items built on the host).
#include "engine.h"

namespace {

constexpr int LR_QWEN_IMAGES_PER_ROW = 4;      // capacity of the per-call image table: 4 images per row on average

struct QwenTables {       // byte offsets inside one slot of the per-forward table ring
    size_t src, hw, slot2row, items_win, items_full, imgs, total;
};

QwenTables qwen_table_layout(const lr_model_desc& d) {
    const size_t P = d.max_patches, M = P / (size_t)(d.vit_merge * d.vit_merge), NI = (size_t)LR_QWEN_IMAGES_PER_ROW * d.max_batch;
    QwenTables t;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    t.src = take(P * 4);
    t.hw = take(P * 8);
    t.slot2row = take(M * 4);
    t.items_win = take(M * 16);
    t.items_full = take((P / 128 + NI + 1) * 16);
    t.imgs = take(NI * 16);
    t.total = o;
    return t;
}

}  // namespace

void validate_desc_qwen(const lr_model_desc& d) {
    auto bad = [](const char* m) { throw std::invalid_argument(m); };
    if (d.head_dim != 128) bad("Qwen2.5-VL decoder head_dim must be 128");
    if (d.kv_heads < 1 || d.heads % d.kv_heads) bad("heads must be a multiple of kv_heads");
    if (d.heads * d.head_dim != d.hidden) bad("Qwen2.5-VL: heads * head_dim must equal hidden (Qwen2_5_VLAttention)");
    if (d.image_token_id < 0 || d.image_token_id >= d.vocab_size) bad("image_token_id out of range");
    if (d.mrope_section[0] < 0 || d.mrope_section[1] < 0 || d.mrope_section[2] < 0 ||
        d.mrope_section[0] + d.mrope_section[1] + d.mrope_section[2] != d.head_dim / 2)
        bad("mrope_section must sum to head_dim / 2");
    if (d.vit_depth < 0 || d.vit_hidden <= 0 || d.vit_heads <= 0 || d.vit_hidden % d.vit_heads) bad("bad ViT geometry");
    const int vhd = d.vit_hidden / d.vit_heads;
    if (vhd % 4 || vhd > 96 || vhd < 8) bad("ViT head_dim must be a multiple of 4, at most 96");
    if (d.vit_hidden % 64) bad("vit_hidden must be a multiple of 64");
    if (d.vit_hidden * d.vit_merge * d.vit_merge > 8192) bad("vit_hidden * merge^2 above 8192 is not supported");
    if (d.vit_intermediate <= 0 || d.vit_patch <= 0 || d.vit_temporal_patch <= 0 || d.vit_in_ch <= 0) bad("bad ViT patch geometry");
    if (d.vit_merge < 1 || d.vit_window < d.vit_merge * d.vit_patch) bad("vit_window must cover at least one merged patch");
    if (d.vit_n_fullatt < 0 || d.vit_n_fullatt > LR_MAX_FULLATT) bad("vit_n_fullatt out of range");
    if (d.max_patches < d.vit_merge * d.vit_merge || d.max_patches % (d.vit_merge * d.vit_merge)) bad("max_patches must be a positive multiple of merge^2");
    if (d.ca_token_id < 0) bad("ca_token_id must be non-negative");
}

// Qwen2.5-VL-*-Instruct checkpoint names (transformers 4.50 layout: visual.*, model.*) + reward heads (rw_model:314-326)
void build_weight_table_qwen(lr_engine* e) {
    const lr_model_desc& d = e->d;
    const int D = d.hidden, I = d.intermediate, hd = e->hd, Hq = e->Hq, Hkv = e->Hkv;
    const int od = e->op_dt;
    e->vH = d.vit_hidden;
    e->vhd = d.vit_hidden / d.vit_heads;
    e->vhdp = e->vhd > 64 ? 96 : 64;
    e->vHp = d.vit_heads * e->vhdp;
    e->vI = d.vit_intermediate;
    e->vIp = (e->vI + 63) / 64 * 64;
    e->vK = d.vit_in_ch * d.vit_temporal_patch * d.vit_patch * d.vit_patch;
    e->vKpad = (e->vK + 63) / 64 * 64;
    e->vunit = d.vit_merge * d.vit_merge;
    e->vHm = e->vH * e->vunit;
    const int vH = e->vH, vhd = e->vhd, vhdp = e->vhdp, vHp = e->vHp, vI = e->vI, vIp = e->vIp, vHm = e->vHm;

    e->wte = (unsigned short*)e->dalloc((size_t)d.vocab_size * D * 2, true);      // bf16 embedding table (not an MFMA operand)
    add_slot(e, "model.embed_tokens.weight", {d.vocab_size, D}, e->wte, D, D, DT_BF16, PACK_PLAIN, 0.02, 0);
    e->vpatch_w = oalloc(e, (size_t)vH * e->vKpad);
    add_slot(e, "visual.patch_embed.proj.weight", {vH, d.vit_in_ch, d.vit_temporal_patch, d.vit_patch, d.vit_patch}, e->vpatch_w,
             e->vKpad, e->vKpad, od, PACK_PLAIN, 0.02, 0);
    e->vl.resize(d.vit_depth);
    for (int l = 0; l < d.vit_depth; ++l) {
        VitLayer& L = e->vl[l];
        const std::string p = "visual.blocks." + std::to_string(l) + ".";
        L.n1 = falloc(e, vH); L.n2 = falloc(e, vH);
        L.qkv_w = oalloc(e, (size_t)3 * vHp * vH); L.qkv_b = falloc(e, 3 * vHp);
        L.proj_w = oalloc(e, (size_t)vH * vHp); L.proj_b = falloc(e, vH);
        L.gu_w = oalloc(e, (size_t)2 * vIp * vH); L.gu_b = falloc(e, 2 * vIp);
        L.down_w = oalloc(e, (size_t)vH * vIp); L.down_b = falloc(e, vH);
        vec_slot(e, p + "norm1.weight", {vH}, L.n1, 0.05, 1.0);
        // fused qkv [3 * vH, vH]: heads widened to vhdp, q and k dims pair-interleaved for the RoPE epilogue; bias alike
        add_slot(e, p + "attn.qkv.weight", {3 * vH, vH}, L.qkv_w, vH, vH, od, PACK_ROPE_QKV, 0.02, 0);
        e->slots.back().aux_d = vH; e->slots.back().aux_hd = vhd; e->slots.back().aux_hdp = vhdp;
        add_slot(e, p + "attn.qkv.bias", {3 * vH}, L.qkv_b, 1, 1, DT_F32, PACK_ROPE_QKV, 0.02, 0);
        e->slots.back().rows = 3 * vH; e->slots.back().cols = 1;
        e->slots.back().aux_d = vH; e->slots.back().aux_hd = vhd; e->slots.back().aux_hdp = vhdp;
        add_slot(e, p + "attn.proj.weight", {vH, vH}, L.proj_w, vHp, vH, od, PACK_HEADPAD_COLS, 0.02, 0);
        e->slots.back().aux_hd = vhd; e->slots.back().aux_hdp = vhdp;
        vec_slot(e, p + "attn.proj.bias", {vH}, L.proj_b, 0.02, 0);
        vec_slot(e, p + "norm2.weight", {vH}, L.n2, 0.05, 1.0);
        add_slot(e, p + "mlp.gate_proj.weight", {vI, vH}, L.gu_w, vH, vH, od, PACK_SWIGLU_GATE, 0.02, 0);
        add_slot(e, p + "mlp.gate_proj.bias", {vI}, L.gu_b, 1, 1, DT_F32, PACK_SWIGLU_GATE, 0.02, 0);
        e->slots.back().rows = vI; e->slots.back().cols = 1;
        add_slot(e, p + "mlp.up_proj.weight", {vI, vH}, L.gu_w, vH, vH, od, PACK_SWIGLU_UP, 0.02, 0);
        add_slot(e, p + "mlp.up_proj.bias", {vI}, L.gu_b, 1, 1, DT_F32, PACK_SWIGLU_UP, 0.02, 0);
        e->slots.back().rows = vI; e->slots.back().cols = 1;
        add_slot(e, p + "mlp.down_proj.weight", {vH, vI}, L.down_w, vIp, vIp, od, PACK_PLAIN, 0.02, 0);
        vec_slot(e, p + "mlp.down_proj.bias", {vH}, L.down_b, 0.02, 0);
    }
    e->vlnq = falloc(e, vH);
    vec_slot(e, "visual.merger.ln_q.weight", {vH}, e->vlnq, 0.05, 1.0);
    e->m0_w = oalloc(e, (size_t)vHm * vHm); e->m0_b = falloc(e, vHm);
    e->m2_w = oalloc(e, (size_t)D * vHm); e->m2_b = falloc(e, D);
    add_slot(e, "visual.merger.mlp.0.weight", {vHm, vHm}, e->m0_w, vHm, vHm, od, PACK_PLAIN, 0.02, 0);
    vec_slot(e, "visual.merger.mlp.0.bias", {vHm}, e->m0_b, 0.02, 0);
    add_slot(e, "visual.merger.mlp.2.weight", {D, vHm}, e->m2_w, vHm, vHm, od, PACK_PLAIN, 0.02, 0);
    vec_slot(e, "visual.merger.mlp.2.bias", {D}, e->m2_b, 0.02, 0);

    e->dl.resize(d.layers);
    for (int l = 0; l < d.layers; ++l) {
        DecLayer& L = e->dl[l];
        const std::string p = "model.layers." + std::to_string(l) + ".";
        L.ln1 = falloc(e, D); L.ln2 = falloc(e, D);
        L.qkv_w = oalloc(e, (size_t)e->Nqkv * D); L.qkv_b = falloc(e, e->Nqkv); L.o_w = oalloc(e, (size_t)D * Hq);
        L.gu_w = oalloc(e, (size_t)2 * I * D); L.down_w = oalloc(e, (size_t)D * I);
        vec_slot(e, p + "input_layernorm.weight", {D}, L.ln1, 0.05, 1.0);
        const char* nm[3] = {"q_proj", "k_proj", "v_proj"};
        const int n[3] = {Hq, Hkv, Hkv}, off[3] = {0, Hq, Hq + Hkv};
        for (int i = 0; i < 3; ++i) {
            const int mode = i < 2 ? PACK_ROPE_QKV : PACK_PLAIN;
            add_slot(e, p + "self_attn." + nm[i] + ".weight", {n[i], D}, (char*)L.qkv_w + (size_t)off[i] * D * 2, D, D, od, mode, 0.02, 0);
            e->slots.back().aux_d = n[i]; e->slots.back().aux_hd = hd;
            add_slot(e, p + "self_attn." + nm[i] + ".bias", {n[i]}, L.qkv_b + off[i], 1, 1, DT_F32, mode, 0.02, 0);
            e->slots.back().rows = n[i]; e->slots.back().cols = 1;
            e->slots.back().aux_d = n[i]; e->slots.back().aux_hd = hd;
        }
        add_slot(e, p + "self_attn.o_proj.weight", {D, Hq}, L.o_w, Hq, Hq, od, PACK_PLAIN, 0.02, 0);
        vec_slot(e, p + "post_attention_layernorm.weight", {D}, L.ln2, 0.05, 1.0);
        add_slot(e, p + "mlp.gate_proj.weight", {I, D}, L.gu_w, D, D, od, PACK_SWIGLU_GATE, 0.02, 0);
        add_slot(e, p + "mlp.up_proj.weight", {I, D}, L.gu_w, D, D, od, PACK_SWIGLU_UP, 0.02, 0);
        add_slot(e, p + "mlp.down_proj.weight", {D, I}, L.down_w, I, I, od, PACK_PLAIN, 0.02, 0);
        if (d.lora_rank > 0) {      // utils/utils.py:223-242 create_lora_config_qwen: q, k, v, o, gate, up, down of every decoder layer
            for (int i = 0; i < 3; ++i)
                register_lora(e, L.lqkv, p + "self_attn." + nm[i], e->Nqkv, D, i, 3, n[i], off[i], i < 2 ? PACK_ROPE_QKV : PACK_PLAIN, n[i], hd);
            register_lora(e, L.lo, p + "self_attn.o_proj", D, Hq, 0, 1, D, 0, PACK_PLAIN);
            register_lora(e, L.lgu, p + "mlp.gate_proj", 2 * I, D, 0, 2, I, 0, PACK_SWIGLU_GATE);
            register_lora(e, L.lgu, p + "mlp.up_proj", 2 * I, D, 1, 2, I, 0, PACK_SWIGLU_UP);
            register_lora(e, L.ldown, p + "mlp.down_proj", D, I, 0, 1, D, 0, PACK_PLAIN);
        }
    }
    e->norm_w = falloc(e, D);
    vec_slot(e, "model.norm.weight", {D}, e->norm_w, 0.05, 1.0);
    if (d.add_cross_attention) {
        // W_q and W_k are accepted (they are in the checkpoint, reward_adaptor_loader.py:84-91) but cannot influence the
        // reward: all un-masked K rows are identical, so the softmax is uniform whatever the scores are (rw_model:387-395).
        e->Wq = falloc(e, (size_t)D * D);
        e->WkT = falloc(e, (size_t)D * D);
        e->Wv = falloc(e, (size_t)D * D);
        e->ca_w = falloc(e, D);
        e->ca_u = falloc(e, D);
        add_slot(e, "W_q.weight", {D, D}, e->Wq, D, D, DT_F32, PACK_PLAIN, 0.02, 0);
        add_slot(e, "W_k.weight", {D, D}, e->WkT, D, D, DT_F32, PACK_PLAIN, 0.02, 0);
        add_slot(e, "W_v.weight", {D, D}, e->Wv, D, D, DT_F32, PACK_PLAIN, 0.02, 0);
        vec_slot(e, "ca_layernorm.weight", {D}, e->ca_w, 0.05, 1.0);
    }
    e->vh = falloc(e, (size_t)d.value_head_dim * D);
    add_slot(e, "value_head.weight", {d.value_head_dim, D}, e->vh, D, D, DT_F32, PACK_PLAIN, 1.0 / std::sqrt((double)D), 0);
    upload_rope_tables(e);
    // Qwen2_5_VisionRotaryEmbedding(head_dim / 2): inv_freq[j] = theta^(-2j / (head_dim / 2)), j < head_dim / 4
    std::vector<float> inv(vhd / 4);
    for (int j = 0; j < vhd / 4; ++j) inv[j] = (float)(1.0 / std::pow((double)d.vit_rope_theta, (double)(2 * j) / (double)(vhd / 2)));
    e->vinv = falloc(e, inv.size());
    LR_HIP_CHECK(hipMemcpy(e->vinv, inv.data(), inv.size() * 4, hipMemcpyHostToDevice));
}

void finalize_qwen(lr_engine* h) {
    const lr_model_desc& d = h->d;
    const size_t B = d.max_batch, S = d.max_seq, P = d.max_patches, D = d.hidden, I = d.intermediate;
    const size_t PAD = 256, Rv = P + PAD, Rm = P / h->vunit + PAD, Rl = B * S + PAD;
    auto W = [&](size_t bytes) { return h->dalloc(bytes, false); };
    const size_t ob = 2 * (size_t)(1 + h->prec);      // bytes per operand element (hi [+ lo])
    h->vA = W(Rv * h->vKpad * ob); h->vx = (float*)W(Rv * h->vH * 4); h->vhn = W(Rv * h->vH * ob);
    h->vqkv = W(Rv * 3 * h->vHp * ob); h->vatt = W(Rv * h->vHp * ob);
    h->vff = W(Rv * h->vIp * ob); h->vcs = (float*)W(Rv * h->vhdp * 4); h->vm1 = W(Rm * h->vHm * ob);
    h->ev = (float*)W(Rm * D * 4);
    h->x = (float*)W(Rl * D * 4); h->h = W(Rl * D * ob); h->qkv = W(Rl * h->Nqkv * ob);
    h->att = W(Rl * h->Hq * ob); h->ff = W(Rl * I * ob); h->cs = (float*)W(Rl * h->hd * 4);
    if (h->lora_k2max > 0) h->lt = W(Rl * (size_t)h->lora_k2max * ob);
    h->pos_ids = (int*)W(Rl * 4); h->img_row = (int*)W(Rl * 4); h->pos3 = (int*)W(3 * Rl * 4);
    h->tstat = (int*)W(B * 16); h->rstat = (int*)W(B * 16);
    h->hL = (float*)W(B * D * 4); h->tao = (float*)W(B * D * 4);
    if (d.mean_hidden_state) alloc_mean_pool(h, Rl, 0);
    alloc_gather_ws(h);
    h->tab_bytes = qwen_table_layout(d).total;
    LR_HIP_CHECK(hipHostMalloc((void**)&h->tab_host, h->tab_bytes * lr_engine::NSLOT));
    h->tab_dev = (char*)W(h->tab_bytes * lr_engine::NSLOT);
    for (int i = 0; i < lr_engine::NSLOT; ++i) LR_HIP_CHECK(hipEventCreateWithFlags(&h->tab_ev[i], hipEventDisableTiming));
    {   // e4m3 twins of the GEMM weights (default parity mode / W8A8 mode)
        std::vector<GemmWeight> ws;
        const int vH = h->vH, vHp = h->vHp, vIp = h->vIp, vHm = h->vHm;
        ws.push_back({h->vpatch_w, vH, h->vKpad, Rv});
        for (const VitLayer& L : h->vl) {
            ws.push_back({L.qkv_w, 3 * vHp, vH, Rv}); ws.push_back({L.proj_w, vH, vHp, Rv});
            ws.push_back({L.gu_w, 2 * vIp, vH, Rv}); ws.push_back({L.down_w, vH, vIp, Rv});
        }
        ws.push_back({h->m0_w, vHm, vHm, Rm}); ws.push_back({h->m2_w, (int)D, vHm, Rm});
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
    LR_HIP_CHECK(hipMemset(h->tstat, 0, B * 16));
    LR_HIP_CHECK(hipMemset(h->rstat, 0, B * 16));
    LR_HIP_CHECK(hipMemset(h->pos_ids, 0, Rl * 4));
    if (d.add_cross_attention) {
        // u = W_v wte[ca_token] (fp32 accumulation over the stored bf16 weights), 0 when the token is not in the vocabulary
        LR_HIP_CHECK(hipMemset(h->ca_u, 0, D * 4));
        if (d.ca_token_id < d.vocab_size) {
            int64_t* id = (int64_t*)h->pos3;     // scratch
            const int64_t tok = d.ca_token_id;
            LR_HIP_CHECK(hipMemcpy(id, &tok, 8, hipMemcpyHostToDevice));
            int* ir = h->img_row;
            const int m1 = -1;
            LR_HIP_CHECK(hipMemcpy(ir, &m1, 4, hipMemcpyHostToDevice));
            launch_embed(id, ir, h->wte, nullptr, h->x, 1, (int)D, d.vocab_size, 0);
            launch_rowvec_linear(h->x, h->Wv, h->ca_u, 1, (int)D, (int)D, 0);
            LR_HIP_CHECK(hipStreamSynchronize(0));
        }
    }
}

extern "C" int lr_forward_qwen(lr_handle h, const int64_t* input_ids, const int64_t* attention_mask, const void* pixel_values,
                               int pix_dtype, const int64_t* grid_thw, int n_images, int B, int S, int flags, float* rewards_out,
                               void* hip_stream) {
    if (!h) return LR_EINVAL;
    return guarded(h, [&] {
        if (!h->finalized) throw std::logic_error("lr_forward_qwen: call lr_finalize first");
        if (!h->qwen) throw std::logic_error("lr_forward_qwen: the handle was not created for LR_BACKBONE_QWEN2_5_VL");
        if (!input_ids || !attention_mask || !pixel_values || !grid_thw || !rewards_out)
            throw std::invalid_argument("lr_forward_qwen: null argument (inputs_batch needs input_ids, attention_mask, pixel_values, image_grid_thw)");
        const lr_model_desc& d = h->d;
        if (B < 1 || B > d.max_batch) throw std::invalid_argument("lr_forward_qwen: batch exceeds max_batch");
        if (S < 1 || S > d.max_seq) throw std::invalid_argument("lr_forward_qwen: sequence exceeds max_seq");
        if (n_images < 1 || n_images > LR_QWEN_IMAGES_PER_ROW * d.max_batch)
            throw std::invalid_argument("lr_forward_qwen: n_images must be in [1, 4 * max_batch]");
        if (pix_dtype != LR_DT_F32 && pix_dtype != LR_DT_BF16) throw std::invalid_argument("lr_forward_qwen: pixel dtype must be F32 or BF16");
        hipStream_t st = (hipStream_t)hip_stream;
        const int D = d.hidden, m = d.vit_merge, unit = h->vunit;
        const int vH = h->vH, vHp = h->vHp, vhdp = h->vhdp, vIp = h->vIp, vHm = h->vHm;

        // ---- host plan (transformers.vision_utils get_vision_window_index / get_vision_position_ids / cu_seqlens) ----
        const int slot = h->slot_i; h->slot_i = (h->slot_i + 1) % lr_engine::NSLOT;
        if (h->tab_used[slot]) LR_HIP_CHECK(hipEventSynchronize(h->tab_ev[slot]));
        char* th = h->tab_host + (size_t)slot * h->tab_bytes;
        char* td = h->tab_dev + (size_t)slot * h->tab_bytes;
        const QwenTables T = qwen_table_layout(d);
        int* src = (int*)(th + T.src);
        int2* hw = (int2*)(th + T.hw);
        int* slot2row = (int*)(th + T.slot2row);
        int4* items_win = (int4*)(th + T.items_win);
        int4* items_full = (int4*)(th + T.items_full);
        int4* imgs = (int4*)(th + T.imgs);
        const int ws = d.vit_window / m / d.vit_patch;           // window side in merged tokens
        long P = 0;
        for (int i = 0; i < n_images; ++i) {
            const int64_t t = grid_thw[3 * i], gh = grid_thw[3 * i + 1], gw = grid_thw[3 * i + 2];
            if (t != 1) throw std::invalid_argument("lr_forward_qwen: image_grid_thw[:, 0] must be 1 (video grids are out of scope)");
            if (gh < m || gw < m || gh % m || gw % m) throw std::invalid_argument("lr_forward_qwen: grid h, w must be positive multiples of the merge size");
            if (gh * gw > 8192) throw std::invalid_argument("lr_forward_qwen: more than 8192 patches per image are not supported");
            P += gh * gw;
        }
        if (P > d.max_patches) throw std::invalid_argument("lr_forward_qwen: pixel_values has more patches than max_patches");
        const int N = (int)P, M = N / unit;
        int nwin = 0, nfull = 0, row = 0, mbase = 0, max_seg = 0;
        for (int i = 0; i < n_images; ++i) {
            const int gh = (int)grid_thw[3 * i + 1], gw = (int)grid_thw[3 * i + 2];
            const int lh = gh / m, lw = gw / m;
            imgs[i] = make_int4(lh, lw, std::max(gh, gw) / m, mbase);
            const int row_img = row;
            // windows in row-major order of the (padded) window grid; a full extra window when the grid divides evenly
            // stays empty, as in the reference
            const int nh = (lh + (ws - lh % ws)) / ws, nw = (lw + (ws - lw % ws)) / ws;
            for (int wy = 0; wy < nh; ++wy)
                for (int wx = 0; wx < nw; ++wx) {
                    const int row0 = row;
                    for (int y = wy * ws; y < std::min(lh, (wy + 1) * ws); ++y)
                        for (int x = wx * ws; x < std::min(lw, (wx + 1) * ws); ++x) {
                            const int u = y * lw + x;                       // merge unit (processor order) of this image
                            slot2row[mbase + u] = row / unit;
                            for (int k = 0; k < unit; ++k) {
                                // processor patch order inside the image: [lh, lw, m, m]
                                src[row] = (mbase + u) * unit + k;
                                hw[row] = make_int2(y * m + k / m, x * m + k % m);
                                ++row;
                            }
                        }
                    const int len = row - row0;
                    for (int q = 0; q * 128 < len; ++q) items_win[nwin++] = make_int4(row0, len, q, 0);
                    max_seg = std::max(max_seg, len);
                }
            const int len = row - row_img;
            for (int q = 0; q * 128 < len; ++q) items_full[nfull++] = make_int4(row_img, len, q, 0);
            max_seg = std::max(max_seg, len);
            mbase += lh * lw;
        }
        LR_HIP_CHECK(hipMemcpyAsync(td, th, h->tab_bytes, hipMemcpyHostToDevice, st));
        LR_HIP_CHECK(hipEventRecord(h->tab_ev[slot], st));
        h->tab_used[slot] = true;
        const int* d_src = (const int*)(td + T.src);
        const int2* d_hw = (const int2*)(td + T.hw);
        const int* d_slot2row = (const int*)(td + T.slot2row);
        const int4* d_items_win = (const int4*)(td + T.items_win);
        const int4* d_items_full = (const int4*)(td + T.items_full);
        const int4* d_imgs = (const int4*)(td + T.imgs);
        h->lastB = B; h->lastS = S; h->lastP = N; h->lastSV = M;

        // ---- ViT (Qwen2_5_VisionTransformerPretrainedModel.forward), window order throughout ----
        h->set_form(h->pm_clip);          // lr_set_precision_map: the vision tower's operand form
        launch_qwen_patch_gather(pixel_values, pix_dtype == LR_DT_F32 ? DT_F32 : DT_BF16, d_src, N, h->vK, h->vKpad, h->vA, h->op_dt, st, h->prec);
        gemm(h, st, h->vA, h->vpatch_w, h->vx, nullptr, N, vH, h->vKpad, h->vKpad, h->vKpad, vH, EPI_OUT_F32, ACT_NONE);
        launch_vit_rope_table(d_hw, N, h->vinv, h->vhd / 4, vhdp / 2, h->vcs, st);
        const int nvl = h->lim_clip >= 0 && h->lim_clip < d.vit_depth ? h->lim_clip : d.vit_depth;
        const float vscale = 1.0f / std::sqrt((float)h->vhd);
        for (int l = 0; l < nvl; ++l) {
            const VitLayer& L = h->vl[l];
            bool full = false;
            for (int i = 0; i < d.vit_n_fullatt; ++i) full = full || d.vit_fullatt[i] == l;
            launch_norm_rows(h->vx, L.n1, nullptr, h->vhn, N, vH, d.vit_eps, h->op_dt, st, h->prec);
            {
                GemmParams gp{h->vhn, L.qkv_w, h->vqkv, L.qkv_b, N, 3 * vHp, vH, vH, vH, 3 * vHp, EPI_ROPE_OP, ACT_NONE, h->vcs, 2 * vHp, vhdp};
                apply_prec_base(h, gp);
                if ((2 * vHp) % 256 == 0 && w8a8_eligible(h, gp)) {
                    launch_w8a8(h, gp, st);
                } else if ((2 * vHp) % 256 == 0 && (lo8_eligible(h, gp) || gemm_bt_is_deep(gp, h->gemm_tile))) {
                    upgrade_lo8(h, gp, st);
                    launch_gemm_bt(gp, h->op_dt, h->gemm_tile, st);
                } else {
                    if (!h->vqkv32) h->vqkv32 = (float*)h->dalloc(((size_t)d.max_patches + 256) * 3 * vHp * 4, false);      // fallback only
                    gemm(h, st, h->vhn, L.qkv_w, h->vqkv32, L.qkv_b, N, 3 * vHp, vH, vH, vH, 3 * vHp, EPI_OUT_F32, ACT_NONE);
                    launch_rope_split(h->vqkv32, h->vcs, h->vqkv, N, 2 * vHp, vHp, vhdp, h->op_dt, st, h->prec);
                }
            }
            AttnParams ap{h->vqkv, h->vqkv, h->vqkv, h->vatt, nullptr, nullptr, 0, 3 * vHp, vHp, 0, vHp, 2 * vHp, max_seg, d.vit_heads,
                          vscale, 1, full ? d_items_full : d_items_win, full ? nfull : nwin};
            apply_prec(h, ap);
            launch_attention(ap, 1, vhdp, false, h->op_dt, st);
            gemm(h, st, h->vatt, L.proj_w, h->vx, L.proj_b, N, vH, vHp, vHp, vHp, vH, EPI_RESADD_F32, ACT_NONE);
            launch_norm_rows(h->vx, L.n2, nullptr, h->vhn, N, vH, d.vit_eps, h->op_dt, st, h->prec);
            gemm(h, st, h->vhn, L.gu_w, h->vff, L.gu_b, N, 2 * vIp, vH, vH, vH, vIp, EPI_SWIGLU_OP, ACT_NONE, nullptr, L.down_w, vH);
            gemm(h, st, h->vff, L.down_w, h->vx, L.down_b, N, vH, vIp, vIp, vIp, vH, EPI_RESADD_F32, ACT_NONE);
        }
        // ---- merger (Qwen2_5_VLPatchMerger): RMSNorm, 4 consecutive rows = one LLM token, Linear-GELU-Linear ----
        launch_norm_rows(h->vx, h->vlnq, nullptr, h->vhn, N, vH, 1e-6f, h->op_dt, st, h->prec, unit);     // rows of one merged token side by side
        gemm(h, st, h->vhn, h->m0_w, h->vm1, h->m0_b, M, vHm, vHm, vHm, vHm, vHm, EPI_OUT_OP, ACT_GELU_ERF, nullptr, h->m2_w, D);
        gemm(h, st, h->vm1, h->m2_w, h->ev, h->m2_b, M, D, vHm, vHm, vHm, D, EPI_OUT_F32, ACT_NONE);

        h->set_form(-1);
        // ---- embeddings + 3-D positions (Qwen2_5_VLModel.forward: masked_scatter, get_rope_index) ----
        const int Rl = B * S;
        // token_plan fills tstat (last / first valid index); its slot ranks are superseded by qwen_plan, so the offsets are don't-cares
        launch_token_plan(input_ids, attention_mask, B, S, h->rstat, h->img_row, h->pos_ids, h->tstat, st,
                          (long)d.image_token_id, 1);
        launch_qwen_plan(input_ids, attention_mask, B, S, (long)d.image_token_id, (long)d.ca_token_id, d_imgs, n_images, d_slot2row, M,
                         h->rstat, h->pos3, h->img_row, st);
        launch_embed(input_ids, h->img_row, h->wte, h->ev, h->x, Rl, D, d.vocab_size, st);
        launch_mrope_table(h->pos3, Rl, h->inv_s, d.mrope_section[0], d.mrope_section[1], h->half, h->cs, st);
        const int last_pos = (flags & LR_FWD_TRAINING_LAST_TOKEN) ? 1 : 0;
        h->last_pruned = run_decoder_stack(h, st, attention_mask, B, S, (d.mean_hidden_state || (flags & LR_FWD_KEEP_HIDDEN_STATES)) ? 0 : 1 + last_pos);
        if (d.mean_hidden_state) {      // rw_model:398-406; the as-written SkipCA adds the same vector to every token of a row
            if (d.add_cross_attention) launch_qwen_ca_vec(h->rstat, h->ca_u, B, D, h->tao, st);
            run_mean_pool_head(h, st, attention_mask, B, S, nullptr, 0, h->tao, true, rewards_out);
            LR_HIP_CHECK(hipGetLastError());
            return;
        }
        // ---- tail: final norm of the gathered row (hidden_states[-1]), as-written SkipCA, value head (rw_model:387-448) ----
        launch_gather_norm_rows(h->last_pruned ? h->xg : h->x, h->tstat, h->last_pruned ? 1 : S, h->last_pruned ? 1 : last_pos, h->norm_w, d.rms_eps,
                                h->hL, B, D, st);
        const float* ao = nullptr;
        if (d.add_cross_attention) {
            launch_qwen_ca_vec(h->rstat, h->ca_u, B, D, h->tao, st);
            ao = h->tao;
        }
        launch_reward_head(h->hL, ao, h->ca_w, d.ca_eps, h->vh, d.value_head_dim, rewards_out, B, D, st);
        LR_HIP_CHECK(hipGetLastError());
    });
}
