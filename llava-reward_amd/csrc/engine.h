// Engine internals shared by engine.hip (Phi-3.5-V, LLaVA-1.6) and qwen.hip (Qwen2.5-VL): the weight-slot table,
// the handle and small host helpers.  Not part of the ABI (include/llava_reward_hip.h is).
#pragma once
#include "../../include/llava_reward_hip.h"
#include "common.h"
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

using namespace lr;

inline thread_local std::string g_create_error;

inline uint64_t fnv1a64(const char* s) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (; *s; ++s) { h ^= (unsigned char)*s; h *= 0x100000001B3ull; }
    return h;
}
inline uint64_t splitmix64_host(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t tensor_seed(uint64_t seed, const char* name) { return splitmix64_host(seed ^ fnv1a64(name)); }
inline float uniform_scale(double std) { return (float)(std * std::sqrt(12.0) / 16777216.0); }

struct Slot {
    std::string name;
    std::vector<int64_t> shape;
    int rows, cols;            // 2-D view: [shape[0], prod(rest)] (1-D: [1, n])
    void* dst;                 // destination (already offset for concatenated tensors)
    int ld_dst, cols_dst, dst_dtype, mode;
    double std_, offset;       // synthetic init (llava_reward_amd.synth.weight_specs)
    int aux_d = 0, aux_hd = 0, aux_hdp = 0; // PACK_ROPE_QKV: rows of a section, head width, stored head width
    void* lo_dst = nullptr;    // split-operand mode: where the rounding residual of this tensor goes (same layout as dst)
    int wid = -1;              // index of the operand-weight buffer that holds dst
    bool provided = false;
};

struct ClipLayer {
    void *qkv_w, *out_w, *fc1_w, *fc2_w;
    float *qkv_b, *out_b, *fc1_b, *fc2_b, *ln1_w, *ln1_b, *ln2_w, *ln2_b;
};
// Un-merged LoRA adapter of one (fused) linear y = x W^T (peft 0.13.2 tuners/lora/layer.py Linear.forward: result + lora_B(lora_A(x)) *
// scaling; the reference loads the adapter un-merged, eval/reward_adaptor_loader.py:44-45): t = x A^T is a GEMM of its own and
// y += t B^T rides in the K loop of the main GEMM as k2 extra columns (GemmParams::A2 / W2).  The scaling alpha / r is folded
// into B at upload.  A fused linear whose parts carry separate adapters (q/k/v, gate/up) stacks their A matrices (k2 = parts x
// padded rank) and holds B block-diagonally.
struct Lora {
    void* A = nullptr;         // [k2, K] operand dtype
    void* B = nullptr;         // [N, k2] operand dtype, rows packed like the base weight's
    int k2 = 0;                // 0 = no adapter
};
struct DecLayer {
    void *qkv_w, *o_w, *gu_w, *down_w;
    float *ln1, *ln2;
    float* qkv_b = nullptr;    // Qwen2.5-VL: q/k/v bias, packed like the rows of qkv_w
    Lora lqkv, lo, lgu, ldown;
};
struct VitLayer {              // Qwen2_5_VLVisionBlock
    void *qkv_w, *proj_w, *gu_w, *down_w;
    float *qkv_b, *proj_b, *gu_b, *down_b, *n1, *n2;
};

struct lr_engine {
    lr_model_desc d;
    int device = 0;
    std::string err;
    bool finalized = false;
    std::vector<void*> allocs;
    std::vector<Slot> slots;
    std::unordered_map<std::string, int> index;
    // split-operand mode: every operand-typed weight buffer has a residual twin; a buffer is "inexact" once any residual is non-zero
    struct WBuf { char* hi; char* lo; size_t bytes; };
    std::vector<WBuf> wbufs;
    std::unordered_map<const void*, int> wbuf_of;      // hi base pointer -> index
    int* inexact_dev = nullptr;
    std::vector<int> inexact;
    size_t ws_bytes = 0, weight_bytes = 0;
    int gemm_tile = -1, lim_clip = -1, lim_layers = -1;
    uint64_t weights_epoch = 0;        // lr_weights_epoch: bumped by every call that changes a weight

    // derived
    int T = 0, G = 0, Kpatch = 0, Kpad = 0, hd = 0, half = 0, Vcap = 0;
    int Hq = 0, Hkv = 0, Nqkv = 0;      // decoder projection widths: heads*hd, kv_heads*hd, Hq + 2*Hkv
    bool llava = false, qwen = false;
    int op_dt = DT_BF16;
    int prec = 0;              // split-operand mode: operand buffers are [hi | lo], twice as wide
    int lo8 = 0;               // ... with the residual pass of the big GEMMs in e4m3 (desc.precise == 2)
    // block scales of one-byte residuals (common.h lo8_scale_at): two arrays used in turn -- a GEMM reads its operand's scales from
    // one while its epilogue writes its output's into the other -- and the row exponents of the e4m3 copy of a hi half
    unsigned char* sc[2] = {nullptr, nullptr}; size_t sc_cap = 0; int sc_i = 0;
    const unsigned char* pre_enc_sc = nullptr;       // the scale array that belongs to pre_enc
    int* aexp2 = nullptr; size_t aexp2_cap = 0;
    unsigned* amax_word = nullptr;
    std::unordered_map<const void*, int> w8exp;      // weight base pointer -> E8M0 exponent of its prepared e4m3 twin
    std::unordered_map<const void*, int> w8exp2;     // ... and of e4m3(W_lo) for weights that are inexact in the operand type
    void* w8tmp = nullptr; size_t w8tmp_cap = 0;
    const void* pre_enc = nullptr;                   // operand buffer whose residual half its producer has already written in e4m3
    bool pre_enc_hi8 = false;                        // ... together with the e4m3 copy of its hi half (third e4m3 segment)
    std::unordered_map<const void*, void*> own8;     // adapter A matrices: weight base pointer -> its e4m3 twin in a buffer of its own
    // weights that are NOT exact in the operand type: their rows [e4m3(W) | e4m3(W_lo)] live in a buffer of their own, so that the
    // 16-bit residuals stay intact in the residual twin and the handle can still run the strict form (lr_set_precision_map)
    std::unordered_map<const void*, void*> pair8;
    int lora_rp = 0, lora_k2max = 0;                 // adapter rank padded to a K-tile (64); widest K-extension of any linear
    void* lt = nullptr;                              // t = x A^T of the linear being launched, [rows, k2 (x2 in split-operand mode)]
    int w8a8 = 0;              // W8A8 mode (desc.w8a8): e4m3 GEMM operands with per-row / per-channel fp32 scales
    struct W8 { void* q; float* scale; };
    std::unordered_map<const void*, W8> w8;          // weight base pointer -> its e4m3 twin (prepared on first use)
    void* q8 = nullptr; size_t q8_cap = 0;           // quantised rows of the A operand of the GEMM being launched
    float* q8s = nullptr; size_t q8s_cap = 0;

    // Qwen2.5-VL ViT geometry: head dim vhd stored vhdp wide, MLP width vI stored vIp wide, patch vector vK padded to vKpad
    int vH = 0, vhd = 0, vhdp = 0, vHp = 0, vI = 0, vIp = 0, vK = 0, vKpad = 0, vHm = 0, vunit = 0;
    std::vector<VitLayer> vl;
    void *vpatch_w = nullptr, *m0_w = nullptr, *m2_w = nullptr;
    float *vlnq = nullptr, *m0_b = nullptr, *m2_b = nullptr, *vinv = nullptr, *ca_u = nullptr;
    void *vA = nullptr, *vhn = nullptr, *vqkv = nullptr, *vatt = nullptr, *vff = nullptr, *vm1 = nullptr;
    float *vx = nullptr, *vqkv32 = nullptr, *vcs = nullptr;
    int *pos3 = nullptr, *rstat = nullptr;
    int lastP = 0;

    // weights
    float *cls = nullptr, *pos = nullptr, *pre_w = nullptr, *pre_b = nullptr;
    void* patch_w = nullptr;
    std::vector<ClipLayer> cl;
    float *sub_gn = nullptr, *glb_gn = nullptr, *p0_b = nullptr, *p2_b = nullptr, *newline = nullptr;
    void *p0_w = nullptr, *p2_w = nullptr;
    unsigned short* wte = nullptr;
    std::vector<DecLayer> dl;
    float* norm_w = nullptr;
    float *Wq = nullptr, *WkT = nullptr, *Wv = nullptr;      // SkipCA projections for the fp32 tail, kept in fp32
    float *ca_w = nullptr, *vh = nullptr;
    float *inv_s = nullptr, *inv_l = nullptr;

    // staging for uploads
    void* stage_raw = nullptr; size_t stage_raw_cap = 0;
    float* stage_f32 = nullptr; size_t stage_f32_cap = 0;

    // workspace
    void *patchA = nullptr, *clip_h = nullptr, *clip_qkv = nullptr, *clip_att = nullptr, *clip_ff = nullptr;
    float *patch_out = nullptr, *clip_x = nullptr;
    void *hdA = nullptr, *proj1 = nullptr;
    float *ev = nullptr, *pf32 = nullptr;
    float *x = nullptr, *qkv32 = nullptr, *cs = nullptr;
    void *h = nullptr, *qkv = nullptr, *att = nullptr, *ff = nullptr;
    int *pos_ids = nullptr, *img_row = nullptr, *tstat = nullptr;
    float *mh_h = nullptr, *mh_t1 = nullptr, *mh_t2 = nullptr, *mh_sc = nullptr;   // mean-pooling head (fp32, all tokens)
    float *hL = nullptr, *tq = nullptr, *tkq = nullptr, *tsc = nullptr, *tctx = nullptr, *tao = nullptr;
    // per-forward tables (ring of pinned host slots + device mirrors)
    static constexpr int NSLOT = 4;
    int slot_i = 0;
    char* tab_host = nullptr; char* tab_dev = nullptr; size_t tab_bytes = 0;
    hipEvent_t tab_ev[NSLOT] = {}; bool tab_used[NSLOT] = {};
    // last forward geometry (for taps)
    int lastB = 0, lastS = 0, lastNC = 0, lastSV = 0, lastVmax = 0;
    std::vector<int> last_voff;                      // first row in ev of every sample of the last forward (+ the end), lr_vision_embeds
    bool last_pruned = false;                        // the last forward ran its final decoder layer for the gathered rows only (x is stale there)
    // last decoder layer, gathered rows only (run_decoder_stack): compact [max_batch (+ pad), ...] twins of x / h / att / ff
    float* xg = nullptr; void *hg = nullptr, *attg = nullptr, *ffg = nullptr;
    int* sched_mem = nullptr;                        // tile-scheduler words of this engine's persistent GEMM launches (GemmParams::sched_mem)
    // precision map (lr_set_precision_map): operand form per stage, -1 = the descriptor's.  0 single pass, 1 split (16-bit residuals),
    // 2 split with e4m3 residual passes.  Decoder layers [pm_first, layers - pm_last) take pm_mid, the others the descriptor's form.
    int prec0 = 0, lo8_0 = 0, pm_clip = -1, pm_mid = -1, pm_first = 0, pm_last = 0;
    // ... and per SITE of a decoder layer in that range (lr_set_precision_sites, round 6): -1 = the layer's form.  A site = an operand
    // producer together with the GEMM (or attention launch) that reads it, so a site's form never crosses into its neighbours'.
    enum : int { SITE_QKV = 0, SITE_ATTN = 1, SITE_O = 2, SITE_GATE_UP = 3, SITE_DOWN = 4, N_SITES = 5 };
    int pm_site[N_SITES] = {-1, -1, -1, -1, -1};
    // lazy reference maximum of the attention launches in DEFAULT-form stages (lr_set_attention_lazy_threshold; log2 units, 0 = exact).
    // Strict stages run the exact maximum (att_lazy_t_strict = 0) except in the probe's noise-floor pass (model.py _compare_forms).
    float att_lazy_t = ATT_LAZY_T_DEFAULT, att_lazy_t_strict = 0.f;
    void set_form(int m) { if (m < 0) { prec = prec0; lo8 = lo8_0; } else { prec = m ? 1 : 0; lo8 = m == 2 ? 1 : 0; } pre_enc = nullptr; }
    int layer_form(int l) const { return (pm_mid >= 0 && l >= pm_first && l < d.layers - pm_last) ? pm_mid : -1; }
    int site_form(int l, int site) const { const int f = layer_form(l); return (f >= 0 && pm_site[site] >= 0) ? pm_site[site] : f; }

    void* dalloc(size_t bytes, bool weight) {
        void* p = nullptr;
        bytes = (bytes + 255) & ~(size_t)255;
        LR_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 256));
        if (weight) LR_HIP_CHECK(hipMemset(p, 0, bytes ? bytes : 256));     // padded operand layouts rely on zero fill
        allocs.push_back(p);
        (weight ? weight_bytes : ws_bytes) += bytes;
        return p;
    }
    // pair8 rows ([e4m3(W) | e4m3(W_lo)], +1x the weight's bytes) of weight buffers that are exact in the operand type AGAIN (a
    // re-upload / re-synthesis replaced merged-adapter or fp32-valued weights): nothing reads them any more -- give the HBM back.
    // Called where `inexact` has just been read back, behind a device synchronisation.
    void release_stale_pair8() {
        for (auto it = pair8.begin(); it != pair8.end();) {
            auto wb = wbuf_of.find(it->first);
            if (wb == wbuf_of.end() || inexact.empty() || !it->second) { ++it; continue; }      // (not a tracked buffer / flags never read back: keep)
            if (inexact[wb->second]) { ++it; continue; }                                        // still inexact: still read
            const size_t bytes = (wbufs[wb->second].bytes + 255) & ~(size_t)255;
            for (auto a = allocs.begin(); a != allocs.end(); ++a) if (*a == it->second) { allocs.erase(a); break; }
            (void)hipFree(it->second);
            weight_bytes -= bytes < weight_bytes ? bytes : weight_bytes;
            w8exp.erase(it->first); w8exp2.erase(it->first);
            it = pair8.erase(it);
        }
    }
    size_t opsz() const { return 2; }
};

inline void add_slot(lr_engine* e, const std::string& name, std::vector<int64_t> shape, void* dst, int ld_dst, int cols_dst,
              int dst_dtype, int mode, double std_, double offset) {
    Slot s;
    s.name = name;
    s.shape = shape;
    if (shape.size() == 1) { s.rows = 1; s.cols = (int)shape[0]; }
    else { s.rows = (int)shape[0]; int64_t c = 1; for (size_t i = 1; i < shape.size(); ++i) c *= shape[i]; s.cols = (int)c; }
    s.dst = dst; s.ld_dst = ld_dst; s.cols_dst = cols_dst; s.dst_dtype = dst_dtype; s.mode = mode;
    s.std_ = std_; s.offset = offset;
    if (e->prec && (dst_dtype == DT_F16 || dst_dtype == DT_BF16)) {
        for (size_t i = 0; i < e->wbufs.size(); ++i) {
            const lr_engine::WBuf& w = e->wbufs[i];
            if ((char*)dst >= w.hi && (char*)dst < w.hi + w.bytes) { s.lo_dst = w.lo + ((char*)dst - w.hi); s.wid = (int)i; break; }
        }
    }
    e->index[name] = (int)e->slots.size();
    e->slots.push_back(s);
}

inline float* falloc(lr_engine* e, size_t n) { return (float*)e->dalloc(n * 4, true); }
inline void* oalloc(lr_engine* e, size_t n) {
    void* hi = e->dalloc(n * 2, true);
    if (e->prec) {
        void* lo = e->dalloc(n * 2, true);
        e->wbuf_of[hi] = (int)e->wbufs.size();
        e->wbufs.push_back({(char*)hi, (char*)lo, n * 2});
    }
    return hi;
}

inline void vec_slot(lr_engine* e, const std::string& name, std::vector<int64_t> shape, float* dst, double std_, double off) {
    int64_t n = 1; for (auto v : shape) n *= v;
    add_slot(e, name, shape, dst, (int)n, (int)n, DT_F32, PACK_PLAIN, std_, off);
    // 1-D view of multi-dim vectors (glb_GN [1,1,4H]) : force [1, n]
    e->slots.back().rows = 1; e->slots.back().cols = (int)n;
}

inline void register_clip(lr_engine* e, const std::string& cp) {
    const lr_model_desc& d = e->d;
    const int Hc = d.clip_hidden, Mc = d.clip_mlp;
    const int od = e->op_dt;
    e->cls = falloc(e, Hc);
    vec_slot(e, cp + "embeddings.class_embedding", {Hc}, e->cls, 0.02, 0);
    e->patch_w = oalloc(e, (size_t)Hc * e->Kpad);
    add_slot(e, cp + "embeddings.patch_embedding.weight", {Hc, 3, d.clip_patch, d.clip_patch}, e->patch_w, e->Kpad, e->Kpad, od,
             PACK_PLAIN, 0.02, 0);
    e->pos = falloc(e, (size_t)e->T * Hc);
    add_slot(e, cp + "embeddings.position_embedding.weight", {e->T, Hc}, e->pos, Hc, Hc, DT_F32, PACK_PLAIN, 0.02, 0);
    e->pre_w = falloc(e, Hc); e->pre_b = falloc(e, Hc);
    vec_slot(e, cp + "pre_layrnorm.weight", {Hc}, e->pre_w, 0.05, 1.0);
    vec_slot(e, cp + "pre_layrnorm.bias", {Hc}, e->pre_b, 0.02, 0);
    e->cl.resize(d.clip_layers);
    for (int l = 0; l < d.clip_layers; ++l) {
        ClipLayer& c = e->cl[l];
        const std::string p = cp + "encoder.layers." + std::to_string(l) + ".";
        c.qkv_w = oalloc(e, (size_t)3 * Hc * Hc); c.qkv_b = falloc(e, 3 * Hc);
        const char* nm[3] = {"q_proj", "k_proj", "v_proj"};
        for (int i = 0; i < 3; ++i) {
            add_slot(e, p + "self_attn." + nm[i] + ".weight", {Hc, Hc}, (char*)c.qkv_w + (size_t)i * Hc * Hc * 2, Hc, Hc, od,
                     PACK_PLAIN, 0.02, 0);
            vec_slot(e, p + "self_attn." + nm[i] + ".bias", {Hc}, c.qkv_b + i * Hc, 0.02, 0);
        }
        c.out_w = oalloc(e, (size_t)Hc * Hc); c.out_b = falloc(e, Hc);
        add_slot(e, p + "self_attn.out_proj.weight", {Hc, Hc}, c.out_w, Hc, Hc, od, PACK_PLAIN, 0.02, 0);
        vec_slot(e, p + "self_attn.out_proj.bias", {Hc}, c.out_b, 0.02, 0);
        c.ln1_w = falloc(e, Hc); c.ln1_b = falloc(e, Hc); c.ln2_w = falloc(e, Hc); c.ln2_b = falloc(e, Hc);
        vec_slot(e, p + "layer_norm1.weight", {Hc}, c.ln1_w, 0.05, 1.0);
        vec_slot(e, p + "layer_norm1.bias", {Hc}, c.ln1_b, 0.02, 0);
        c.fc1_w = oalloc(e, (size_t)Mc * Hc); c.fc1_b = falloc(e, Mc);
        add_slot(e, p + "mlp.fc1.weight", {Mc, Hc}, c.fc1_w, Hc, Hc, od, PACK_PLAIN, 0.02, 0);
        vec_slot(e, p + "mlp.fc1.bias", {Mc}, c.fc1_b, 0.02, 0);
        c.fc2_w = oalloc(e, (size_t)Hc * Mc); c.fc2_b = falloc(e, Hc);
        add_slot(e, p + "mlp.fc2.weight", {Hc, Mc}, c.fc2_w, Mc, Mc, od, PACK_PLAIN, 0.02, 0);
        vec_slot(e, p + "mlp.fc2.bias", {Hc}, c.fc2_b, 0.02, 0);
        vec_slot(e, p + "layer_norm2.weight", {Hc}, c.ln2_w, 0.05, 1.0);
        vec_slot(e, p + "layer_norm2.bias", {Hc}, c.ln2_b, 0.02, 0);
    }
}

inline void upload_rope_tables(lr_engine* e) {
    e->inv_s = falloc(e, LR_MAX_HALF_HEAD); e->inv_l = falloc(e, LR_MAX_HALF_HEAD);
    LR_HIP_CHECK(hipMemcpy(e->inv_s, e->d.inv_freq_short, sizeof(e->d.inv_freq_short), hipMemcpyHostToDevice));
    LR_HIP_CHECK(hipMemcpy(e->inv_l, e->d.inv_freq_long, sizeof(e->d.inv_freq_long), hipMemcpyHostToDevice));
}


template <typename F> int guarded(lr_engine* e, F&& f) {
    try {
        if (e) LR_HIP_CHECK(hipSetDevice(e->device));
        f();
        return LR_OK;
    } catch (const std::invalid_argument& ex) {
        (e ? e->err : g_create_error) = ex.what();
        return LR_EINVAL;
    } catch (const std::logic_error& ex) {
        (e ? e->err : g_create_error) = ex.what();
        return LR_ESTATE;
    } catch (const std::exception& ex) {
        (e ? e->err : g_create_error) = ex.what();
        return LR_EHIP;
    }
}

// Split-operand mode: callers pass LOGICAL shapes; here A becomes [A_hi | A_lo] (K doubled, W re-read from column 0) and an
// operand-typed output becomes [C_hi | C_lo].
// With desc.precise == 2 the residual pass of a GEMM that runs on the deep-pipelined kernel is made in e4m3: the residual half of
// A arrives as block-scaled e4m3 from its producer (norms, SwiGLU / operand-out epilogues: pre_enc) or is re-encoded in place
// (stream-ordered, right here: every operand buffer feeds exactly one GEMM after it is produced), and W's e4m3 twin is prepared on
// first use in the rows of its residual buffer (exact weights only; one synchronisation per weight).
// apply_prec_base only rewrites the parameters (no side effects); upgrade_lo8 commits the e4m3 form right before the launch.
inline void apply_prec_base(const lr_engine* e, GemmParams& p) {
    if (!e->prec) return;
    p.kw = p.K; p.K *= 2; p.lda *= 2;
    auto it = e->wbuf_of.find(p.W);
    if (it != e->wbuf_of.end() && !e->inexact.empty() && e->inexact[it->second]) {     // weights not exact in the operand type
        p.Wlo = e->wbufs[it->second].lo;
        p.K = 3 * p.kw;
    }
    if (p.epi == EPI_OUT_OP || p.epi == EPI_SWIGLU_OP || p.epi == EPI_ROPE_OP) { p.split = p.ldc; p.ldc *= 2; }
}
// The choice must not depend on M: a row's reward has to be bit-identical whatever else is in the batch (sharding across GPUs
// must not change a preference), so every eligible GEMM takes this form -- on the deep-pipelined kernel -- at any row count.
inline bool lo8_eligible(const lr_engine* e, const GemmParams& p) {      // p after apply_prec_base
    if (!e->lo8 || p.kw <= 0) return false;
    if (e->wbuf_of.find(p.W) == e->wbuf_of.end()) return false;
    const bool aligned = p.N % 8 == 0 && p.ldc % 8 == 0 && (((uintptr_t)p.C) & 15) == 0 && (!p.bias || (((uintptr_t)p.bias) & 15) == 0);
    return aligned && p.kw % 128 == 0 && p.ldw == p.kw && (e->gemm_tile < 0 || e->gemm_tile == 6);
}
// e4m3 twin(s) of weight W [N, K] in the rows of its residual buffer; synchronous.  lr_finalize calls it for every GEMM weight it
// knows (prepare_twins below), the launch path only as a fallback (weights re-uploaded after lr_finalize).
inline void ensure_lo8_twin(lr_engine* e, const void* W, int N, int K, int ldw, hipStream_t st) {
    if (e->w8exp.count(W)) return;
    auto it = e->wbuf_of.find(W);
    if (it == e->wbuf_of.end()) return;
    char* twin = e->wbufs[it->second].lo;
    const bool inexact = !e->inexact.empty() && e->inexact[it->second];
    if (!e->amax_word) e->amax_word = (unsigned*)e->dalloc(256, false);
    int E = 127, E2 = 127;
    auto own = e->own8.find(W);
    if (own != e->own8.end()) {          // adapter matrix: e4m3(W) in its own buffer, the 16-bit residuals stay where they are
        E = prepare_weight_e4m3(W, ldw, K, N, own->second, e->op_dt, e->amax_word, st);
    } else if (inexact) {
        const size_t need = (size_t)N * ldw * 2;
        if (need > e->w8tmp_cap) { e->w8tmp_cap = need; e->w8tmp = e->dalloc(need, false); }
        void*& pair = e->pair8[W];
        if (!pair) pair = e->dalloc(need, true);
        LR_HIP_CHECK(hipMemcpyAsync(pair, twin, need, hipMemcpyDeviceToDevice, st));      // the 16-bit residuals: converted in the copy
        prepare_weight_e4m3_pair(W, pair, ldw, K, N, e->w8tmp, e->op_dt, e->amax_word, st, &E, &E2);
    } else {
        E = prepare_weight_e4m3(W, ldw, K, N, twin, e->op_dt, e->amax_word, st);
    }
    e->w8exp[W] = E;
    e->w8exp2[W] = E2;
}
// scale arrays (and the hi-half row exponents) large enough for an operand of `rows` x K; presized at lr_finalize (prepare_twins)
inline void ensure_aexp(lr_engine* e, size_t rows, size_t K) {
    const size_t need = lo8_scale_bytes((int)rows, (int)K);
    if (need > e->sc_cap) {
        // (presized at lr_finalize; a growth during a forward is a fallback: hipMalloc synchronises the device, and the fill is
        //  ordered for every stream by the device-wide synchronisation behind it)
        e->sc_cap = (need + 65535) & ~(size_t)65535;
        for (int i = 0; i < 2; ++i) {
            e->sc[i] = (unsigned char*)e->dalloc(e->sc_cap, false);
            LR_HIP_CHECK(hipMemset(e->sc[i], 127, e->sc_cap));           // rows beyond M of a last tile read these: any finite scale
        }
        LR_HIP_CHECK(hipDeviceSynchronize());
    }
    if (rows > e->aexp2_cap) {
        e->aexp2_cap = (rows + 4095) & ~(size_t)4095;
        e->aexp2 = (int*)e->dalloc(e->aexp2_cap * 4, false);
    }
}
// the array the next producer of one-byte residuals writes its scales to
inline unsigned char* next_scales(lr_engine* e) { e->sc_i ^= 1; return e->sc[e->sc_i]; }

// keep_enc: another GEMM reads the same operand buffer next (the main GEMM behind an adapter's t GEMM): leave the "already encoded"
// mark on it.  force_hi8: that next GEMM needs the e4m3 copy of the hi half as well, so the one in-place encoding pass writes it now.
inline void upgrade_lo8(lr_engine* e, GemmParams& p, hipStream_t st, bool keep_enc = false, bool force_hi8 = false) {
    if (p.aexp) return;
    if (lo8_eligible(e, p)) {
        // Weights inexact in the operand type (Wlo set by apply_prec_base: a merged LoRA adapter, fp32-trained weights): a third
        // e4m3 segment, A_hi8 x e4m3(W_lo)^T, replaces the 16-bit [x_hi] x [W_lo] segment (2x instead of 3x the single pass).
        // Adapter matrices (own8) keep the 16-bit third segment: their e4m3 twin lives in a buffer of its own.
        const bool inexact = p.Wlo != nullptr;
        auto own = e->own8.find(p.W);
        const bool third8 = inexact && own == e->own8.end();
        const bool hi8 = third8 || force_hi8;
        auto it = e->wbuf_of.find(p.W);
        char* lo_rows = e->wbufs[it->second].lo;
        ensure_lo8_twin(e, p.W, p.N, p.kw, p.ldw, st);
        auto w8 = e->w8exp.find(p.W);
        ensure_aexp(e, (size_t)p.M, (size_t)p.kw);
        int* aexp2 = e->aexp2;
        if (!(e->pre_enc == p.A && (!hi8 || e->pre_enc_hi8))) {
            if (e->pre_enc == p.A) throw std::logic_error("operand buffer already carries e4m3 residuals without the e4m3 copy of its hi half");
            unsigned char* sc = next_scales(e);
            launch_quantize_lo_inplace(const_cast<void*>(p.A), p.lda, p.kw, p.M, sc, e->op_dt, st, hi8 ? aexp2 : nullptr);
            e->pre_enc_hi8 = hi8;
            e->pre_enc_sc = sc;
        }
        p.aexp = e->pre_enc_sc;
        e->pre_enc = keep_enc ? p.A : nullptr;      // (a norm kernel may have written [hi | e4m3(lo)] and the scales itself: lo8_norm_target)
        if (own != e->own8.end()) { p.Wlo16 = inexact ? lo_rows : nullptr; p.Wlo = own->second; }
        else if (inexact) p.Wlo = e->pair8.at(p.W);
        else p.Wlo = lo_rows;
        p.wexp = w8->second;
        p.aexp2 = third8 ? aexp2 : nullptr;
        p.wexp2 = e->w8exp2[p.W];
    }
}
inline void apply_prec(lr_engine* e, GemmParams& p, hipStream_t st) {
    apply_prec_base(e, p);
    upgrade_lo8(e, p, st);
}
// For a norm kernel that feeds the GEMM described by `probe` (logical shapes, as passed to gemm()): where to put the block scales if
// that GEMM will take the e4m3 residual form with exact weights, else null (the norm then writes 16-bit residuals as usual).
inline unsigned char* lo8_norm_target(lr_engine* e, GemmParams probe) {
    e->pre_enc = nullptr;
    if (!e->lo8) return nullptr;
    apply_prec_base(e, probe);
    if (!lo8_eligible(e, probe) || probe.Wlo) return nullptr;
    ensure_aexp(e, (size_t)probe.M, (size_t)probe.kw);
    e->pre_enc = probe.A;
    e->pre_enc_hi8 = false;
    unsigned char* sc = next_scales(e);
    e->pre_enc_sc = sc;
    return sc;
}
inline void apply_prec(const lr_engine* e, AttnParams& p) {
    if (!e->prec) return;
    p.lo_off = p.ldq; p.ldq *= 2;
    p.o_split = p.ldo; p.ldo *= 2;
    // strict stages (16-bit residual passes: the probe's yardstick, and what an amplifying weight set is locked to) keep the exact
    // softmax maximum; stages in the default form take the lazy one (attention.hip; -14 % per launch)
    // (A/B switches for tools/outlier_fp64_probe.py, read once: LR_ATT_LAZY_T overrides the default-form stages' threshold, LR_ATT_LAZY_T_STRICT the strict stages')
    static const float t_def = [] { const char* v = getenv("LR_ATT_LAZY_T"); return v ? (float)atof(v) : -1.f; }();
    static const float t_strict = [] { const char* v = getenv("LR_ATT_LAZY_T_STRICT"); return v ? (float)atof(v) : 0.f; }();
    p.lazy_t = e->lo8 ? (t_def >= 0.f ? t_def : e->att_lazy_t) : (t_strict > 0.f ? t_strict : e->att_lazy_t_strict);
}

// W8A8 mode: quantise the rows of A, make sure W has its e4m3 twin, launch the e4m3 form.  Like lo8_eligible, the choice never
// depends on M.  Returns false when this GEMM has to stay in 16 bits (K % 128 != 0, unaligned output).
inline bool w8a8_eligible(const lr_engine* e, const GemmParams& p) {
    const bool aligned = p.N % 8 == 0 && p.ldc % 8 == 0 && (((uintptr_t)p.C) & 15) == 0 && (!p.bias || (((uintptr_t)p.bias) & 15) == 0);
    return e->w8a8 && aligned && p.K % 128 == 0 && p.lda % 8 == 0 && p.ldw % 8 == 0 && (e->gemm_tile < 0 || e->gemm_tile == 6);
}
inline void ensure_w8a8_twin(lr_engine* e, const void* W, int N, int K, int ldw, hipStream_t st) {
    if (e->w8.count(W)) return;
    lr_engine::W8 t{e->dalloc((size_t)N * K, true), (float*)e->dalloc((size_t)N * 4, true)};
    launch_quantize_rows_fp8(W, ldw, K, N, t.q, K, t.scale, e->op_dt, st);
    e->w8.emplace(W, t);
}
inline void ensure_q8(lr_engine* e, size_t rows, size_t K) {
    const size_t need = rows * K;
    if (need > e->q8_cap) { e->q8_cap = (need + (1u << 20)) & ~(size_t)((1u << 20) - 1); e->q8 = e->dalloc(e->q8_cap, false); }
    if (rows > e->q8s_cap) { e->q8s_cap = (rows + 4095) & ~(size_t)4095; e->q8s = (float*)e->dalloc(e->q8s_cap * 4, false); }
}
// lr_finalize: twins of every GEMM weight {W, N, K} with K % 128 == 0 and scratch for `rows` operand rows, so that no forward has to
// synchronise or allocate (weights re-uploaded afterwards fall back to preparation at their first launch).
struct GemmWeight { const void* W; int N, K; size_t rows; };
inline void prepare_twins(lr_engine* e, const std::vector<GemmWeight>& ws) {
    if (!e->lo8 && !e->w8a8) return;
    for (const GemmWeight& g : ws) {
        if (!g.W || g.K % 128) continue;
        if (e->lo8) { ensure_lo8_twin(e, g.W, g.N, g.K, g.K, 0); ensure_aexp(e, g.rows, (size_t)g.K); }
        else { ensure_w8a8_twin(e, g.W, g.N, g.K, g.K, 0); ensure_q8(e, g.rows, (size_t)g.K); }
    }
    LR_HIP_CHECK(hipStreamSynchronize(0));
}

inline bool launch_w8a8(lr_engine* e, GemmParams p, hipStream_t st) {
    if (!w8a8_eligible(e, p)) return false;
    ensure_w8a8_twin(e, p.W, p.N, p.K, p.ldw, st);
    auto it = e->w8.find(p.W);
    ensure_q8(e, (size_t)p.M, (size_t)p.K);
    launch_quantize_rows_fp8(p.A, p.lda, p.K, p.M, e->q8, p.K, e->q8s, e->op_dt, st);
    p.A = e->q8; p.W = it->second.q; p.lda = p.K; p.ldw = p.K;
    p.ascale = e->q8s; p.wscale = it->second.scale;
    launch_gemm_bt8_fp8(p, e->op_dt, st);
    return true;
}

// For a GEMM whose operand-typed output C [M, Kout] is read next by the GEMM with weight next_W [next_N, Kout]: where the producer's
// epilogue should put C's block scales if that consumer will take the e4m3 residual form with exact weights (it then finds its
// operand encoded: no in-place pass), else null.  p: the producer after apply_prec (its kernel must be the deep-pipelined one).
inline unsigned char* lo8_out_target(lr_engine* e, const GemmParams& p, const void* next_W, int next_N) {
    if (!e->lo8 || !next_W || p.split <= 0 || !(p.epi == EPI_OUT_OP || p.epi == EPI_SWIGLU_OP)) return nullptr;
    if (!(p.A2 || gemm_bt_is_deep(p, e->gemm_tile)) || (e->gemm_tile >= 0 && e->gemm_tile != 6)) return nullptr;
    const int Kout = p.split;                                  // logical output columns = the consumer's K
    GemmParams probe{p.C, next_W, nullptr, nullptr, p.M, next_N, Kout, Kout, Kout, next_N, EPI_OUT_F32, ACT_NONE, nullptr, 0, 0};
    apply_prec_base(e, probe);
    if (!lo8_eligible(e, probe) || probe.Wlo) return nullptr;
    ensure_aexp(e, (size_t)p.M, (size_t)Kout);
    return next_scales(e);
}
inline void mark_lo8_out(lr_engine* e, const GemmParams& p) {
    if (!p.oexp) return;
    e->pre_enc = p.C;
    e->pre_enc_hi8 = false;
    e->pre_enc_sc = p.oexp;
}

// One linear layer, p in LOGICAL shapes (as gemm() takes them).  With an adapter L: t = x A^T first (always on the deep-pipelined
// kernel, in the same operand form as the main GEMM -- they read the same rows of x), then the main GEMM with the K-extension.
// next_W / next_N: the weight of the GEMM that reads this one's output as its operand (lo8_out_target), or null.
// next_form: the operand form of the site that reads this GEMM's output (lr_engine::site_form), -2 = the form in force now: decides
// whether the epilogue writes one-byte residuals (lo8_out_target looks at the CONSUMER's form, not at this GEMM's).
inline unsigned char* lo8_out_target_for(lr_engine* e, const GemmParams& p, const void* next_W, int next_N, int next_form) {
    if (next_form == -2) return lo8_out_target(e, p, next_W, next_N);
    const int sp = e->prec, sl = e->lo8;
    if (next_form < 0) { e->prec = e->prec0; e->lo8 = e->lo8_0; } else { e->prec = next_form ? 1 : 0; e->lo8 = next_form == 2 ? 1 : 0; }
    unsigned char* t = (e->prec == sp) ? lo8_out_target(e, p, next_W, next_N) : nullptr;      // (a single-pass neighbour: no residual half at all)
    e->prec = sp; e->lo8 = sl;
    return t;
}
inline void gemm_p(lr_engine* e, hipStream_t st, GemmParams p, const Lora* L = nullptr, const void* next_W = nullptr, int next_N = 0, int next_form = -2) {
    p.sched_mem = e->sched_mem;
    if (L && L->k2 > 0) {
        if (e->w8a8) throw std::logic_error("W8A8 mode runs merged weights only (no un-merged adapters)");
        GemmParams probe = p;
        apply_prec_base(e, probe);
        GemmParams t{p.A, L->A, e->lt, nullptr, p.M, L->k2, p.K, p.lda, p.K, L->k2, EPI_OUT_OP, ACT_NONE, nullptr, 0, 0};
        t.sched_mem = e->sched_mem;
        apply_prec_base(e, t);
        const bool both8 = lo8_eligible(e, probe) && lo8_eligible(e, t);
        if (e->pre_enc == p.A && !both8) throw std::logic_error("operand pre-encoded in e4m3 for a GEMM that takes the 16-bit form");
        if (both8) upgrade_lo8(e, t, st, true, probe.Wlo != nullptr);
        launch_gemm_bt(t, e->op_dt, 6, st);
        p.A2 = e->lt; p.lda2 = L->k2 * (1 + e->prec); p.W2 = L->B; p.ldw2 = L->k2; p.k2 = L->k2;
        auto it = e->wbuf_of.find(L->B);
        if (it != e->wbuf_of.end() && !e->inexact.empty() && e->inexact[it->second]) p.W2lo = e->wbufs[it->second].lo;
        apply_prec_base(e, p);
        if (both8) upgrade_lo8(e, p, st);
        p.oexp = lo8_out_target_for(e, p, next_W, next_N, next_form);
        launch_gemm_bt(p, e->op_dt, e->gemm_tile, st);
        mark_lo8_out(e, p);
        return;
    }
    if (launch_w8a8(e, p, st)) return;
    apply_prec(e, p, st);
    p.oexp = lo8_out_target_for(e, p, next_W, next_N, next_form);
    launch_gemm_bt(p, e->op_dt, e->gemm_tile, st);
    mark_lo8_out(e, p);
}
inline void gemm(lr_engine* e, hipStream_t st, const void* A, const void* W, void* C, const float* bias, int M, int N, int K, int lda,
          int ldw, int ldc, int epi, int act, const Lora* L = nullptr, const void* next_W = nullptr, int next_N = 0, int next_form = -2) {
    gemm_p(e, st, GemmParams{A, W, C, bias, M, N, K, lda, ldw, ldc, epi, act, nullptr, 0, 0}, L, next_W, next_N, next_form);
}

// Adapter slots of one linear `mod` (checkpoint names mod.lora_A.weight [r, K], mod.lora_B.weight [n_rows, r]); part / parts: this
// adapter's place among the stacked adapters of a fused linear; b_row0: first row of the fused weight this part owns (B is block
// diagonal); pack_mode / aux_*: the row packing of the base weight's slot.  B must arrive pre-scaled by lora_alpha / r.
inline void register_lora(lr_engine* e, Lora& L, const std::string& mod, int N_fused, int K, int part, int parts, int n_rows, int b_row0,
                          int pack_mode, int aux_d = 0, int aux_hd = 0) {
    const int r = e->d.lora_rank, rp = e->lora_rp, od = e->op_dt;
    if (part == 0) {
        L.k2 = parts * rp;
        L.A = oalloc(e, (size_t)L.k2 * K);
        L.B = oalloc(e, (size_t)N_fused * L.k2);
        if (e->lo8) e->own8[L.A] = e->dalloc((size_t)L.k2 * K * 2, true);
        e->lora_k2max = std::max(e->lora_k2max, L.k2);
    }
    add_slot(e, mod + ".lora_A.weight", {r, K}, (char*)L.A + (size_t)part * rp * K * 2, K, K, od, PACK_PLAIN, 0.02, 0);
    add_slot(e, mod + ".lora_B.weight", {n_rows, r}, (char*)L.B + ((size_t)b_row0 * L.k2 + (size_t)part * rp) * 2, L.k2, r, od, pack_mode, 0.02, 0);
    e->slots.back().aux_d = aux_d; e->slots.back().aux_hd = aux_hd;
}


// engine.hip: pre-norm decoder stack shared by the three backbones (x, cs, tstat prepared by the caller).
// gather: 0 = every row through every layer; 1 / 2 = only the row of each sequence the reward is read from is wanted afterwards
// (1: its last valid position, tstat[4 b]; 2: position S - 1): the LAST layer then runs its attention, o_proj and MLP for those B
// rows only (rw_model:408-421 reads one row per sample) and leaves them in h->xg [B, hidden]; returns true when it did.
bool run_decoder_stack(lr_engine* h, hipStream_t st, const int64_t* attention_mask, int B, int S, int gather = 0);
void alloc_gather_ws(lr_engine* h);
// engine.hip: rw_model:398-406 -- SkipCA for every token (o: per-token term, u: per-sample term, either may be null), masked mean
// pooling into h->hL, value head.  Vb / voff: image rows of every sample in h->ev (host), Vmax = their maximum.
void run_mean_pool_head(lr_engine* h, hipStream_t st, const int64_t* attention_mask, int B, int S, const int* voff_host, int Vmax,
                        const float* u, bool final_norm, float* rewards_out);
void alloc_mean_pool(lr_engine* h, size_t rows, size_t vcap);
// qwen.hip
void build_weight_table_qwen(lr_engine* e);
void validate_desc_qwen(const lr_model_desc& d);
void finalize_qwen(lr_engine* h);
