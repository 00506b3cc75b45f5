"""Training-side reuse and the device-side batch builder (§8 row a17) on the GPU.

* llava_reward_amd.trainer_shim.concatenated_forward / evaluate mirror GeneralPreferenceRewardTrainer.concatenated_forward
  (rm_trainer_general_preference.py:447-460) and its evaluate loop (:381-445); the train-mode return conventions
  (rw_model_general_preference.py:410-415, :429-434) are pinned by goldens the reference produced in model.train() mode
  (tests/golden/ref_small_train_*.json, make_goldens.py `train`; run by test_gpu_forward.test_reference_goldens_small).
* batch_inference_process_phi3v_device: ragged captions -> one left-padded [B, S] batch on the device
  (reward_dataset.py:164-179 + datasets/utils.py:5-13)."""
import numpy as np
import pytest
import torch

from llava_reward_amd import preprocess as P
from llava_reward_amd import synth, trainer_shim
from llava_reward_amd.model import RewardModel
from oracle import phi3v_hd_transform_oracle as HD
from oracle import phi3v_reward_oracle as orc

pytestmark = pytest.mark.gpu


def _pairwise_loss(tau):
    """PairWiseLoss.forward of the reference's trainer (llava_reward/models/loss.py:120-129), as a trainer would pass it in."""
    def f(c, r, margin=None):
        d = (c - r - margin) if margin is not None else (c - r)
        return -torch.nn.functional.logsigmoid(d / tau).mean(), torch.sigmoid(d / tau).mean()
    return f


def test_concatenated_forward_and_evaluate_match_the_oracle():
    cfg = synth.tiny_config()
    seed = 61
    m = RewardModel(cfg, synth_seed=seed, max_batch=4, max_seq=1024, max_crops=5, operand_dtype="f16x2").to("cuda")
    W = orc.weights_to_torch(synth.make_weights(cfg, seed))
    batches = []
    for k in range(2):
        bc = synth.synth_batch(cfg, seed + k, [5, 8, 3], (1, 1))
        br = synth.synth_batch(cfg, seed + 10 + k, [4, 4, 9], (1, 1))
        # the dataloader's layout: every tensor with the collate's singleton dim (rm_trainer...:394-402 squeezes it)
        batches.append(tuple(torch.from_numpy(b[key])[:, None] for b in (bc, br) for key in ("input_ids", "attention_mask", "pixel_values", "image_sizes")))
    # train mode: last position, BT -> [B]
    m.train()
    c_ids, c_mask, c_pix, c_sz, r_ids, r_mask, r_pix, r_sz = [t.squeeze(1) for t in batches[0]]
    c, r, outs = trainer_shim.concatenated_forward(m, c_ids.cuda(), c_mask.cuda(), c_pix.cuda(), c_sz, r_ids.cuda(), r_mask.cuda(), r_pix.cuda(), r_sz,
                                                   return_output=True)
    assert c.shape == (3,) and r.shape == (3,) and len(outs) == 2
    ref_c = orc.custom_forward(W, cfg, c_ids, c_mask, c_pix, c_sz, training=True).squeeze(-1)
    ref_r = orc.custom_forward(W, cfg, r_ids, r_mask, r_pix, r_sz, training=True).squeeze(-1)
    assert (c.cpu() - ref_c).abs().max().item() < 1e-4 and (r.cpu() - ref_r).abs().max().item() < 1e-4
    # outputs["last_hidden_state"] (what evaluate reads, :414): final norm of every token, [B, S, D]
    taps = {}
    orc.custom_forward(W, cfg, r_ids, r_mask, r_pix, r_sz, training=True, taps=taps)
    x = taps[f"layer{cfg.layers - 1}"]
    want = orc.rms_norm(x, W["model.norm.weight"], cfg.rms_eps)
    got = outs[1]["last_hidden_state"].cpu()
    valid = r_mask.bool()
    assert got.shape == want.shape and (got - want)[valid].abs().max().item() < 2e-4 * want[valid].abs().max().item() + 1e-5
    # evaluate: eval-mode rewards (EOS gather, [B, 1]), the trainer's loss, means over the batches, model left in train mode
    tau = cfg.general_preference_tau
    stats = trainer_shim.evaluate(m, batches, _pairwise_loss(tau))
    assert m.training is True
    exp_loss, exp_prob = [], []
    for b in batches:
        ci, cm, cp, cs, ri, rm, rp, rs = [t.squeeze(1) for t in b]
        ec = orc.custom_forward(W, cfg, ci, cm, cp, cs)
        er = orc.custom_forward(W, cfg, ri, rm, rp, rs)
        l, p = _pairwise_loss(tau)(ec, er)
        exp_loss.append(float(l)); exp_prob.append(float(p))
    assert abs(stats["eval_loss_mean"] - np.mean(exp_loss)) < 2e-3 and abs(stats["prob_mean"] - np.mean(exp_prob)) < 1e-3
    # GPM head in train mode: [B, d]
    cfg2 = synth.tiny_config(is_general_preference=True, value_head_dim=2)
    m2 = RewardModel(cfg2, synth_seed=seed, max_batch=4, max_seq=1024, max_crops=5, operand_dtype="f16x2").to("cuda").train()
    g, _ = m2.custom_forward(c_ids.cuda(), c_mask.cuda(), c_pix.cuda(), c_sz)
    W2 = orc.weights_to_torch(synth.make_weights(cfg2, seed))
    assert g.shape == (3, 2) and (g.cpu() - orc.custom_forward(W2, cfg2, c_ids, c_mask, c_pix, c_sz, training=True)).abs().max().item() < 1e-4


def test_ragged_captions_collated_on_the_device(tmp_path):
    """Three (image, caption) items with captions of different length -> ONE custom_forward over a left-padded [3, S] batch built
    without leaving the GPU.  Checked against the oracle on the same collated batch (pixels from the HD-transform oracle), and the
    layout against the reference's collate rules: ids padded on the left with the tokenizer's pad id, masks with 0."""
    from PIL import Image
    cfg = synth.tiny_config(is_general_preference=True, value_head_dim=2)
    seed = 67
    tok = synth.StandInTokenizer()
    caps = ["a cat", "a considerably longer caption about a dog on a skateboard", "tree"]
    items, imgs = [], []
    for i, (hw, cap) in enumerate(zip([(200, 300), (336, 336), (300, 200)], caps)):
        a = synth.synth_image(seed, f"collate.{i}", hw[0], hw[1], True)
        p = str(tmp_path / f"im{i}.png")
        Image.fromarray(a).save(p)
        items.append((p, cap)); imgs.append(a)
    batch = P.batch_inference_process_phi3v_device(None, tok, items, device="cuda", num_crops=4, pad_token_id=cfg.vocab_size - 1)
    ids, mask = batch["input_ids"], batch["attention_mask"]
    assert ids.is_cuda and ids.dim() == 2 and ids.shape == mask.shape and batch["pixel_values"].shape == (3, 5, 3, 336, 336)
    lens = mask.sum(dim=1).tolist()
    S = ids.shape[1]
    assert max(lens) == S and len(set(lens)) > 1                      # ragged; the longest row un-padded
    for b, n in enumerate(lens):
        assert mask[b, : S - n].sum() == 0 and mask[b, S - n:].all() and (ids[b, : S - n] == cfg.vocab_size - 1).all()
    m = RewardModel(cfg, synth_seed=seed, max_batch=4, max_seq=1024, max_crops=5, operand_dtype="f16x2").to("cuda").eval()
    got, _ = m.custom_forward(**batch)
    W = orc.weights_to_torch(synth.make_weights(cfg, seed))
    pix = np.stack([HD.preprocess(a, 4)[0] for a in imgs])
    ref = orc.custom_forward(W, cfg, ids.cpu(), mask.cpu(), pix, batch["image_sizes"])
    err = (got.cpu() - ref).abs().max().item()
    print(f"[device collate] max |reward err| = {err:.2e}")
    assert err < 1e-4
    # the rows, scored one by one from inference_process_phi3v_device (B = 1, no padding), agree with their batched rewards:
    # all three images give the same V here (V_max does not change), only the padding and the position of the row differ
    for b, (p, cap) in enumerate(items):
        row = P.inference_process_phi3v_device(None, tok, [p], cap, num_crops=4)[0]
        one, _ = m.custom_forward(**row)
        assert (one[0] - got[b]).abs().max().item() < 2e-5


def test_score_candidates_in_memory_images():
    """Reward-guided sampling (the Fk-steering use of the reference's README): a population of candidate images for
    one prompt, decoded pixels already on the GPU -> rewards [N, d] in input order.  Equal, bit for bit, to scoring the same images
    from files through batch_inference_process_phi3v_device + custom_forward, whatever the chunking (same-sized candidates share
    V_max, so the reference's un-masked SkipCA padding does not come into it), and to the oracle within the f16x2 bound."""
    import tempfile
    from PIL import Image
    from llava_reward_amd.scoring import score_candidates
    cfg = synth.tiny_config(is_general_preference=True, value_head_dim=2)
    seed = 73
    tok = synth.StandInTokenizer()
    pad = cfg.vocab_size - 1
    m = RewardModel(cfg, synth_seed=seed, max_batch=8, max_seq=1024, max_crops=5, operand_dtype="f16x2").to("cuda").eval()
    arrs = [synth.synth_image(seed, f"cand.{i}", 300, 200, True) for i in range(5)]
    cands = [torch.from_numpy(a).cuda() for a in arrs]
    prompt = "an astronaut riding a horse"
    got = score_candidates(m, tok, prompt, cands, num_crops=4, batch_size=32, pad_token_id=pad)
    assert got.shape == (5, 2) and got.is_cuda and got.dtype == torch.float32
    for bs in (2, 1):
        assert torch.equal(score_candidates(m, tok, prompt, cands, num_crops=4, batch_size=bs, pad_token_id=pad), got)
    # numpy arrays, PIL images and files are the same request
    assert torch.equal(score_candidates(m, tok, [prompt] * 5, [Image.fromarray(a) for a in arrs[:3]] + arrs[3:], num_crops=4, pad_token_id=pad), got)
    with tempfile.TemporaryDirectory() as d:
        paths = []
        for i, a in enumerate(arrs):
            paths.append(f"{d}/c{i}.png")
            Image.fromarray(a).save(paths[-1])
        batch = P.batch_inference_process_phi3v_device(None, tok, [(p, prompt) for p in paths], device="cuda", num_crops=4, pad_token_id=pad)
    direct, _ = m.custom_forward(**batch)
    assert torch.equal(direct, got)
    W = orc.weights_to_torch(synth.make_weights(cfg, seed))
    pix = np.stack([HD.preprocess(a, 4)[0] for a in arrs])
    ref = orc.custom_forward(W, cfg, batch["input_ids"].cpu(), batch["attention_mask"].cpu(), pix, batch["image_sizes"])
    assert (got.cpu() - ref).abs().max().item() < 1e-4
    # train mode is left as it was found, and the rewards are the inference (EOS position) ones either way
    m.train()
    assert torch.equal(score_candidates(m, tok, prompt, cands, num_crops=4, pad_token_id=pad), got) and m.training
    with pytest.raises(ValueError):
        score_candidates(m, tok, [prompt] * 4, cands, num_crops=4, pad_token_id=pad)


def test_outputs_hidden_states_on_demand_and_reward_dtype():
    """`outputs["hidden_states"]` (the reference returns the backbone's whole output object, rw_model_general_preference.py:346-353,
    :422-425) in the reference's layout: layers + 2 entries for phi3v -- element 0 the embeddings, element k the stream entering layer k
    (what layer_id = k reads), element `layers` the final norm (= last_hidden_state), the LAST one `vision_embeds` [B, V_max, hidden]
    (modeling_phi3_v.py:1505) -- against the oracle's taps; inner elements are recomputed on demand and cached; asking for them leaves the
    model as it was; after a weight change an access raises.  reward_dtype=torch.bfloat16: the reference's GPU return dtype, the fp32
    reward rounded once."""
    cfg = synth.tiny_config(layers=3)
    seed = 67
    m = RewardModel(cfg, synth_seed=seed, max_batch=4, max_seq=1024, max_crops=5, operand_dtype="f16x2").to("cuda").eval()
    W = orc.weights_to_torch(synth.make_weights(cfg, seed))
    b = synth.synth_batch(cfg, seed, [5, 8, 3], [(1, 1), (1, 2), (1, 1)], max_crops=3)          # ragged crop grids: V_max zero padding
    tb = {k: torch.from_numpy(v) for k, v in b.items()}
    args = (tb["input_ids"].cuda(), tb["attention_mask"].cuda(), tb["pixel_values"].cuda(), tb["image_sizes"])
    r0, outs = m.custom_forward(*args, return_output=True)
    r0 = r0.clone()
    hs = outs["hidden_states"]
    assert len(hs) == cfg.layers + 2
    taps = {}
    orc.custom_forward(W, cfg, tb["input_ids"], tb["attention_mask"], tb["pixel_values"], tb["image_sizes"], taps=taps)
    valid = tb["attention_mask"].bool()
    want = [taps["embeds"]] + [taps[f"layer{k}"] for k in range(cfg.layers - 1)] + [orc.rms_norm(taps[f"layer{cfg.layers - 1}"], W["model.norm.weight"], cfg.rms_eps)]
    for k in range(cfg.layers + 1):
        got = hs[k].cpu()
        assert got.shape == want[k].shape
        assert (got - want[k])[valid].abs().max().item() < 2e-4 * want[k][valid].abs().max().item() + 1e-5, k
    assert hs[1] is hs[1]                                              # cached: one extra forward per inner element, not one per access
    assert torch.equal(hs[cfg.layers], outs["last_hidden_state"]) and torch.equal(hs[-2], outs["last_hidden_state"]) and len(list(hs[1:3])) == 2
    # the last entry: the oracle's projected image tokens, zero-padded per sample to V_max (modeling_phi3_v.py:242-245)
    counts = (tb["input_ids"] < 0).sum(dim=1).tolist()
    ve = hs[-1].cpu()
    assert ve.shape == (3, max(counts), cfg.hidden) and len(set(counts)) > 1
    off = 0
    for i, n in enumerate(counts):
        assert (ve[i, :n] - taps["proj"][off:off + n]).abs().max().item() < 2e-4 * taps["proj"].abs().max().item() + 1e-5
        assert not ve[i, n:].any()
        off += n
    # what rw_model:353 evaluates on the reference's object works on this one and gives that tensor
    vision_embedding = outs["hidden_states"][-1][:, :outs["hidden_states"][-1].shape[1], :]
    assert torch.equal(vision_embedding, hs[-1])
    with pytest.raises(IndexError):
        hs[len(hs)]
    # element k is what a model built with layer_id = k reads its reward from, bit for bit
    mk = RewardModel(cfg, synth_seed=seed, max_batch=4, max_seq=1024, max_crops=5, operand_dtype="f16x2", layer_id=1).to("cuda").eval()
    _, ok = mk.custom_forward(*args, return_output=True)
    assert torch.equal(ok["hidden_states_at_layer_id"], hs[1])
    with pytest.raises(KeyError):
        ok["hidden_states"]
    # the model is as it was: same rewards
    assert torch.equal(m.custom_forward(*args)[0], r0)
    # other weights behind the same object: an element that is not cached yet refuses (it would be another model's state)
    _, outs2 = m.custom_forward(*args, return_output=True)
    m.engine.synth_weights(seed + 1, False, 0)
    with pytest.raises(RuntimeError, match="weights changed"):
        outs2["hidden_states"][2]
    assert outs2["hidden_states"][-1] is not None                     # (taken at the forward itself)
    mb = RewardModel(cfg, synth_seed=seed, max_batch=4, max_seq=1024, max_crops=5, operand_dtype="f16x2", reward_dtype=torch.bfloat16).to("cuda").eval()
    rb = mb.custom_forward(*args)[0]
    assert rb.dtype == torch.bfloat16 and torch.equal(rb, r0.to(torch.bfloat16))
    with pytest.raises(NotImplementedError, match="vision_layer_id"):
        RewardModel(cfg, synth_seed=seed, vision_layer_id=3)


def _fingerprint_err(fp, t):
    """|t.flatten()[idx] - vals| / (1 + |vals|) of a make_goldens.py fingerprint (taken of the REFERENCE's tensor)."""
    assert list(t.shape) == fp["shape"], (t.shape, fp["shape"])
    a = t.reshape(-1)[torch.tensor(fp["idx"])].cpu()
    v = torch.tensor(fp["vals"])
    return ((a - v).abs() / (1.0 + v.abs())).max().item()


@pytest.mark.parametrize("name", ["ref_small_bt_ca", "ref_small_bt_ca_ragged", "ref_full_b2_ragged_bt_ca"])
def test_outputs_hidden_states_match_the_reference_held_fingerprints(name):
    """The tuple a drop-in caller indexes, against what the REFERENCE's own tuple held (make_goldens.py fingerprints hs[0] = embeds,
    hs[-1] = vision_embeds, outputs['last_hidden_state'] = final_norm and hs[l + 1] = layerL of the reference's run): same length
    (layers + 2), same shapes -- [B, V_max, hidden] for the last entry, zero-padded rows included in the full-size ragged B = 2 case --
    and the sampled values within 5e-5 (2e-4 at full depth), relative to 1 + |value|."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", name + ".json")))
    cfg = synth.RewardConfig.from_json(g["config"])
    grids = g["grids"]
    grids = tuple(grids) if isinstance(grids[0], int) else [tuple(x) for x in grids]
    b = synth.synth_batch(cfg, g["seed"], g["caption_lens"], grids, max_crops=g["max_crops"])
    B, S = b["input_ids"].shape
    full = cfg.layers > 4
    m = RewardModel(cfg, synth_seed=g["seed"], max_batch=B, max_seq=max(S, 1024), max_crops=17 if full else 5, operand_dtype="f16x2").to("cuda").eval()
    tb = {k: torch.from_numpy(v) for k, v in b.items()}
    _, outs = m.custom_forward(tb["input_ids"].cuda(), tb["attention_mask"].cuda(), tb["pixel_values"].cuda(), tb["image_sizes"], return_output=True)
    hs = outs["hidden_states"]
    assert len(hs) == cfg.layers + 2
    tol = 2e-4 if full else 5e-5
    t = g["taps"]
    errs = {"vision_embeds": _fingerprint_err(t["vision_embeds"], hs[-1]), "final_norm": _fingerprint_err(t["final_norm"], hs[cfg.layers]),
            "embeds": _fingerprint_err(t["embeds"], hs[0])}
    # (fingerprints sample padded token rows too, where the states are don't-care values on both sides: valid rows only)
    valid = tb["attention_mask"].bool().reshape(-1)

    def on_valid(fp, ten):
        idx = torch.tensor(fp["idx"])
        keep = valid[idx // cfg.hidden]
        a = ten.reshape(-1)[idx[keep]].cpu()
        v = torch.tensor(fp["vals"])[keep]
        return ((a - v).abs() / (1.0 + v.abs())).max().item()
    errs["final_norm"] = on_valid(t["final_norm"], hs[cfg.layers])
    errs["embeds"] = on_valid(t["embeds"], hs[0])
    l = min(int(k[5:]) for k in t if k.startswith("layer"))
    errs[f"layer{l}"] = on_valid(t[f"layer{l}"], hs[l + 1])
    print(f"[{name}] hidden_states vs the reference's fingerprints: {errs}")
    assert max(errs.values()) < tol, errs


def test_outputs_hidden_states_llava_layout():
    """llava branch (rw_model:372-375, LlavaNext's output object): layers + 1 entries, the last one the final norm (no vision_embeds
    entry), inner elements against the oracle's taps."""
    from oracle import llava_next_reward_oracle as lorc
    cfg = synth.llava_tiny_config()
    seed = 23
    batch = synth.llava_synth_batch(cfg, seed, [7, 3], [(336, 336), (512, 640)], max_crops=5)
    W = orc.weights_to_torch(synth.llava_make_weights(cfg, seed))
    taps = {}
    lorc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"], taps=taps)
    m = RewardModel(cfg, synth_seed=seed, max_batch=3, max_seq=4096, max_crops=5, operand_dtype="f16x2").to("cuda").eval()
    tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    _, outs = m.custom_forward(inputs_batch=tb, return_output=True)
    hs = outs["hidden_states"]
    assert len(hs) == cfg.layers + 1 and torch.equal(hs[-1], outs["last_hidden_state"])
    valid = torch.from_numpy(batch["attention_mask"]).bool()
    want = [taps["embeds"]] + [taps[f"layer{k}"] for k in range(cfg.layers - 1)]
    for k, w in enumerate(want):
        got = hs[k].cpu()
        assert got.shape == w.shape and (got - w)[valid].abs().max().item() < 2e-4 * w[valid].abs().max().item() + 1e-5, k
