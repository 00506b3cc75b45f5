"""End-to-end parity of the HIP scoring path (through the C ABI) on a real MI355X.

Checkers: (1) the CPU oracle on the same seeded inputs (small configs); (2) goldens produced by the
reference itself (tests/golden/ref_*.json), including the FULL-SIZE Phi-3.5-V case, for which the
weights are regenerated in HBM by the same integer hash the golden script used.

Tolerance (north star): |reward - reference fp32 CPU reward| <= 1e-3.  The default, split-operand mode
("f16x2": activations as f16 hi + lo, bf16-valued weights exact in f16) is held to 1e-4 everywhere,
full-size rows included (measured <= 1.5e-5).  The single-pass fast modes are noise-limited: "f16"
meets 1e-3 on the small configs and lands within 1e-3 in the assert_close sense (atol = rtol) on the
full-size rows; "bf16" (the reference's own GPU dtype) is held to 8e-3.
Preference ordering / batch invariance: bit-exact in every mode."""
import glob
import json
import os

import numpy as np
import pytest
from conftest import record_locked_form
import torch

from llava_reward_amd import synth
from llava_reward_amd.model import RewardModel
from llava_reward_amd import probe as probe_mod
from llava_reward_amd.reward_adaptor_loader import preference_compute
from oracle import phi3v_reward_oracle as orc

pytestmark = pytest.mark.gpu

TOL_F16 = 1e-3
TOL_X2 = 1e-4
TOL_X8 = 3e-4          # "f16x2f8": residual pass of the deep-pipelined GEMMs in e4m3 (measured <= 8e-5 on the full-size rows)
TOL_BF16 = 8e-3
# Outlier-bearing full-size rows (synth.PROFILE_OUTLIER: |reward| up to 3.8, every rounding amplified 15-25x).  Round 6 settled whose
# error the distance to the reference is there: tests/golden/fp64_full_rows.json holds the oracle's DOUBLE-precision reward of these rows
# (make_fp64_fixture.py), and on the GPM row the REFERENCE's own fp32 arithmetic sits 2.1e-4 from it (|reference_fp32 - fp64|; the fp32
# oracle, another draw of the same noise: 3.1e-5), while the HIP strict form sits 1.5e-5 .. 2.3e-5 from the fp64 value (two GEMM tile shapes; 1.0e-4 before the fused RoPE took the reference's rounding points,
# profiles/r6_outlier_fp64.log).  So |hip - reference| <= 3e-4 on these rows is mostly the reference's noise (2.1e-4) plus ours (<= 2.3e-5),
# and the sharper statement is the one against fp64: the strict form within TOL_FP64 of it.  (Round 5 had widened the bound to 8e-4
# after the attention kernels' lazy softmax maximum moved the strict form to 5.2e-4 from the reference / 3.1e-4 from fp64: the strict
# stages run the exact maximum again -- AttnParams::lazy_t = 0 -- and the 3e-4 bound is back.)
TOL_OUTLIER = 3e-4
TOL_FP64 = 1.5e-4      # |hip - fp64 oracle| on the rows of fp64_full_rows.json, strict form and the form .to('cuda') locks
FP64_ROWS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fp64_full_rows.json")))
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _model(cfg, seed, dtype, upload, max_batch=4, max_seq=1024, max_crops=5, layer_id=32, mean=False, profile=0, keep=False, **kw):
    """The reference's sequence and nothing else: build -> .to('cuda') -> .eval() (eval/simple_inference.py:16-18).  kw: calibrate=False
    pins the default operand form for tests of its kernels; parity_budget for tests of the automatic form check."""
    if upload:
        W = {k: torch.from_numpy(v) for k, v in synth.make_weights(cfg, seed, profile).items()}
        m = RewardModel(cfg, weights=W, max_batch=max_batch, max_seq=max_seq, max_crops=max_crops, operand_dtype=dtype, layer_id=layer_id,
                        mean_hidden_state=mean, **kw)
    else:
        m = RewardModel(cfg, synth_seed=seed, max_batch=max_batch, max_seq=max_seq, max_crops=max_crops, operand_dtype=dtype,
                        layer_id=layer_id, mean_hidden_state=mean, synth_profile=profile, **kw)
    m.keep_hidden_states = keep          # True: the "x" tap is read afterwards (by default the last layer computes the reward rows only)
    return m.to("cuda").eval()


def _fwd(m, batch, rows=None):
    tb = {k: torch.from_numpy(v if rows is None else v[rows]) for k, v in batch.items()}
    r, _ = m.custom_forward(tb["input_ids"].cuda(), tb["attention_mask"].cuda(), tb["pixel_values"].cuda(),
                            tb["image_sizes"].cuda())
    torch.cuda.synchronize()
    return r.cpu()


@pytest.mark.parametrize("dtype,tol", [("f16x2", TOL_X2), ("f16", TOL_F16), ("bf16", TOL_BF16)])
@pytest.mark.parametrize("variant", ["bt_ca", "gpm2_ca", "bt_noca"])
def test_tiny_vs_oracle(dtype, tol, variant):
    kw = dict(bt_ca={}, gpm2_ca=dict(is_general_preference=True, value_head_dim=2), bt_noca=dict(add_cross_attention=False))[variant]
    cfg = synth.tiny_config(**kw)
    seed = 11
    batch = synth.synth_batch(cfg, seed, [7, 3, 5], [(1, 1), (1, 2), (2, 1)], max_crops=4)
    W = orc.weights_to_torch(synth.make_weights(cfg, seed))
    ref = orc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
    m = _model(cfg, seed, dtype, upload=True)
    got = _fwd(m, batch)
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    print(f"[tiny {variant} {dtype}] max |reward err| = {err:.3e}  rewards={got.flatten().tolist()}")
    assert err < tol
    # against the oracle run with the same operand rounding the kernels apply: only summation order,
    # exp2/rsqrt implementations and the point of rounding differ
    if dtype != "f16x2":
        emu = orc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"],
                                 opr=orc.f16_round if dtype == "f16" else orc.bf16_round)
        # (rounding noise of the single-pass modes, not a parity bar -- that is the line above.  bf16: measured 2.0e-3 .. 3.3e-3 across
        #  builds; f16: < 5e-4 until round 6, 5.4e-4 .. 5.9e-4 across the round-6 builds that moved the fused RoPE's rounding points --
        #  the last of them to the reference's own arithmetic, common.h rope_pair, with which the distance to the fp32 oracle above
        #  FELL from 3.4e-4 to 2.9e-4)
        assert (got - emu).abs().max().item() < (7e-4 if dtype == "f16" else 4e-3)
    # device-side synthetic weights == uploaded numpy weights, bit for bit
    m2 = _model(cfg, seed, dtype, upload=False)
    assert torch.equal(_fwd(m2, batch), got)


@pytest.mark.parametrize("profile", [synth.PROFILE_OUTLIER, synth.PROFILE_E4M3])
def test_weight_profiles_device_equals_numpy(profile):
    """The outlier-bearing and e4m3-valued synthetic weight sets (synth.PROFILE_*) are generated in HBM by synth_profile_kernel and
    on the host by numpy: same bits (a single differing weight bit would move the reward), and the engine matches the oracle on them."""
    cfg = synth.tiny_config(is_general_preference=True, value_head_dim=2)
    seed = 13
    batch = synth.synth_batch(cfg, seed, [7, 3, 5], [(1, 1), (1, 2), (2, 1)], max_crops=4)
    Wn = synth.make_weights(cfg, seed, profile)
    if profile & synth.PROFILE_E4M3:          # matrices sit on the e4m3 grid of their tensor's power-of-two scale
        w = Wn["model.layers.0.mlp.down_proj.weight"]
        q = torch.from_numpy(w * np.float32(2.0 ** -synth.e4m3_tensor_exponent(0.02)))
        assert torch.equal(q.to(torch.float8_e4m3fn).float(), q)
    if profile & synth.PROFILE_OUTLIER:
        ch = synth.outlier_channels(seed, cfg.hidden)
        emb = np.abs(Wn["model.embed_tokens.weight"])
        assert emb[:, ch].mean() > 50 * np.delete(emb, ch, axis=1).mean() and Wn["model.norm.weight"].max() > 2.0
    ref = orc.custom_forward(orc.weights_to_torch(Wn), cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
    up = _fwd(_model(cfg, seed, "f16x2", upload=True, profile=profile), batch)
    dev = _fwd(_model(cfg, seed, "f16x2", upload=False, profile=profile), batch)
    assert torch.equal(up, dev)
    err = (dev - ref).abs().max().item()
    print(f"[weight profile {profile}] max |reward err| vs oracle = {err:.2e}")
    assert err < TOL_X2


def test_operand_form_is_locked_on_the_weights_by_to_cuda():
    """The drop-in sequence -- build, .to('cuda'), custom_forward, nothing else (eval/simple_inference.py:16-31) -- locks the operand
    form of the default parity mode by itself: .to('cuda') scores the seeded probe rows (probe.py: a function of the weights and the
    engine's capacity alone) in the default and in the strict form and keeps the strict one when they differ by more than the budget.
    (1) benign and outlier-bearing tiny weight sets: whatever form was locked, the rewards sit on the oracle; (2) a budget the default
    form cannot meet: strict, on the oracle to 1e-4; (3) the decision is static (bit-stable rewards, a row scored alone equals the row
    in its batch), survives .to() again, is re-taken when a weight is re-uploaded, is skipped with calibrate=False, and a failing
    calibrate() call leaves the engine in the form it was in."""
    cfg = synth.tiny_config(hidden=1024, intermediate=2048, heads=16, layers=3)
    seed = 19
    batch = synth.synth_batch(cfg, seed, [7, 3, 5], (1, 1))
    kw = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    for profile in (synth.PROFILE_OUTLIER, 0):
        W = orc.weights_to_torch(synth.make_weights(cfg, seed, profile))
        ref = orc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
        m = _model(cfg, seed, "f16x2f8", upload=False, profile=profile)
        info = m.form_info
        got = _fwd(m, batch)
        print(f"[form probe, profile {profile}] {info}; err {(got - ref).abs().max().item():.2e}")
        assert info["source"] == "probe" and info["rows"] == probe_mod.PROBE_ROWS == 8 and info["seconds"] > 0 and info["form"] == m.operand_form and info["default_vs_strict"] < 1.0
        assert (got - ref).abs().max().item() < TOL_X8
        if profile == 0:
            assert m.operand_form == "default" and info["default_vs_strict"] < m.parity_budget
        assert torch.equal(_fwd(m, batch), got)
        for b in range(3):
            assert torch.equal(_fwd(m, batch, rows=slice(b, b + 1))[0], got[b])
        assert m.to("cuda") is m and m.form_info is info                          # already there: nothing re-done
        # (2) a budget the default form cannot meet
        ms = _model(cfg, seed, "f16x2f8", upload=False, profile=profile, parity_budget=1e-9)
        strict = _fwd(ms, batch)
        assert ms.operand_form == "strict" == ms.form_info["form"] and (strict - ref).abs().max().item() < TOL_X2
        assert torch.equal(_fwd(ms, batch, rows=slice(1, 2))[0], strict[1])
        # the strict form of a default-mode handle IS the f16x2 mode's arithmetic
        assert torch.equal(_fwd(_model(cfg, seed, "f16x2", upload=False, profile=profile), batch), strict)
        # a failing calibrate() (a batch the wrapper rejects) leaves form and engine as they were
        bad = dict(kw, input_ids=kw["input_ids"].clone())
        bad["input_ids"][0, int((bad["input_ids"][0] < 0).nonzero()[0])] = 7             # one image slot fewer than its image needs
        with pytest.raises(RuntimeError):
            ms.calibrate(bad)
        assert ms.operand_form == "strict" and torch.equal(_fwd(ms, batch), strict)
        # calibrate() on the caller's own batches overrides the probe's choice either way
        assert ms.calibrate(kw, parity_budget=1.0)["form"] == "default" == ms.operand_form and ms.form_info["source"] == "calibrate"
        assert torch.equal(_fwd(ms, batch), _fwd(_model(cfg, seed, "f16x2f8", upload=False, profile=profile, calibrate=False), batch))
        if profile == 0:
            # (3) a re-uploaded weight: the next forward takes the decision again on the new weights
            e0 = m._form_epoch
            name = "model.layers.1.mlp.down_proj.weight"
            m.engine.upload(name, torch.from_numpy(synth.make_weights(cfg, seed, profile)[name]))
            assert m.engine.weights_epoch() != e0
            again = _fwd(m, batch)
            assert m._form_epoch == m.engine.weights_epoch() and m.form_info is not info and m.form_info["source"] == "probe"
            assert torch.equal(again, got)
    off = _model(cfg, seed, "f16x2f8", upload=False, calibrate=False)
    assert off.form_info is None and off.operand_form == "default"


def test_reward_dtype_does_not_reach_the_operand_form_probe():
    """Advisor (round 5): with reward_dtype=torch.bfloat16 the probe compared bf16-rounded rewards -- an ulp of 4e-3 .. 1.6e-2 against a
    budget of 1.5e-4, so a form 5e-4 from strict read as 0 and was locked.  The rounding is applied to the caller's rewards only: both
    models measure the same distances and lock the same form, on benign and on outlier-bearing weights."""
    cfg = synth.tiny_config(hidden=1024, intermediate=2048, heads=16, layers=3)
    seed = 37
    batch = synth.synth_batch(cfg, seed, [7, 3, 5], (1, 1))
    for profile in (0, synth.PROFILE_OUTLIER):
        a = _model(cfg, seed, "f16x2f8", upload=False, profile=profile)
        b = _model(cfg, seed, "f16x2f8", upload=False, profile=profile, reward_dtype=torch.bfloat16)
        assert a.form_info["distance_to_strict"] == b.form_info["distance_to_strict"] and a.operand_form == b.operand_form
        assert a.form_info["strict_noise_floor"] == b.form_info["strict_noise_floor"]
        ra, rb = _fwd(a, batch), _fwd(b, batch)
        assert rb.dtype == torch.bfloat16 and torch.equal(rb, ra.to(torch.bfloat16))
        info = b.calibrate({k: torch.from_numpy(v).cuda() for k, v in batch.items()})          # the explicit check on caller batches: fp32 inside as well
        assert all(d == d and (d == 0.0 or d > 1e-9) for d in info["distance_to_strict"].values())


def test_precision_sites_are_independent_and_consistent():
    """lr_set_precision_sites (round 6): the operand form of each SITE of the decoder layers the precision map covers -- qkv (input norm +
    projection), attention (exact / lazy softmax maximum), o_proj, gate_up (post-attention norm + projection), down -- on top of the
    per-layer form.  All sites strict IS the strict form, all sites default IS strict-vision (bit for bit: a site's form does not leak
    into its neighbours' operands); every single-site mix is a valid model (on the oracle) and really changes the arithmetic of that
    site; with un-merged adapters too; the gathered last layer and the kept-hidden-states pass agree; bad values are refused."""
    from llava_reward_amd._lib import HipError
    cfg = synth.tiny_config(hidden=1024, intermediate=2048, heads=16, layers=4)
    seed = 31
    batch = synth.synth_batch(cfg, seed, [7, 3, 5], [(1, 1), (1, 2), (1, 1)], max_crops=3)
    W = orc.weights_to_torch(synth.make_weights(cfg, seed, synth.PROFILE_OUTLIER))
    ref = orc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
    m = _model(cfg, seed, "f16x2f8", upload=False, profile=synth.PROFILE_OUTLIER, calibrate=False)
    eng = m.engine

    def fwd(cmap, sites=()):
        eng.set_precision_map(*cmap)
        eng.set_precision_sites(*sites)
        return _fwd(m, batch)
    strict, sv = fwd((1, 1, 0, 0)), fwd((1, -1, 0, 0))
    assert torch.equal(strict, _fwd(_model(cfg, seed, "f16x2", upload=False, profile=synth.PROFILE_OUTLIER), batch))
    assert torch.equal(fwd((1, 1, 0, 0), (1, 1, 1, 1, 1)), strict)
    assert torch.equal(fwd((1, 1, 0, 0), (2, 2, 2, 2, 2)), sv)                 # every decoder site back in the default form: strict-vision
    assert torch.equal(fwd((1, -1, 0, 0), (1, 1, 1, 1, 1)), sv)                # no strict layer range: the sites have nothing to refine
    seen = {tuple(strict.flatten().tolist()), tuple(sv.flatten().tolist())}
    for i, name in enumerate(("qkv", "attention", "o_proj", "gate_up", "down")):
        one = fwd((1, 1, 0, 0), tuple(2 if j == i else 1 for j in range(5)))
        err = (one - ref).abs().max().item()
        print(f"[sites] {name} default, the others strict: |reward - oracle| = {err:.2e}  |. - strict| = {(one - strict).abs().max().item():.2e}")
        assert torch.isfinite(one).all() and err < TOL_X8
        if name != "attention":                                                # (S < 64 keys here: the lazy maximum never differs from the exact one)
            assert tuple(one.flatten().tolist()) not in seen                   # that site's arithmetic really changed, and differently from every other mix
        seen.add(tuple(one.flatten().tolist()))
        m.keep_hidden_states = True                                            # the gathered last layer and the full one walk the same sites
        assert torch.equal(_fwd(m, batch), one)
        m.keep_hidden_states = False
    # a layer range: sites apply inside it only
    part = fwd((1, 1, 0, 2), (1, 2, 2, 1, 2))
    assert torch.isfinite(part).all() and (part - ref).abs().max().item() < TOL_X8 and not torch.equal(part, fwd((1, 1, 0, 2)))
    for bad in ((0, -1, -1, -1, -1), (3, 1, 1, 1, 1), (-2, 1, 1, 1, 1)):
        with pytest.raises(HipError):
            eng.set_precision_sites(*bad)
    with pytest.raises(HipError):                                              # a strict-only handle has no e4m3 form to give a site
        _model(cfg, seed, "f16x2", upload=False).engine.set_precision_sites(2, 1, 1, 1, 1)
    eng.set_precision_sites()
    assert torch.equal(fwd((1, 1, 0, 0)), strict)


def test_probe_rows_do_not_depend_on_the_engine_capacity():
    """Round 6: the probe rows are fixed tiers (probe.py) -- an engine scores the largest tier that fits it whole and reshapes nothing --
    so two engines of different capacity (batch 1 / 6, other max_seq and max_crops, hence other chunkings of the same fixed batch) on ONE
    weight set measure the same distances, bit for bit, and lock the same form; an engine too small for that tier takes the next one."""
    cfg = synth.tiny_config(hidden=1024, intermediate=2048, heads=16, layers=3)
    seed = 29
    for profile in (0, synth.PROFILE_OUTLIER):
        a = _model(cfg, seed, "f16x2f8", upload=False, max_batch=1, max_seq=900, max_crops=5, profile=profile)
        b = _model(cfg, seed, "f16x2f8", upload=False, max_batch=6, max_seq=1500, max_crops=6, profile=profile)
        print(f"[probe capacity, profile {profile}] {a.form_info['distance_to_strict']} | {b.form_info['distance_to_strict']}")
        assert a.form_info["rows"] == b.form_info["rows"] == 8
        assert a.form_info["distance_to_strict"] == b.form_info["distance_to_strict"] and a.operand_form == b.operand_form
        small = _model(cfg, seed, "f16x2f8", upload=False, max_batch=2, max_seq=512, max_crops=2, profile=profile)      # only the (1, 1) tier fits
        assert small.form_info["rows"] == 8 and probe_mod.probe_batches(small)[0]["pixel_values"].shape[1] == 2
    assert probe_mod.PROBE_MIN_SEQ == {"phi3v": 2642, "llava": 3061, "qwen": 389}
    assert [len(x["input_ids"]) for x in probe_mod.probe_batches(a)] == [1] * 8 and [len(x["input_ids"]) for x in probe_mod.probe_batches(b)] == [4, 4]
    assert probe_mod.probe_batches(a)[0]["input_ids"].shape[1] == probe_mod.probe_batches(b)[0]["input_ids"].shape[1] == 757 + 5 + 128      # the (2, 2) tier


def test_operand_form_pin_failed_probe_and_deferred_input_check(monkeypatch):
    """Round 5: (1) operand_form="<name>" pins the form (no probe runs; the same form on any engine capacity), bit-equal to the form
    a probe would have had to lock; an unknown name is refused at construction.  (2) A probe that FAILS (advisor, round 4: OOM or a
    rejected batch inside the re-lock path) must not leave the engine un-checked in the cheapest form: the model falls back to the
    strict form, re-raises, and tries the probe again on the next forward.  (3) The probe is not a collective: with a process group
    initialised, .to('cuda') issues no all_reduce (the reference's .to() and forward have none).  (4) check_inputs="deferred": no host
    check; a row whose slot count mismatches comes back NaN from the engine, the other rows are untouched."""
    cfg = synth.tiny_config(hidden=1024, intermediate=2048, heads=16, layers=4)
    seed = 23
    batch = synth.synth_batch(cfg, seed, [7, 3, 5], (1, 1))
    names = [n for n, _ in RewardModel(cfg, synth_seed=seed)._form_candidates()]
    i0 = names.index("default")
    assert i0 == 0 and names[1] == "strict-vision" and names[-1] == "strict" and len(set(names)) == len(names) >= 5
    mf = RewardModel(synth.full_config(), synth_seed=seed)
    full = [n for n, _ in mf._form_candidates()]
    assert full == ["default", "strict-vision", "strict-vision+decoder/8", "strict-vision+decoder/4", "strict-vision+decoder*3/8",
                    "strict-vision+decoder/2", "strict"]
    # single-pass tails: pinnable by name, never probe candidates (a lottery at the budget's edge)
    assert [n for n, _ in mf._pinnable_forms()] == ["default+single-tail/4", "default+single-tail/8", "default+single-tail/16"]
    assert RewardModel(synth.full_config(), synth_seed=seed, operand_form="default+single-tail/8").pinned_form == "default+single-tail/8"
    with pytest.raises(ValueError):
        RewardModel(cfg, synth_seed=seed, operand_form="nope")
    strict = _fwd(_model(cfg, seed, "f16x2", upload=False), batch)
    for name in (names[i0 + 2], "strict"):
        m = _model(cfg, seed, "f16x2f8", upload=False, operand_form=name)
        assert m.operand_form == name and m.form_info["source"].startswith("pinned") and m.form_info["rows"] == 0
        got = _fwd(m, batch)
        assert torch.equal(got, strict) == (name == "strict")
        m.engine.synth_weights(seed)               # new weights behind the handle: the pin survives the re-lock
        assert torch.equal(_fwd(m, batch), got) and m.operand_form == name
    # (2) failing probe
    m = _model(cfg, seed, "f16x2f8", upload=False)
    assert m.form_info["source"] == "probe"
    calls = {"n": 0}
    real = probe_mod.probe_batches

    def boom(model, rows=probe_mod.PROBE_ROWS):
        calls["n"] += 1
        if calls["n"] == 1:
            raise RuntimeError("probe blew up")
        return real(model, rows)
    monkeypatch.setattr(probe_mod, "probe_batches", boom)
    m.engine.synth_weights(seed)
    with pytest.raises(RuntimeError, match="probe blew up"):
        _fwd(m, batch)
    assert m.operand_form == "strict" and m.form_info["source"] == "probe failed" and m._form_epoch != m.engine.weights_epoch()
    again = _fwd(m, batch)                          # the next forward retries the probe (second call succeeds)
    assert calls["n"] == 2 and m.form_info["source"] == "probe" and m._form_epoch == m.engine.weights_epoch()
    assert torch.equal(again, _fwd(_model(cfg, seed, "f16x2f8", upload=False), batch))
    monkeypatch.undo()
    # (3) no collective in the probe
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % (29800 + os.getpid() % 100), rank=0, world_size=1)
        made = True
    else:
        made = False
    try:
        def no_collective(*a, **k):
            raise AssertionError("collective inside .to('cuda') / the probe")
        monkeypatch.setattr(dist, "all_reduce", no_collective)
        monkeypatch.setattr(dist, "get_world_size", lambda *a, **k: 2)
        mp = _model(cfg, seed, "f16x2f8", upload=False)
        assert mp.form_info["source"] == "probe"
    finally:
        monkeypatch.undo()
        if made:
            dist.destroy_process_group()
    # (4) deferred input check
    md = _model(cfg, seed, "f16x2f8", upload=False, check_inputs="deferred", calibrate=False)
    me = _model(cfg, seed, "f16x2f8", upload=False, calibrate=False)
    good = _fwd(me, batch)
    assert torch.equal(_fwd(md, batch), good)
    bad = {k: v.copy() for k, v in batch.items()}
    bad["input_ids"][1, int(np.nonzero(bad["input_ids"][1] < 0)[0][0])] = 7          # row 1: one image slot fewer than its image needs
    with pytest.raises(RuntimeError):
        _fwd(me, bad)
    r = _fwd(md, bad)
    assert torch.isnan(r[1]).all() and torch.equal(r[0], good[0]) and torch.equal(r[2], good[2])
    # host-resident inputs are counted on the host even in deferred mode (free: no stream drain)
    tb = {k: torch.from_numpy(v) for k, v in bad.items()}
    with pytest.raises(RuntimeError):
        md.custom_forward(tb["input_ids"], tb["attention_mask"], tb["pixel_values"], tb["image_sizes"])


def test_stage_taps_tiny():
    """Localise divergences: CLIP output, projected vision tokens, residual stream after the stack."""
    cfg = synth.tiny_config()
    seed = 3
    batch = synth.synth_batch(cfg, seed, [4, 6], [(1, 1), (1, 1)])
    W = orc.weights_to_torch(synth.make_weights(cfg, seed))
    taps = {}
    orc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"], taps=taps)
    m = _model(cfg, seed, "f16", upload=True, keep=True)
    _fwd(m, batch)
    e = m.engine
    Hc, D, T = cfg.clip.hidden, cfg.hidden, cfg.clip.tokens
    B, S = batch["input_ids"].shape
    clip = e.read_tap("clip_x", 4 * T * Hc).reshape(4, T, Hc)[:, 1:]
    ref_clip = taps["clip_out"].reshape(4, T - 1, Hc).numpy()
    assert np.abs(clip - ref_clip).max() < 2e-2 * np.abs(ref_clip).max()
    ev = e.read_tap("ev", taps["proj"].numel()).reshape(taps["proj"].shape)
    assert np.abs(ev - taps["proj"].numpy()).max() < 2e-2 * taps["proj"].abs().max().item()
    x = e.read_tap("x", B * S * D).reshape(B, S, D)
    ref_x = taps[f"layer{cfg.layers - 1}"].numpy()
    valid = batch["attention_mask"].astype(bool)
    assert np.abs(x - ref_x)[valid].max() < 2e-2 * np.abs(ref_x[valid]).max()


@pytest.mark.parametrize("dtype", ["f16x2", "f16x2f8", "f16"])
def test_batch_invariance_and_preference_order_bit_exact(dtype):
    """A row's reward must not depend on what else is in the batch (fixed reduction order, no atomics),
    so preference ordering is bit-exact however rows are sharded across GPUs."""
    cfg = synth.tiny_config(is_general_preference=True, value_head_dim=2)
    batch = synth.synth_batch(cfg, 21, [5, 5, 5, 5], (1, 1))
    m = _model(cfg, 21, dtype, upload=False)
    full = _fwd(m, batch)
    for b in range(4):
        one = _fwd(m, batch, rows=slice(b, b + 1))
        assert torch.equal(one[0], full[b])
    two = _fwd(m, batch, rows=slice(2, 4))
    assert torch.equal(two, full[2:4])

    class A:
        is_general_preference, value_head_dim, general_preference_tau = True, 2, 0.1
    p_full = preference_compute(A, full[:2], full[2:])
    p_split = preference_compute(A, torch.cat([_fwd(m, batch, rows=slice(0, 1)), _fwd(m, batch, rows=slice(1, 2))]), two)
    assert np.array_equal(p_full, p_split) and p_full.dtype == np.float32


def test_training_flag_and_errors():
    cfg = synth.tiny_config()
    batch = synth.synth_batch(cfg, 5, [3, 6], (1, 1))
    m = _model(cfg, 5, "f16", upload=False)
    ev = _fwd(m, batch)
    m.train()
    tr = _fwd(m, batch)
    m.eval()
    # row 1 is un-padded: last position == last valid token; row 0 is left-padded and its last position is
    # also its last valid token (left padding), so both modes agree (rw_model:410-421) -- in value; the BT head returns
    # [B] in train mode (values.squeeze(-1)[:, -1], :413-415) and [B, 1] in eval mode (:420-421)
    assert ev.shape == (2, 1) and tr.shape == (2,) and torch.equal(ev.squeeze(-1), tr)
    bad = dict(batch)
    bad["input_ids"] = batch["input_ids"].copy()
    bad["input_ids"][0, -1] = -1          # one image slot too many
    with pytest.raises(RuntimeError):
        _fwd(m, bad)
    with pytest.raises(UnboundLocalError):
        m.custom_forward(torch.from_numpy(batch["input_ids"]).cuda(), torch.from_numpy(batch["attention_mask"]).cuda())


def test_sequence_longer_than_original_max_position_embeddings():
    """S = 4231 > original_max_position_embeddings = 4096 with the stock config: the su-RoPE long factors apply to the whole batch
    (the switch is on the padded length, modeling_phi3_v.py:673), position ids run past 4096 on the long row, and the attention
    kernels walk 67 key tiles; a short left-padded row rides along.  Against the oracle, strict and default parity modes."""
    cfg = synth.tiny_config()
    assert cfg.orig_max_pos == 4096
    seed = 83
    batch = synth.synth_batch(cfg, seed, [1717, 40], [(4, 4), (1, 2)], max_crops=16)
    S = batch["input_ids"].shape[1]
    assert S > cfg.orig_max_pos
    W = orc.weights_to_torch(synth.make_weights(cfg, seed))
    ref = orc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
    for dtype, tol in (("f16x2", TOL_X2), ("f16x2f8", TOL_X8)):
        m = _model(cfg, seed, dtype, upload=False, max_batch=2, max_seq=S, max_crops=17)
        got = _fwd(m, batch)
        err = (got - ref).abs().max().item()
        print(f"[S={S} > 4096, {dtype}] max |reward err| = {err:.2e}")
        assert err < tol
    # the long factors really are in effect: with the short factors the long row's reward is elsewhere
    import dataclasses
    ref_short = orc.custom_forward(W, dataclasses.replace(cfg, long_factor=cfg.short_factor), batch["input_ids"], batch["attention_mask"],
                                   batch["pixel_values"], batch["image_sizes"])
    assert (ref_short - ref).abs().max().item() > 10 * TOL_X8


def test_c_abi_caller_with_wrong_slot_count_gets_nan_not_garbage():
    """lr_forward cannot compare a row's image-slot count with the image tokens its image_sizes produce without a device sync (the
    Python wrapper does, and raises as the reference would: test_training_flag_and_errors).  A direct C-ABI caller gets NaN for
    such a row -- surplus slots never index another row's features or run past the buffer -- and correct rewards for the others."""
    cfg = synth.tiny_config()
    batch = synth.synth_batch(cfg, 5, [3, 6, 4], (1, 1))
    m = _model(cfg, 5, "f16x2", upload=False)
    good = _fwd(m, batch)
    for delta in (+1, -1):
        ids = batch["input_ids"].copy()
        row = 1
        if delta > 0:
            ids[row, -2] = -1                                   # one slot too many (a caption token turned into an image slot)
        else:
            first = int(np.argmax(ids[row] < 0))
            ids[row, first] = 7                                 # one slot too few
        tb = {k: torch.from_numpy(v) for k, v in dict(batch, input_ids=ids).items()}
        r = m.engine.forward(tb["input_ids"].cuda(), tb["attention_mask"].cuda(), tb["pixel_values"].cuda(), tb["image_sizes"])    # no wrapper check
        torch.cuda.synchronize()
        r = r.cpu()
        assert torch.isnan(r[row]).all() and torch.equal(r[0], good[0]) and torch.equal(r[2], good[2])


CASES = sorted(glob.glob(os.path.join(GOLD, "ref_small_*.json")))


@pytest.mark.parametrize("dtype,tol", [("f16x2", TOL_X2), ("f16x2f8", TOL_X8), ("f16", TOL_F16)])
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-5] for p in CASES])
def test_reference_goldens_small(path, dtype, tol):
    """Rewards produced by the reference itself (fp32 CPU) on full CLIP ViT-L + a 2-layer decoder."""
    g = json.load(open(path))
    cfg = synth.RewardConfig.from_json(g["config"])
    grids = g["grids"]
    grids = tuple(grids) if isinstance(grids[0], int) else [tuple(x) for x in grids]
    batch = synth.pad_left(synth.synth_batch(cfg, g["seed"], g["caption_lens"], grids, max_crops=g["max_crops"]), g.get("extra_left_pad", 0))
    if g.get("right_padded"):
        batch = synth.right_pad(batch)
    ref = torch.tensor(g["reward"], dtype=torch.float32)
    m = _model(cfg, g["seed"], dtype, upload=False, max_batch=2, max_seq=1024, max_crops=5, layer_id=g.get("layer_id", 32),
               mean=g.get("mean_hidden_state", False), profile=g.get("weight_profile", 0))
    if g.get("train"):          # model.train(): reward of the last position; the BT head returns [B] (rw_model:413-415), GPM [B, d]
        m.train()
    got = _fwd(m, batch)
    assert list(got.shape) == g.get("reward_shape", list(ref.shape))
    got = got.reshape(ref.shape)
    err = (got - ref).abs().max().item()
    print(f"[{g['name']} {dtype}] max |reward err| vs reference = {err:.3e}")
    assert err < tol
    # by default the last decoder layer runs its attention / o_proj / MLP for the reward rows only (rw_model:408-421 reads one row per
    # sample): bit-identical to the forward that keeps every token through it (eval and train selection, left and right padding)
    m.keep_hidden_states = True
    assert torch.equal(_fwd(m, batch).reshape(ref.shape), got)
    m.keep_hidden_states = False
    if g.get("layer_id", 32) != 32:      # the same engine gives the last-layer reward again once layer_id is the literal 32
        m.layer_id = 32
        W = orc.weights_to_torch(synth.make_weights(cfg, g["seed"], g.get("weight_profile", 0)))
        full = orc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"],
                                  mean_hidden_state=bool(g.get("mean_hidden_state", False)))
        assert (_fwd(m, batch).reshape(full.shape) - full).abs().max().item() < tol


TAP_CASES = [p for p in CASES if json.load(open(p)).get("taps")]
TOL_TAP = 5e-5


def _check_golden_taps(g, max_seq=1024, max_crops=5, tol=None):
    """Stage-level pins against the REFERENCE itself (not the oracle): the fingerprints make_goldens.py took of the reference's
    hidden_states -- `embeds` (hs[0]), `layerL` (hs[L + 1]: the stream leaving decoder layer L), `final_norm` (last_hidden_state),
    `vision_embeds` (hs[-1], zero-padded to V_max) -- are read back from the HIP engine through lr_read_tap in the strict parity
    mode (f16x2) and compared at 5e-5 (relative to 1 + |value|) on valid (un-padded) token rows.  The engine keeps one residual
    stream, so the stack is stopped after 0 / L + 1 / all layers (lr_set_layer_limits via layer_id) to observe each state; the final
    norm of every row is applied on the host from the engine's un-normed stream and its norm weight (the engine itself only norms
    the gathered row)."""
    cfg = synth.RewardConfig.from_json(g["config"])
    grids = g["grids"]
    grids = tuple(grids) if isinstance(grids[0], int) else [tuple(x) for x in grids]
    batch = synth.pad_left(synth.synth_batch(cfg, g["seed"], g["caption_lens"], grids, max_crops=g["max_crops"]), g.get("extra_left_pad", 0))
    B, S = batch["input_ids"].shape
    D = cfg.hidden
    prof = g.get("weight_profile", 0)
    valid = torch.from_numpy(batch["attention_mask"]).bool().reshape(-1)
    m = _model(cfg, g["seed"], "f16x2", upload=False, max_batch=B, max_seq=max(max_seq, S), max_crops=max_crops, profile=prof, keep=True)
    taps = g["taps"]

    def stream_after(n_layers):
        m.layer_id = n_layers if n_layers < cfg.layers else 32
        _fwd(m, batch)
        return torch.from_numpy(m.engine.read_tap("x", B * S * D).copy()).reshape(B * S, D)

    def check(name, mine_flat_rows, ok_rows):
        fp = taps[name]
        idx = torch.tensor(fp["idx"])
        rows, cols = idx // D, idx % D
        keep = ok_rows[rows]
        assert keep.any(), name
        a = mine_flat_rows[rows[keep], cols[keep]]
        b = torch.tensor(fp["vals"])[keep]
        err = ((a - b).abs() / (1.0 + b.abs())).max().item()          # (outlier goldens carry |x| ~ 500 in three channels)
        print(f"[{g['name']}] tap {name}: {int(keep.sum())} samples, max err {err:.2e}")
        assert err < (tol or TOL_TAP), (name, a, b)

    check("embeds", stream_after(0), valid)
    for name in sorted(taps):
        if name.startswith("layer"):
            check(name, stream_after(int(name[5:]) + 1), valid)
    x = stream_after(cfg.layers)
    w = torch.from_numpy(synth.gen_tensor(g["seed"], "model.norm.weight", (D,), 0.05, 1.0, profile=prof))
    normed = w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + cfg.rms_eps))          # modeling_phi3_v.py:377-391 on the engine's stream
    check("final_norm", normed, valid)
    # vision_embeds: [B, V_max, D] zero-padded per sample (modeling_phi3_v.py:242-245); the engine keeps the rows packed
    counts = (batch["input_ids"] < 0).sum(axis=1)
    Vmax = int(counts.max())
    ev = torch.from_numpy(m.engine.read_tap("ev", int(counts.sum()) * D).copy()).reshape(-1, D)
    padded = torch.zeros(B * Vmax, D)
    off = 0
    for b, n in enumerate(counts.tolist()):
        padded[b * Vmax: b * Vmax + n] = ev[off: off + n]
        off += n
    assert taps["vision_embeds"]["shape"] == [B, Vmax, D]
    check("vision_embeds", padded, torch.ones(B * Vmax, dtype=torch.bool))


@pytest.mark.parametrize("path", TAP_CASES, ids=[os.path.basename(p)[:-5] for p in TAP_CASES])
def test_reference_golden_stage_taps(path):
    _check_golden_taps(json.load(open(path)))


FULL = [p for p in sorted(glob.glob(os.path.join(GOLD, "ref_full_*.json"))) if "pair_sample" not in p]     # (the pair has its own test)
# every full-size golden in both parity forms; the single-pass fast mode (not a parity mode) on the first two rows only
FULL_PARAMS = [(p, d) for p in FULL for d in ("f16x2", "f16x2f8")] + [(p, "f16") for p in FULL if os.path.basename(p) in
                                                                       ("ref_full_bt_ca.json", "ref_full_gpm2_ca.json")]


@pytest.mark.parametrize("path,dtype", FULL_PARAMS, ids=[os.path.basename(p)[:-5] + "-" + d for p, d in FULL_PARAMS])
def test_reference_golden_full_size(path, dtype):
    """Full Phi-3.5-V shapes (32 layers, D=3072, up to 17 crops, V=2509, S=2643): reward of the reference's fp32 CPU custom_forward vs
    the HIP path with weights regenerated in HBM.  Rows: several seeds / caption lengths / crop grids, a B=2 batch with ragged
    captions AND ragged crop grids (left padding, V_max zero padding, SkipCA over the padded vision rows), and rows on the
    outlier-bearing weight profile (synth.PROFILE_OUTLIER: massive residual channels, large norm gains, 50-sigma and sub-normal
    weight elements)."""
    g = json.load(open(path))
    cfg = synth.RewardConfig.from_json(g["config"])
    grids = g["grids"]
    grids = tuple(grids) if isinstance(grids[0], int) else [tuple(x) for x in grids]
    batch = synth.synth_batch(cfg, g["seed"], g["caption_lens"], grids, max_crops=g["max_crops"])
    ref = torch.tensor(g["reward"], dtype=torch.float32)
    B, S = batch["input_ids"].shape
    outlier = bool(g.get("weight_profile", 0) & synth.PROFILE_OUTLIER)
    # (max_seq admits every probe tier, as bench.py's engines do: the locked form is then the one any full-size deployment locks)
    m = _model(cfg, g["seed"], dtype, upload=False, max_batch=2 * B, max_seq=max(S, probe_mod.PROBE_MIN_SEQ["phi3v"]), max_crops=17, profile=g.get("weight_profile", 0))
    got = _fwd(m, batch).reshape(ref.shape)
    err = (got - ref).abs().max().item()
    print(f"[{g['name']} {dtype}] reward hip={got.flatten().tolist()} ref={ref.flatten().tolist()} err={err:.3e}")
    f64 = FP64_ROWS.get(g["name"])
    err64 = None
    if f64 is not None and dtype != "f16":
        # beside the reference's fp32 reward: the double-precision value of the same function (ORACLE fixture, make_fp64_fixture.py)
        err64 = (got.double() - torch.tensor(f64["reward_fp64"], dtype=torch.float64).reshape(ref.shape)).abs().max().item()
        print(f"[{g['name']} {dtype}] |hip - fp64 oracle| = {err64:.3e}   (|reference_fp32 - fp64| = {f64['reference_minus_fp64']:.3e})")
        assert err64 < TOL_FP64
    if dtype == "f16x2":
        # strict parity form: measured 2.6e-6 / 5.5e-6 on benign rows.  On the outlier-bearing rows (|reward| up to 3.8, every rounding
        # amplified 15-25x) the fp32 summation order itself shows: 5e-6 (BT row), 1.07e-4 (GPM row, where the reference itself is
        # 2.1e-4 from the fp64 value and this form 2.3e-5) -- held to 3e-4 there
        assert err < (TOL_OUTLIER if outlier else TOL_X2)
    elif dtype == "f16x2f8" and outlier:
        # The outlier-bearing weight set amplifies ANY operand rounding 15-25x (single-pass f16 lands 1.3e-2 from the strict form there
        # against 5e-4 on benign weights, tools/prec_map_probe.py): the default form's 15 bits would give 4.8e-4 (BT row) / 2.6e-3
        # (GPM row).  .to('cuda') measured that on its probe rows -- no reference, no caller batches -- and locked a form with 16-bit
        # residual passes where this model needs them, so the unchanged drop-in sequence stays inside the bar.
        print(f"[{g['name']} {dtype}] form locked by .to('cuda'): {m.form_info}")
        record_locked_form(g['name'], dtype, m, err)
        assert m.operand_form != "default" and err < TOL_OUTLIER     # (strict, or strict from the front of the model: _form_candidates)
    elif dtype == "f16x2f8":
        # default parity mode (e4m3 residual passes): <= 7e-5 on every benign row, and the probe keeps benign weights in that form
        # (the form is printed, not asserted: a benign weight set whose probe rows land above the budget runs strict -- slower, never
        #  less exact; bench.py prints the form its timed engine locked)
        print(f"[{g['name']} {dtype}] form locked by .to('cuda'): {m.form_info}")
        record_locked_form(g['name'], dtype, m, err)
        assert err < TOL_X8
    else:
        # single-pass f16: 1e-3 in the assert_close sense (atol = rtol = 1e-3).  At full depth (23 + 32 layers) numerically
        # equivalent builds land anywhere within about +-1e-3 of the reference on this row (sigma ~ 7e-4 at |r| = 1.3,
        # tools/noise_probe.py; bf16 operands: +-8e-3), see DESIGN.md §4.
        assert (((got - ref).abs() <= TOL_F16 + TOL_F16 * ref.abs()).all())
    # the same rows twice in one batch: bit-identical rewards
    dup = {k: np.concatenate([v, v]) for k, v in batch.items()}
    r2 = _fwd(m, dup)
    assert torch.equal(r2[:B], r2[B:]) and torch.equal(r2[:B].reshape(got.shape), got)
    # the gathered last layer (default) against the same forward with every token kept through it: bit-identical
    m.keep_hidden_states = True
    assert torch.equal(_fwd(m, batch).reshape(got.shape), got)


FULL_TAPS = [p for p in FULL if json.load(open(p)).get("taps")]


@pytest.mark.parametrize("path", FULL_TAPS, ids=[os.path.basename(p)[:-5] for p in FULL_TAPS])
def test_reference_golden_full_size_stage_taps(path):
    """64-element slices of the reference's hidden states at full size (B = 2, left padding, ragged crop grids):
    embeddings, the stream after layers 0 / 1 / 15 / 16 / 30, the final norm and the zero-padded vision rows."""
    g = json.load(open(path))
    _check_golden_taps(g, max_crops=17, tol=2e-4)        # (fp32 summation-order noise grows with depth: 1e-5 after layer 1, 5e-5 after 15)


def test_config0_sample_pair_through_the_drop_in_api(tmp_path):
    """BASELINE configs[0] (eval/simple_inference.py:16-31) at full Phi-3.5-V size through the drop-in callables: ONE caption, TWO
    512x640 images on disk -> inference_process_phi3v_device (image hand-over + prompt/slot merge on the GPU side) -> two B=1
    custom_forward calls -> preference_compute, against tests/golden/ref_full_pair_sample.json: the REFERENCE's custom_forward on
    the same two rows (pixel_values from the HD-transform oracle, same stand-in tokenizer; make_goldens.py pair_sample)."""
    from PIL import Image
    from llava_reward_amd import preprocess as P
    g = json.load(open(os.path.join(GOLD, "ref_full_pair_sample.json")))
    cfg = synth.RewardConfig.from_json(g["config"])
    paths = []
    for i in range(2):
        p = str(tmp_path / f"img{i}.png")
        Image.fromarray(synth.synth_image(g["seed"], f"pair.image{i}", 640, 512, True)).save(p)
        paths.append(p)
    assert Image.open(paths[0]).size == (512, 640)

    class Args:
        is_general_preference, value_head_dim, general_preference_tau = cfg.is_general_preference, cfg.value_head_dim, g["tau"]
    rows = P.inference_process_phi3v_device(Args, synth.StandInTokenizer(), paths, g["caption"], device="cuda", num_crops=g["num_crops"])
    for d, meta in zip(rows, g["rows"]):
        ids = d["input_ids"][0].tolist()
        assert d["image_sizes"].tolist() == [meta["image_sizes"]] and len(ids) == meta["seq_len"]
        assert sum(1 for t in ids if t < 0) == meta["num_img_tokens"] == 2509
        assert ids[:8] == meta["input_ids_head"] and ids[-8:] == meta["input_ids_tail"]
    S = rows[0]["input_ids"].shape[1]
    model = RewardModel(cfg, synth_seed=g["seed"], max_batch=1, max_seq=S, max_crops=17).to("cuda").eval()       # default mode f16x2f8
    with torch.no_grad():
        c, _ = model.custom_forward(**rows[0])
        r, _ = model.custom_forward(**rows[1])
    prob = preference_compute(Args, c, r)
    ref = torch.tensor(g["reward"])
    err = max((c.cpu() - ref[0]).abs().max().item(), (r.cpu() - ref[1]).abs().max().item())
    print(f"[configs[0] sample pair] rewards hip=({c.item():.6f}, {r.item():.6f}) ref=({ref[0].item():.6f}, {ref[1].item():.6f}) err={err:.2e} "
          f"prob hip={prob[0]:.6f} ref={g['prob'][0]:.6f}")
    assert err < TOL_X8
    assert prob.dtype == np.float32 and prob.shape == (1,) and abs(float(prob[0]) - g["prob"][0]) < 0.25 * 2 * TOL_X8 / g["tau"]   # |sigmoid'| <= 1/4
    assert (prob[0] > 0.5) == (g["prob"][0] > 0.5)                      # same preference as the reference
    c2, _ = model.custom_forward(**rows[0])                             # bit-stable
    assert torch.equal(c2, c)


def test_config2_gpm_pairwise_batch64_full_size():
    """BASELINE configs[2]: Phi-3.5-V GPM head (value_head_dim = 2) + SkipCA, pairwise, B = 64 rows per forward at full size.  Row 0 is
    the reference's own full-size GPM golden row (ref_full_gpm2_ca.json: same seed -> same caption, pixels and weights); size-
    independent properties carry the rest: any row scored alone (B = 1) or in a 32-row shard gives bit-identical rewards, hence a
    bit-identical preference probability."""
    g = json.load(open(os.path.join(GOLD, "ref_full_gpm2_ca.json")))
    cfg = synth.RewardConfig.from_json(g["config"])
    assert cfg.is_general_preference and cfg.value_head_dim == 2 and cfg.add_cross_attention
    B = 64
    b = synth.synth_batch(cfg, g["seed"], [128] * B, (4, 4), with_pixels=False)
    ids, mask = torch.from_numpy(b["input_ids"]).cuda(), torch.from_numpy(b["attention_mask"]).cuda()
    sizes = torch.from_numpy(b["image_sizes"])
    gen = torch.Generator(device="cuda").manual_seed(5)
    pix = torch.randn(B, 17, 3, 336, 336, device="cuda", generator=gen)
    pix[0] = torch.from_numpy(synth.synth_pixels(g["seed"], "pixel_values.0", (17, 3, 336, 336))).cuda()
    m = RewardModel(cfg, synth_seed=g["seed"], max_batch=B, max_seq=ids.shape[1], max_crops=17).to("cuda").eval()
    full, _ = m.custom_forward(ids, mask, pix, sizes)
    torch.cuda.synchronize()
    ref = torch.tensor(g["reward"], dtype=torch.float32)
    err = (full[0].cpu() - ref[0]).abs().max().item()
    print(f"[configs[2] B=64 GPM] row 0 hip={full[0].tolist()} ref={ref[0].tolist()} err={err:.2e}")
    assert full.shape == (B, 2) and torch.isfinite(full).all() and err < TOL_X8
    for i in (0, 17, 63):
        one, _ = m.custom_forward(ids[i:i + 1], mask[i:i + 1], pix[i:i + 1], sizes[i:i + 1])
        assert torch.equal(one[0], full[i])
    lo, _ = m.custom_forward(ids[:32], mask[:32], pix[:32], sizes[:32])
    hi, _ = m.custom_forward(ids[32:], mask[32:], pix[32:], sizes[32:])
    assert torch.equal(torch.cat([lo, hi]), full)

    class A:
        is_general_preference, value_head_dim, general_preference_tau = True, 2, cfg.general_preference_tau
    p_full = preference_compute(A, full[:32], full[32:])
    assert p_full.shape == (32,) and p_full.dtype == np.float32 and np.array_equal(p_full, preference_compute(A, lo, hi))


@pytest.mark.parametrize("backbone", ["phi3v", "qwen"])
def test_parity_mode_with_inexact_weights(backbone):
    """Weights that are NOT bf16-valued (what a LoRA merge W + (alpha/r) B A leaves behind): the split-operand mode carries
    the weights' rounding residuals as a third K segment, [x_hi | x_lo | x_hi] x [W | W | W_lo], and still matches the fp32
    oracle run on the same fp32 weights; the single-pass f16 mode shows the extra weight-rounding error."""
    if backbone == "phi3v":
        cfg = synth.tiny_config()
        Wn = synth.make_weights(cfg, 17)
        batch = synth.synth_batch(cfg, 17, [7, 3, 5], [(1, 1), (1, 2), (2, 1)], max_crops=4)
    else:
        cfg = synth.qwen_tiny_config()
        Wn = synth.qwen_make_weights(cfg, 17)
        batch = synth.qwen_synth_batch(cfg, 17, [7, 3, 5], [(16, 16), (10, 6), (18, 22)])
    g = torch.Generator().manual_seed(5)
    W = {}
    for k, v in Wn.items():
        t = torch.from_numpy(v)
        if t.dim() >= 2 and "embed_tokens" not in k:          # every matrix gets an fp32 perturbation below its bf16 ulp
            t = t * (1.0 + 2.0 ** -10 * (torch.rand(t.shape, generator=g) - 0.5))
        W[k] = t.float()
    if backbone == "phi3v":
        ref = orc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
    else:
        from oracle import qwen2_5_vl_reward_oracle as qorc
        ref = qorc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_grid_thw"])
    errs = {}
    for dtype in ("f16x2", "f16x2f8", "f16"):
        m = RewardModel(cfg, weights=W, max_batch=4, max_seq=1024, max_crops=5, max_patches=4096, operand_dtype=dtype, calibrate=False).to("cuda").eval()
        if dtype == "f16x2f8":
            m.engine.set_gemm_tile(6)       # deep-pipelined kernel everywhere: inexact weights take the e4m3 third segment (A_hi8 x Wlo8)
        tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
        if backbone == "phi3v":
            r, _ = m.custom_forward(tb["input_ids"], tb["attention_mask"], tb["pixel_values"], tb["image_sizes"])
        else:
            r, _ = m.custom_forward(inputs_batch=tb)
        torch.cuda.synchronize()
        errs[dtype] = (r.cpu() - ref).abs().max().item()
    print(f"[inexact weights, {backbone}] f16x2 err {errs['f16x2']:.2e}   f16 err {errs['f16']:.2e}")
    print(f"[inexact weights, {backbone}] f16x2f8 err {errs['f16x2f8']:.2e}")
    assert errs["f16x2"] < TOL_X2 and errs["f16x2f8"] < TOL_X8 and errs["f16x2f8"] < 0.5 * errs["f16"] and errs["f16"] < 3e-3


def test_right_padding_single_row_and_all_padding_row():
    """Edge cases of the EOS gather (rw_model:420: S-1-argmax(flip(mask))): right padding, a batch of one, and a row whose
    mask is all zero (argmax of zeros = 0 -> index S-1; the reference scores garbage there without failing, so must we)."""
    cfg = synth.tiny_config(is_general_preference=True, value_head_dim=2)
    seed = 23
    b = synth.synth_batch(cfg, seed, [6, 2], (1, 1))
    S = b["input_ids"].shape[1]
    ids, mask = b["input_ids"].copy(), b["attention_mask"].copy()
    n1 = int(mask[1].sum())
    ids[1] = np.concatenate([b["input_ids"][1][S - n1:], np.full(S - n1, b["input_ids"][1][0])])      # row 1: right padded
    mask[1] = np.concatenate([np.ones(n1, dtype=np.int64), np.zeros(S - n1, dtype=np.int64)])
    batch = dict(b, input_ids=ids, attention_mask=mask)
    W = orc.weights_to_torch(synth.make_weights(cfg, seed))
    ref = orc.custom_forward(W, cfg, ids, mask, b["pixel_values"], b["image_sizes"])
    m = _model(cfg, seed, "f16x2", upload=False)
    got = _fwd(m, batch)
    assert (got - ref).abs().max().item() < TOL_X2
    one = _fwd(m, batch, rows=slice(1, 2))                        # B = 1
    assert torch.equal(one[0], got[1])
    dead = dict(batch, attention_mask=np.concatenate([mask[:1], np.zeros((1, S), dtype=np.int64)]))
    r = _fwd(m, dead)                                             # row 1 fully masked: finite-or-not, row 0 must be untouched
    assert torch.equal(r[0], got[0]) and r.shape == got.shape


def test_bf16x2_mode():
    """Split-operand mode with bf16 operands (hi + lo = 16 mantissa bits): exposed for completeness; far tighter than single-pass
    bf16 (8e-3) though not the parity mode (f16x2 carries 22 bits)."""
    cfg = synth.tiny_config()
    seed = 29
    batch = synth.synth_batch(cfg, seed, [7, 3], [(1, 1), (1, 2)], max_crops=4)
    W = orc.weights_to_torch(synth.make_weights(cfg, seed))
    ref = orc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
    got = _fwd(_model(cfg, seed, "bf16x2", upload=False), batch)
    err = (got - ref).abs().max().item()
    print(f"[bf16x2] max |reward err| = {err:.3e}")
    assert err < 2e-4


@pytest.mark.parametrize("variant", ["bt_ca", "gpm2_ca"])
def test_e4m3_residual_pass_on_tiny_config(variant):
    """"f16x2f8" with the deep-pipelined kernel forced for every GEMM (tile 6), so that the in-place residual encoder, the W8
    twins and the mixed f16 / e4m3 K loop run on a config the oracle finishes in seconds (K = 128 .. 640)."""
    kw = dict(bt_ca={}, gpm2_ca=dict(is_general_preference=True, value_head_dim=2))[variant]
    cfg = synth.tiny_config(**kw)
    seed = 17
    batch = synth.synth_batch(cfg, seed, [7, 3, 5], [(1, 1), (1, 2), (2, 1)], max_crops=4)
    W = orc.weights_to_torch(synth.make_weights(cfg, seed))
    ref = orc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
    errs = {}
    for dtype in ("f16x2f8", "f16"):
        m = _model(cfg, seed, dtype, upload=False, **({"calibrate": False} if dtype == "f16x2f8" else {}))
        m.engine.set_gemm_tile(6)
        got = _fwd(m, batch)
        errs[dtype] = (got - ref).abs().max().item()
        if dtype == "f16x2f8":
            again = _fwd(m, batch)                  # second pass: the W8 twins are reused, the residuals re-encoded
            assert torch.equal(got, again)
    print(f"[tiny {variant}, tile 6] f16x2f8 err {errs['f16x2f8']:.2e}   f16 err {errs['f16']:.2e}")
    assert errs["f16x2f8"] < TOL_X8 and errs["f16x2f8"] < 0.5 * errs["f16"] + 2e-5


@pytest.mark.parametrize("backbone", ["phi3v", "llava"])
def test_w8a8_mode_against_quantisation_aware_oracle(backbone):
    """W8A8 mode (operand_dtype="fp8", BASELINE configs[4] "fp8 MFMA weight path"): every GEMM with K % 128 == 0 runs on e4m3
    operands with per-row / per-channel scales.  NOT a parity mode.  The oracle is evaluated with the SAME operand quantisation
    (W8A8Round: f16 storage + per-row e4m3 in front of each GEMM).  An e4m3 model is chaotic in its inputs -- two emulations that
    differ only in the 2^-11 activation rounding (f16 vs none) already differ by ~3e-2 in the reward -- so the bars are: the
    engine sits as close to the emulation as such a twin does, far inside the quantisation noise at the first stage (CLIP
    tower, where little has been amplified yet), finite and deterministic."""
    seed = 23
    taps_hip = None
    if backbone == "phi3v":
        cfg = synth.tiny_config()
        batch = synth.synth_batch(cfg, seed, [7, 3, 5], [(1, 1), (1, 2), (2, 1)], max_crops=4)
        W = orc.weights_to_torch(synth.make_weights(cfg, seed))

        def fwd(opr, taps=None):
            return orc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"],
                                      opr=opr, taps=taps)

        def hip(dtype):
            m = _model(cfg, seed, dtype, upload=False)
            r = _fwd(m, batch)
            ncrop = int(sum(h // 336 * (w // 336) + 1 for h, w in batch["image_sizes"].tolist()))
            T, Hc = cfg.clip.tokens, cfg.clip.hidden
            return r, m.engine.read_tap("clip_x", ncrop * T * Hc).reshape(ncrop, T, Hc)[:, 1:]
    else:
        from oracle import llava_next_reward_oracle as lorc
        cfg = synth.llava_tiny_config()
        batch = synth.llava_synth_batch(cfg, seed, [6, 3], [(336, 336), (300, 500)])
        W = orc.weights_to_torch(synth.llava_make_weights(cfg, seed))

        def fwd(opr, taps=None):
            return lorc.custom_forward(W, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"], opr=opr)

        def hip(dtype):
            m = RewardModel(cfg, synth_seed=seed, max_batch=4, max_seq=4096, max_crops=5, operand_dtype=dtype).to("cuda").eval()
            r, _ = m.custom_forward(inputs_batch={k: torch.from_numpy(v).cuda() for k, v in batch.items()})
            torch.cuda.synchronize()
            return r.cpu(), None
    t32, t8 = {}, {}
    ref32 = fwd(orc.Ident, t32)
    ref8 = fwd(orc.W8A8Round(orc.f16_round), t8)
    twin = fwd(orc.W8A8Round(orc.Ident))                       # same quantisation, activations not rounded to f16 first
    got, clip = hip("fp8")
    e_sim = (got - ref8).abs().max().item()
    e_twin = (twin - ref8).abs().max().item()
    q_noise = (ref8 - ref32).abs().max().item()
    print(f"[w8a8 {backbone}] |hip - emulation| = {e_sim:.2e}   |twin emulation - emulation| = {e_twin:.2e}   "
          f"|emulation - fp32 oracle| = {q_noise:.2e}   |hip - fp32 oracle| = {(got - ref32).abs().max().item():.2e}")
    assert torch.isfinite(got).all()
    assert e_sim < 3.0 * max(e_twin, 1e-2)
    if clip is not None and "clip_out" in t8:
        used = [h // 336 * (w // 336) + 1 for h, w in batch["image_sizes"].tolist()]       # the engine skips the zero crops

        def pick(t):
            t = t.reshape(len(used), -1, clip.shape[1], clip.shape[2])
            return np.concatenate([t[i, :n].numpy() for i, n in enumerate(used)])
        rc8, rc32 = pick(t8["clip_out"]), pick(t32["clip_out"])
        d_hip, d_q = np.abs(clip - rc8).max(), np.abs(rc32 - rc8).max()
        print(f"[w8a8 {backbone}] CLIP tower: |hip - emulation| = {d_hip:.2e}   |fp32 - emulation| = {d_q:.2e}")
        assert d_hip < 0.25 * d_q                               # the engine reproduces the quantised tower, not merely something e4m3-ish
    assert torch.equal(got, hip("fp8")[0])                        # deterministic, twins rebuilt identically


@pytest.mark.parametrize("dtype", ["f16x2f8", "fp8"])
def test_weight_reupload_after_finalize_rebuilds_e4m3_twins(dtype):
    """The e4m3 twins of the GEMM weights are built at lr_finalize; a weight uploaded afterwards must get a fresh twin (and, if it is
    no longer bf16-valued, the third-segment form) at its next launch."""
    cfg = synth.tiny_config()
    seed = 29
    batch = synth.synth_batch(cfg, seed, [6, 4], [(1, 1), (1, 2)], max_crops=4)
    Wn = synth.make_weights(cfg, seed)
    m = _model(cfg, seed, dtype, upload=True, **({"calibrate": False} if dtype == "f16x2f8" else {}))
    m.engine.set_gemm_tile(6)
    r0 = _fwd(m, batch)
    name = "model.layers.1.mlp.down_proj.weight"
    g = torch.Generator().manual_seed(1)
    w = torch.from_numpy(Wn[name])
    m.engine.upload(name, (w * (1.0 + 0.05 * torch.randn(w.shape, generator=g))).float())       # fp32-valued: inexact in f16
    r1 = _fwd(m, batch)
    assert (r1 - r0).abs().max().item() > 1e-4                                                   # the new weight is in effect
    if dtype == "f16x2f8":                                                                       # and it is computed to parity
        W2 = orc.weights_to_torch(Wn)
        W2[name] = (w * (1.0 + 0.05 * torch.randn(w.shape, generator=torch.Generator().manual_seed(1)))).float()
        ref = orc.custom_forward(W2, cfg, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
        assert (r1 - ref).abs().max().item() < TOL_X8
    m.engine.upload(name, w)
    r2 = _fwd(m, batch)
    if dtype == "fp8":
        assert torch.equal(r2, r0)
    else:       # the buffer stays flagged inexact (conservative): same value to parity, not necessarily to the bit
        assert (r2 - r0).abs().max().item() < TOL_X8


def test_su_rope_switch_point_of_the_flash_attention_class():
    """lr_model_desc.rope_flash_convention (RewardConfig.rope_flash_convention; set from the checkpoint's `_attn_implementation` or
    args.flash_attn).  Phi3FlashAttention2 calls the rotary module with max(S, position_ids[:, -1].max()) + 1 = S + 1
    (modeling_phi3_v.py:793-794) where the eager / sdpa classes pass S (:673): the long factors start at S >= original_max instead
    of S > original_max.  flash-attn is absent here, so there is no reference golden for it; what can be pinned without one:
    at S == original_max the flash convention equals an eager model whose SHORT factors are the long ones (same scaling constant),
    bit for bit, and differs from the eager model; one token below the boundary the two conventions are the same function."""
    import dataclasses
    seed = 31
    probe = synth.synth_batch(synth.tiny_config(), seed, [9, 4], (1, 1))
    S = probe["input_ids"].shape[1]
    for orig, same in ((S, False), (S + 1, True)):
        cfg_e = synth.tiny_config(orig_max_pos=orig, max_pos=32 * orig)
        cfg_f = dataclasses.replace(cfg_e, rope_flash_convention=True)
        cfg_l = dataclasses.replace(cfg_e, short_factor=cfg_e.long_factor)            # eager, but "short" = the long factors
        batch = synth.synth_batch(cfg_e, seed, [9, 4], (1, 1))
        assert batch["input_ids"].shape[1] == S
        e, f, l = (_fwd(_model(c, seed, "f16x2", upload=False), batch) for c in (cfg_e, cfg_f, cfg_l))
        if same:          # S == original_max - 1: short factors under both conventions
            assert torch.equal(e, f) and not torch.equal(f, l)
        else:             # S == original_max: the flash class is already on the long factors
            assert torch.equal(f, l) and not torch.equal(e, f)
            W = orc.weights_to_torch(synth.make_weights(cfg_l, seed))
            ref = orc.custom_forward(W, cfg_l, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_sizes"])
            assert (f - ref).abs().max().item() < TOL_X2
