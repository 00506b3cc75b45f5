"""The object that stands where the reference's CustomRewardModel stands
(llava_reward/models/rw_model_general_preference.py:290-448): `.to(device)`, `.eval()`,
`.model_type`, and `custom_forward(...) -> (reward, outputs | None)`.

Weights live on the host until `.to('cuda')`, which creates the HIP engine on that GPU (the
reference's `load_reward_adaptor` also returns a CPU model that the caller moves).
Deviations, on purpose: rewards come back in fp32 unless reward_dtype says otherwise (the reference
returns the model dtype, bf16; SURVEY.md §7 shows bf16 rounding alone costs up to 2e-3); `outputs` for
return_output=True holds `last_hidden_state` (what the reference's callers read), the hidden row behind
the reward, and `hidden_states` as a tuple-like in the reference's layout (phi3v: layers + 2 entries, `vision_embeds`
last) that recomputes an inner layer's tensor when it is asked for (the backbone's output object carries all at once)."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .synth import LlavaConfig, QwenConfig, RewardConfig, llava_geometry


def _now(device) -> float:
    """Host clock behind a drained stream (the probe's wall time, reported by bench.py)."""
    import time
    if getattr(device, "type", "cpu") == "cuda":
        torch.cuda.synchronize(device)
    return time.perf_counter()


class _Outputs(dict):
    """`outputs` of custom_forward(return_output=True).  The reference hands back the backbone's whole output object; this one
    holds what its callers read.  A key the forward did not compute fails with the reason instead of a bare KeyError."""

    def __missing__(self, key):
        if key == "last_hidden_state":
            raise KeyError("last_hidden_state: with layer_id != 32 the forward stops after `layer_id` decoder layers (the reward is read from "
                           "hidden_states[layer_id], rw_model_general_preference.py:349-352), so the final norm of the full stack was never "
                           "computed; read outputs['hidden_states_at_layer_id'], or build the model with layer_id=32")
        if key == "hidden_states":
            raise KeyError("hidden_states: the per-layer tuple is served on demand for the phi3v / llava branches with layer_id == 32 only "
                           "(_LazyHiddenStates); here use layer_id=<k> for one of them")
        if key == "vision_embeds":
            raise KeyError("vision_embeds: read outputs['hidden_states'][-1] (phi3v branch, layer_id == 32), as the reference does "
                           "(rw_model_general_preference.py:353)")
        raise KeyError(key)


class _LazyHiddenStates:
    """`outputs["hidden_states"]` of the reference's backbone output (`output_hidden_states=True`, rw_model_general_preference.py:346-353,
    :372-375), in the reference's layout:

      phi3v  (modeling_phi3_v.py:1467-1505)  layers + 2 entries: [k], k < layers = the residual stream entering decoder layer k ([0] = the
             embeddings with the image rows scattered in), [layers] = the final norm's output (the same tensor as `last_hidden_state`),
             [-1] = `vision_embeds` [B, V_max, hidden], the projected image tokens zero-padded per sample (:242-245) -- the entry
             custom_forward itself reads back as the SkipCA key / value source (rw_model:353, vision_layer_id = -1);
      llava  (transformers LlavaNext, rw_model:372-375)  layers + 1 entries, [-1] = the final norm's output.

    The engine keeps ONE residual stream, so an inner element is recomputed when it is asked for -- the same forward stopped after k
    layers (lr_set_layer_limits + LR_FWD_NO_FINAL_NORM, the mechanism behind `layer_id`) -- and cached; the final norm and
    `vision_embeds` are taken at the forward itself.  fp32, on the device.  Holds references to the forward's input tensors.  The
    weights the elements belong to are those of the forward: an access after a weight upload / re-synthesis (lr_weights_epoch moved)
    raises instead of returning another model's states.  An access runs a forward on the shared engine, i.e. it replaces the engine's
    last-forward state (read_tap / lr_last_hidden_state) like any other custom_forward call.  len(), indexing (negative, slices) and
    iteration work as on the tuple."""

    def __init__(self, model, args, shape, final_norm, vision_embeds):
        self._m, self._args, self._shape = model, args, (int(shape[0]), int(shape[1]))
        self._n = int(model.config.layers)
        self._epoch = model.engine.weights_epoch()
        self._engine = model.engine
        self._cache = {self._n: final_norm}
        if vision_embeds is not None:
            self._cache[self._n + 1] = vision_embeds

    def __len__(self):
        return self._n + (2 if self._n + 1 in self._cache else 1)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return tuple(self[i] for i in range(*k.indices(len(self))))
        k = int(k)
        if k < 0:
            k += len(self)
        if not 0 <= k < len(self):
            raise IndexError("tuple index out of range")
        if k in self._cache:
            return self._cache[k]
        m, (B, S) = self._m, self._shape
        if m.engine is not self._engine or m.engine.weights_epoch() != self._epoch:
            raise RuntimeError("outputs['hidden_states']: the model's weights changed (or it moved to another device) after the forward these "
                               "states belong to; run custom_forward(return_output=True) again")
        ids, mask, pix, sz = self._args
        m.engine.set_layer_limits(-1, k)
        try:
            m.engine.forward(ids, mask, pix, sz, training=False, no_final_norm=True, keep_hidden_states=True)
            self._cache[k] = m.engine.last_hidden_state(B, S, no_final_norm=True)
        finally:
            m.engine.set_layer_limits(-1, -1)
        return self._cache[k]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class RewardModel:
    def __init__(self, cfg, weights: Optional[Dict[str, torch.Tensor]] = None, synth_seed: Optional[int] = None,
                 max_batch: int = 32, max_seq: int = 2816, max_crops: int = 17, operand_dtype: str = "f16x2f8",
                 layer_id: int = 32, mean_hidden_state=None, max_patches: int = 0, synth_profile: int = 0,
                 calibrate: bool = True, parity_budget: float = 1.5e-4, operand_form: Optional[str] = None,
                 check_inputs: str = "eager", reward_dtype: Optional[torch.dtype] = None, vision_layer_id: int = -1):
        if weights is None and synth_seed is None:
            raise ValueError("RewardModel needs weights or a synth_seed")
        # rw_model:296,353: the SkipCA key / value source is hidden_states[vision_layer_id][:, :V_max]; -1 (the reference's default, which
        # no caller or script of the reference overrides) = the zero-padded projected image tokens.  Any other index would read the
        # first V_max TOKEN positions of a decoder state instead -- not served (the engine keeps one residual stream and folds W_k / W_v
        # onto the image rows); refused here rather than silently scored with the default source.
        if int(vision_layer_id) != -1:
            raise NotImplementedError(f"vision_layer_id={vision_layer_id}: only -1 (hidden_states[-1] = vision_embeds, the reference's default, "
                                      "rw_model_general_preference.py:296,353) is served")
        self.vision_layer_id = -1
        # rw_model:349-352: layer_id == 32 (the literal) -> last_hidden_state, else hidden_states[layer_id] = the input of decoder
        # layer `layer_id` (un-normed residual stream; index `layers` is the final-norm output again).  phi3v branch only.
        if not (layer_id == 32 or 0 <= layer_id <= cfg.layers):
            raise IndexError("tuple index out of range (hidden_states[layer_id])")
        self.config = cfg
        self.model_type = "qwen" if isinstance(cfg, QwenConfig) else "llava" if isinstance(cfg, LlavaConfig) else "phi3v"
        self._weights = weights
        self._synth_seed = synth_seed
        self._synth_profile = int(synth_profile)      # synth.PROFILE_* (outlier-bearing / e4m3-valued synthetic weights)
        self.mean_hidden_state = bool(mean_hidden_state)      # rw_model:398-406: masked mean of the (SkipCA'd) hidden states
        self._opts = dict(max_batch=max_batch, max_seq=max_seq, max_crops=max_crops, operand_dtype=operand_dtype,
                          max_patches=max_patches, mean_hidden_state=self.mean_hidden_state)
        self.layer_id = layer_id
        self.engine = None
        # True: every forward keeps all hidden states through the last layer (read_tap("x") / last_hidden_state afterwards);
        # custom_forward(return_output=True) does so by itself.  Default: the last layer computes the reward rows only.
        self.keep_hidden_states = False
        # Operand form of the default parity mode ("f16x2f8"), locked by .to('cuda') on the loaded weights (_lock_operand_form):
        # "default" = f16 hi + e4m3 residual passes everywhere, "strict" = 16-bit residual passes everywhere (1.17x the step time),
        # or one of the forms in between (_form_candidates: strict from the front of the model).
        # calibrate=False (load_reward_adaptor: args.calibrate = False) skips the self-check and keeps "default".
        self.operand_form = "default"
        self.form_info: Optional[Dict[str, object]] = None      # {"form", "default_vs_strict", "source", "rows", "budget"} of the last check
        self.auto_calibrate = bool(calibrate)
        self.parity_budget = float(parity_budget)
        self._form_epoch = None                # lr_weights_epoch the locked form belongs to
        # operand_form="<name of a _form_candidates entry>" (load_reward_adaptor: args.operand_form) PINS the form: no probe, the same
        # form on every deployment of the checkpoint whatever the engine's capacity (the probe rows depend on max_crops / max_seq).
        self.pinned_form = operand_form
        if operand_form is not None and operand_form not in dict(self._form_candidates() + self._pinnable_forms()):
            raise ValueError(f"operand_form={operand_form!r}: not one of {[n for n, _ in self._form_candidates() + self._pinnable_forms()]}")
        # "eager": the reference's exceptions for rows whose image-slot count does not match their image_sizes, raised by this call
        # (host tensors are counted on the host; DEVICE input_ids cost a stream drain per forward).  "deferred": no host-side check
        # and no synchronisation -- the engine itself marks such a row's reward NaN (slot_check_kernel), visible at the caller's next
        # read of the rewards (phi3v / llava; the qwen branch always checks: its engine entry has no such kernel).
        if check_inputs not in ("eager", "deferred"):
            raise ValueError("check_inputs must be 'eager' or 'deferred'")
        self.check_inputs = check_inputs
        # None (default): rewards come back in fp32 (module docstring).  torch.bfloat16: rounded to the dtype the reference's GPU path
        # returns (its model dtype, rw_model:420-444) -- for callers that compare dtypes or concatenate with the reference's outputs.
        self.reward_dtype = reward_dtype
        self._in_probe = False
        self.training = False
        self.device = torch.device("cpu")
        self.is_general_preference = cfg.is_general_preference
        self.add_cross_attention = cfg.add_cross_attention
        self.value_head_dim = cfg.value_head_dim

    # -- nn.Module-like surface used by the reference's callers (eval/simple_inference.py:17-18) --
    def to(self, device):
        device = torch.device(device)
        if device.type == "cpu":
            return self
        if device.type != "cuda":
            raise ValueError(f"unsupported device {device}")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self.engine is not None and self.engine.device == idx:
            return self
        from .engine import RewardEngine
        eng = RewardEngine(self.config, device=idx, **self._opts)
        if self._weights is not None:
            eng.load_state_dict(self._weights, strict=True)
        else:
            eng.synth_weights(self._synth_seed, getattr(self, "synth_fp32_valued", False), self._synth_profile)
        eng.finalize()
        moved = self.engine is not None          # same weights on another device: the locked form is a property of the weights
        if self.engine is not None:
            self.engine.close()
        self.engine = eng
        self.device = torch.device("cuda", idx)
        if moved and self.form_info is not None:
            self._form_epoch = eng.weights_epoch()
            self._apply_form()
        else:
            self._lock_operand_form()
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device if device is not None else torch.cuda.current_device()))

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        self.training = bool(mode)
        return self

    # -- operand form of the default parity mode: locked on the weights, never on the data being scored --
    def _form_candidates(self):
        """(name, lr_set_precision_map arguments) in the order of their cost.  Which stages need 16-bit residual passes on a weight
        set that amplifies operand rounding was measured on the outlier-bearing synthetic set (tools/prec_map_probe.py ... strict-stages,
        8 full-size rows, distance to the strict form): default 2.2e-3, vision tower strict 4.4e-4, vision tower + first half of the
        decoder strict 8.7e-5 -- noise injected early is what the depth amplifies; the LAST decoder layers never matter
        (decoder 16..31 strict alone: 2.2e-3) -- so the candidates grow from the front of the model."""
        L = int(self.config.layers)
        out = [("default", (-1, -1, 0, 0)), ("strict-vision", (1, -1, 0, 0))]
        seen = set()
        for num, den, tag in ((1, 8, "/8"), (1, 4, "/4"), (3, 8, "*3/8"), (1, 2, "/2")):
            k = (L * num) // den
            if 0 < k < L and k not in seen:
                seen.add(k)
                out.append((f"strict-vision+decoder{tag}", (1, 1, 0, L - k)))
        out.append(("strict", (1, 1, 0, 0)))
        return out

    def _pinnable_forms(self):
        """Forms that can be PINNED (operand_form="<name>") but are never locked by the probe: single-pass operands in the LAST decoder
        layers on top of the default form -- the only lever that removes MFMA work from a power-limited chip.  Measured (round 5,
        tools/prec_map_probe.py 8 0 ladder, 8 benign full-size Phi-3.5-V rows, max distance to the strict form): default 6.5e-5; last 2
        layers single 1.0e-4 (-0.6 % step time), last 4 1.3e-4 (-2.2 %), last 8 1.5e-4 (-5.4 %) -- the last layers amplify rounding
        least, but 11-bit operands there still cost as much as the whole default form does everywhere.  Tried as probe candidates
        (cheapest first) on 26 full-size goldens: the locked form became a lottery at the budget's edge (tails passed on some benign
        seeds at 0.9-1.5e-4 and failed on others at 1.7-4e-4; one LLaVA golden landed 2.5e-4 from the reference, against 9.4e-5 in
        the default form), so they stay opt-in."""
        L = int(self.config.layers)
        return [(f"default+single-tail/{den}", (-1, 0, L - L // den, 0)) for den in (4, 8, 16) if 2 <= L // den < L]

    def _apply_form(self) -> None:
        if self.engine is None or self._opts["operand_dtype"] != "f16x2f8":
            return
        self.engine.set_precision_map(*dict(self._form_candidates() + self._pinnable_forms())[self.operand_form])

    def _compare_forms(self, batches, budget: float, source: str) -> Dict[str, object]:
        """Score `batches` in the strict form (the yardstick: 16-bit residual passes everywhere) and in the cheaper forms of
        _form_candidates, cheapest first; lock the first one whose rewards all sit within `budget` of the strict form's (NaN anywhere:
        not a form to trust).  Whatever happens, the engine is left in the form `self.operand_form` names."""
        eng = self.engine
        was_training, self.training = self.training, False
        self._in_probe = True
        try:
            def score(args):
                eng.set_precision_map(*args)
                return [self.custom_forward(**b)[0].float().clone() for b in batches]
            t_start = _now(self.device)
            cands = self._form_candidates()
            strict = score(cands[-1][1])
            # The yardstick's own fp32 noise on these weights: the strict form once more with the attention launches' lazy reference
            # maximum (threshold 8: the same softmax, other fp32 roundings of the exponents' arguments -- a numerically equivalent
            # re-statement of the strict form).  On benign weights the two sit ~1e-6 apart; a weight set that amplifies rounding shows
            # 1e-4 and more -- the reference's own fp32 arithmetic sits 2.1e-4 from the fp64 value on such a row
            # (tests/golden/fp64_full_rows.json) -- and no cheaper form can be asked to sit closer to the strict one than the strict one
            # sits to itself: the budget becomes max(budget, 1.5 x floor), capped at 4/3 of the budget (2e-4 at the default: the locked
            # form's error against the reference then stays inside the 3e-4 the golden tests hold it to).
            floor = 0.0
            if source == "probe":
                eng.set_attention_lazy_threshold(8.0, 8.0)
                try:
                    for a, st in zip(score(cands[-1][1]), strict):
                        x = float((a - st).abs().max()) if a.numel() else 0.0
                        floor = float("inf") if x != x else max(floor, x)
                finally:
                    eng.set_attention_lazy_threshold(8.0, 0.0)
                if floor == floor and floor != float("inf"):
                    budget = min(max(budget, 1.5 * floor), budget * 4.0 / 3.0)
            import torch.distributed as dist
            # Collective ONLY in the explicit calibrate() on user batches (documented as collective).  The probe of .to('cuda') / of
            # a forward after a weight upload must not be one: the reference's .to() and forward have no collectives (rank-0-only
            # evaluation, ranks loading at different times, one rank re-uploading a weight would deadlock or mis-pair with
            # gather_rewards), and it would be redundant -- probe rows and weights are identical on every rank, so are the distances.
            multi = source != "probe" and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            tried, chosen = {}, "strict"
            for name, args in cands[:-1]:
                d = 0.0
                for a, st in zip(score(args), strict):
                    x = float((a - st).abs().max()) if a.numel() else 0.0
                    d = float("inf") if x != x else max(d, x)
                if multi:
                    # every rank must end in the same form (rewards are bit-identical across shardings only then): the probe rows are
                    # the same on every rank by construction; user batches should be too -- the largest distance any rank saw decides
                    t = torch.tensor([min(d, 3.0e38)], dtype=torch.float64, device=self.device if dist.get_backend() == "nccl" else "cpu")
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    d = float(t.item())
                tried[name] = d
                if d <= budget:
                    chosen = name
                    break
            self.operand_form = chosen
            self.form_info = {"form": chosen, "default_vs_strict": tried.get("default"), "distance_to_strict": tried, "source": source,
                              "rows": int(sum(a.shape[0] for a in strict)), "budget": budget, "strict_noise_floor": floor if source == "probe" else None,
                              "seconds": _now(self.device) - t_start}
            return dict(self.form_info)
        finally:
            self._in_probe = False
            self.training = was_training
            self._apply_form()

    def _lock_operand_form(self) -> None:
        """Runs in .to('cuda') and again whenever the engine's weights have changed since (lr_weights_epoch): the default form is
        checked against the strict form on the seeded probe rows of probe.py -- a function of the weights and the engine's capacity
        alone, so every rank, batch and shard of a deployment locks the same form -- and the engine keeps the cheapest form of
        _form_candidates whose probe rewards all sit within `parity_budget` of the strict form's (the strict form if none does).
        Measured on the full-size synthetic weight sets, default form, max over the 4 probe rows: benign Phi-3.5-V 3e-5 .. 1.9e-4,
        LLaVA-7B 0.9 .. 1.5e-4, Qwen2.5-VL-7B 1.6 .. 2.3e-4 (their default-form errors against the reference on the golden rows:
        <= 1e-4); outlier-bearing Phi-3.5-V 2.4e-3 .. 1.6e-2 (golden rows 4.8e-4 .. 2.6e-3), with adapters 3.4e-3.  The budget is HALF
        the bound the golden tests hold the locked form to (3e-4, itself a third of the 1e-3 bar): a form whose 4 probe rows sit
        within 1.7e-4 of the strict form still landed 3.3e-4 from the reference on a golden row (the rows are draws of one noise)."""
        epoch = self.engine.weights_epoch()
        if self._opts["operand_dtype"] != "f16x2f8":
            self._form_epoch = epoch
            return
        if self.pinned_form is not None:
            self.operand_form = self.pinned_form
            self.form_info = {"form": self.pinned_form, "default_vs_strict": None, "source": "pinned (operand_form=...)", "rows": 0,
                              "budget": self.parity_budget}
            self._apply_form()
            self._form_epoch = epoch
            return
        self.operand_form, self.form_info = "default", None
        if not self.auto_calibrate:
            self._apply_form()
            self._form_epoch = epoch
            return
        try:
            from .probe import probe_batches
            batches = probe_batches(self)
            if not batches:            # an engine too small for any probe row: nothing to measure on, stay on the safe side
                self.operand_form = "strict"
                self.form_info = {"form": "strict", "default_vs_strict": None, "source": "no probe row fits the engine's capacity", "rows": 0,
                                  "budget": self.parity_budget}
                self._apply_form()
            else:
                self._compare_forms(batches, self.parity_budget, "probe")
        except BaseException:
            # a probe forward failed (OOM, a rejected batch): the form was NOT checked on these weights.  Run the safe form, and leave
            # _form_epoch behind so that the next forward tries the probe again instead of scoring un-checked in the cheapest form.
            self.operand_form = "strict"
            self.form_info = {"form": "strict", "default_vs_strict": None, "source": "probe failed", "rows": 0, "budget": self.parity_budget}
            self._apply_form()
            raise
        self._form_epoch = epoch

    def calibrate(self, *batches, parity_budget: float = 2.5e-4) -> Dict[str, object]:
        """Refinement of the automatic check of .to('cuda') on the caller's OWN data: `batches` (a few representative dicts of
        custom_forward keyword arguments) are scored in the strict form (16-bit residual passes everywhere, 22 bits per operand,
        1.17x the step time; measured <= 6e-6 from the fp32 reference on every full-size golden, outlier-bearing weights included) and in
        the cheaper forms, cheapest first; the first one within `parity_budget` of the strict rewards is locked -- whatever the
        automatic probe had chosen.  Returns {"form", "default_vs_strict", "distance_to_strict", "source", "rows", "budget"}.  Static afterwards (until the weights change): a row's reward stays independent of the batch it is scored in.
        Under torch.distributed call it on every rank with the SAME batches (the decision is all-reduced).  No reference
        counterpart (the reference runs fp32 / bf16 operands)."""
        if self.engine is None:
            raise RuntimeError("calibrate: model is on CPU; call model.to('cuda') first")
        if self._opts["operand_dtype"] != "f16x2f8":
            raise ValueError("calibrate applies to the default parity form (operand_dtype='f16x2f8')")
        self._form_epoch = self.engine.weights_epoch()
        return self._compare_forms(list(batches), parity_budget, "calibrate")

    def custom_forward(self, input_ids=None, attention_mask=None, pixel_values=None, image_sizes=None,
                       return_output=False, inputs_batch=None):
        """rw_model_general_preference.py:334-448, phi3v branch (positional order kept:
        eval/batch_inference_rm_phi.py:93)."""
        if self.engine is None:
            raise RuntimeError("custom_forward: model is on CPU; call model.to('cuda') first "
                               "(the scoring path has no CPU fallback)")
        if not self._in_probe and self._form_epoch != self.engine.weights_epoch():
            self._lock_operand_form()         # weights re-uploaded / re-synthesised since the form was locked: check it again on them
        if self.model_type == "qwen":
            return self._custom_forward_qwen(inputs_batch, return_output)
        if inputs_batch is not None and input_ids is None:
            input_ids = inputs_batch["input_ids"]
            attention_mask = inputs_batch["attention_mask"]
            pixel_values = inputs_batch["pixel_values"]
            image_sizes = inputs_batch["image_sizes"]
        if pixel_values is None or image_sizes is None:
            # the reference cannot run text-only rows either (modeling_phi3_v.py:252 unbound local)
            raise UnboundLocalError("img_token_batch_embedding: every row must carry an image")
        if input_ids.dim() == 3:
            input_ids = input_ids.squeeze(1)
        sz = torch.as_tensor(image_sizes).cpu().long()        # the processors return host tensors: no copy, no stream drain
        if self.check_inputs == "eager" or not input_ids.is_cuda:
            # host tensors are counted on the host, BEFORE the H2D copy; device tensors drain the stream here (check_inputs="deferred"
            # leaves the check to the engine: NaN rewards for such rows)
            if self.model_type == "llava":
                # rw_model:372-375 -> LlavaNext forward: image_sizes are the ORIGINAL (h, w); slots are image_token_id
                n_slots = (input_ids == self.config.image_token_id).sum(dim=1).cpu()
                expect = torch.tensor([llava_geometry(int(h), int(w), self.config.pinpoints, self.config.clip.image,
                                                      self.config.clip.grid)[6] for h, w in sz.tolist()])
                if not torch.equal(n_slots.long(), expect):
                    raise ValueError(f"Image features and image tokens do not match, tokens: {n_slots.tolist()}, "
                                     f"features: {expect.tolist()}")          # modeling_llava_next.py get_placeholder_mask
            else:
                n_slots = (input_ids < 0).sum(dim=1).cpu()
                g2 = self.config.clip.grid // 2
                img = self.config.clip.image
                expect = (sz[:, 0] // img) * g2 * ((sz[:, 1] // img) * g2 + 1) + 1 + g2 * (g2 + 1)
                if not torch.equal(n_slots.long(), expect):
                    raise RuntimeError(f"shape mismatch: image slots per row {n_slots.tolist()} != projected image tokens "
                                       f"{expect.tolist()} (modeling_phi3_v.py:247 index_put)")
        inner = self.model_type == "phi3v" and self.layer_id != 32 and self.layer_id < self.config.layers
        self.engine.set_layer_limits(-1, self.layer_id if inner else -1)
        reward = self.engine.forward(input_ids, attention_mask, pixel_values, sz, training=self.training, no_final_norm=inner,
                                     keep_hidden_states=return_output or self.keep_hidden_states)
        return self._finish(reward, input_ids.shape, return_output, inner,
                            lazy_args=(input_ids, attention_mask, pixel_values, sz) if return_output and not inner else None)

    def _finish(self, reward, shape, return_output, inner=False, lazy_args=None):
        """Return convention of rw_model:407-448.  Train mode without mean pooling reads the LAST position (left-padded batches) and
        the BT head then returns [B] (`values.squeeze(-1)[:, -1]`, :413-415) where eval returns [B, 1] (:420-421); GPM returns
        [B, d] in both (:434, :439-444).  `outputs` (return_output=True, :422-425): the reference hands back the backbone's whole
        output object; its callers read `last_hidden_state` (trainer evaluate, rm_trainer_general_preference.py:414), so that is
        what is materialised here -- [B, S, hidden] fp32, made on demand from the engine's residual stream -- beside the hidden row
        the reward was read from."""
        if self.training and not self.is_general_preference and not self.mean_hidden_state:
            reward = reward.squeeze(-1)
        if self.reward_dtype is not None and not self._in_probe:
            # (never inside the operand-form check: its budget, 1.5e-4, is far below a bf16 ulp of the rewards it compares)
            reward = reward.to(self.reward_dtype)
        if not return_output:
            return reward, None
        B, S = int(shape[0]), int(shape[1])
        D = self.config.hidden
        # the hidden row behind each reward (after final norm / SkipCA input); the mean-pooling head has no such row
        hl = None if self.mean_hidden_state else torch.from_numpy(self.engine.read_tap("hL", B * D).reshape(B, D).copy())
        if inner:
            # layer_id != 32: the reference's outputs["last_hidden_state"] would still be the final norm of the FULL stack, which this
            # forward (stopped after layer_id layers) never computed -- so the key is not offered; what the reward was read from,
            # hidden_states[layer_id] (rw_model:351-352), comes back under its own name
            return reward, _Outputs({"hidden_states_at_layer_id": self.engine.last_hidden_state(B, S, no_final_norm=True),
                                     "last_hidden_state_at_reward_token": hl})
        out = _Outputs({"last_hidden_state": self.engine.last_hidden_state(B, S, no_final_norm=False),
                        "last_hidden_state_at_reward_token": hl})
        if lazy_args is not None:
            out["hidden_states"] = _LazyHiddenStates(self, lazy_args, shape, out["last_hidden_state"],
                                                     self.engine.vision_embeds(B) if self.model_type == "phi3v" else None)
        return reward, out

    def _custom_forward_qwen(self, inputs_batch, return_output):
        """rw_model_general_preference.py:354-371: the qwen branch reads everything from `inputs_batch` (the
        processor's BatchFeature: input_ids, attention_mask, pixel_values [sum t*h*w, 1176], image_grid_thw)."""
        if inputs_batch is None:
            raise TypeError("'NoneType' object is not subscriptable (model_type == 'qwen' takes inputs_batch=..., rw_model:355)")
        ids = inputs_batch["input_ids"]
        mask = inputs_batch["attention_mask"]
        pix = inputs_batch["pixel_values"]
        grid = torch.as_tensor(inputs_batch["image_grid_thw"]).cpu().long()
        unit = self.config.vision.merge_unit
        if bool((grid[:, 0] != 1).any()):
            # rw_model:356 hands image_grid_thw to the ViT as it is; the HF processor gives every IMAGE t = 1 (a still is its own temporal
            # patch pair) and multi-frame grids travel under other keys (pixel_values_videos / video_grid_thw) that custom_forward never
            # reads -- so t > 1 under image_grid_thw is a malformed batch, refused here by name instead of deep inside the engine
            raise ValueError(f"image_grid_thw[:, 0] must be 1 (still images; got t = {grid[:, 0].tolist()}): video grids are not served by the "
                             "qwen branch of custom_forward (rw_model_general_preference.py:354-357 reads image inputs only)")
        # (always checked, check_inputs="deferred" included: lr_forward_qwen has no slot check of its own -- a mismatch would neither
        #  raise nor come back NaN -- and one scalar per forward is all this costs)
        n_slots = int((ids == self.config.image_token_id).sum())
        n_feat = int(grid.prod(dim=1).sum()) // unit
        if n_slots != n_feat:
            raise ValueError(f"Image features and image tokens do not match, tokens: {n_slots}, features: {n_feat}")
        reward = self.engine.forward_qwen(ids, mask, pix, grid, training=self.training,
                                          keep_hidden_states=return_output or self.keep_hidden_states)
        return self._finish(reward, ids.shape, return_output)

    __call__ = custom_forward
