"""Training-side reuse of the HIP scoring path: the two methods of the reference's
GeneralPreferenceRewardTrainer that only SCORE -- `concatenated_forward`
(llava_reward/trainer/rm_trainer_general_preference.py:447-460) and the `evaluate` loop (:381-445) -- as free functions over a
llava_reward_amd RewardModel, so a trainer can route its evaluation passes (and reward-guided sampling, README.md:116) through the
engine while the optimisation step, the losses and the datasets stay the reference's.  Same argument order, same return values.

    chosen, rejected, outputs = concatenated_forward(model, chosen_ids, c_mask, c_pix, c_size, reject_ids, r_mask, r_pix, r_size)
    stats = evaluate(model, eval_dataloader, loss_fn)          # {"eval_loss_mean": ..., "prob_mean": ...}

Train-mode conventions (`model.train()`, rw_model_general_preference.py:410-415, :429-434) are the model object's: the reward is read
at the LAST position (the trainer's batches are left-padded) and the BT head returns [B] instead of [B, 1]."""
from __future__ import annotations

from typing import Callable, Dict, Iterable, Optional

import torch


def concatenated_forward(model, chosen_ids=None, c_mask=None, c_pixel_value=None, c_img_size=None, reject_ids=None, r_mask=None,
                         r_pixel_value=None, r_img_size=None, inputs_batch_c=None, inputs_batch_r=None, return_output: bool = False):
    """Score the chosen and the rejected rows of a batch: two custom_forward calls (the reference does not concatenate either,
    despite the name) -> (chosen_rewards, rejected_rewards, [chosen_outputs, rejected_outputs])."""
    model = getattr(model, "module", model)              # DeepSpeed / DDP wrappers
    if model.model_type == "phi3v":
        c, co = model.custom_forward(input_ids=chosen_ids, attention_mask=c_mask, pixel_values=c_pixel_value, image_sizes=c_img_size,
                                     return_output=return_output)
        r, ro = model.custom_forward(input_ids=reject_ids, attention_mask=r_mask, pixel_values=r_pixel_value, image_sizes=r_img_size,
                                     return_output=return_output)
    else:                                                # qwen / llava: the processor's BatchFeature (:456-458)
        c, co = model.custom_forward(inputs_batch=inputs_batch_c, return_output=return_output)
        r, ro = model.custom_forward(inputs_batch=inputs_batch_r, return_output=return_output)
    return c, r, [co, ro]


def _squeeze_to(t, device):
    t = torch.as_tensor(t)
    return (t.squeeze(1) if t.dim() > 1 and t.shape[1] == 1 else t).to(device)


@torch.no_grad()
def evaluate(model, eval_dataloader: Iterable, loss_fn: Callable, margin: Optional[torch.Tensor] = None,
             all_reduce: Optional[Callable[[Dict[str, float]], Dict[str, float]]] = None) -> Dict[str, float]:
    """The reference's evaluation pass (:381-445): eval mode, per batch of
    (chosen_ids, c_mask, c_pixel_value, c_img_size, reject_ids, r_mask, r_pixel_value, r_img_size) -- the collate adds a singleton
    dim that is squeezed here as there (:394-402) -- `loss, prob = loss_fn(chosen_reward, reject_reward, margin)`, means over the
    batches, optional cross-rank mean (`strategy.all_reduce`), then `model.train()` -- unconditionally, as the reference resets its
    model state (:441): call `model.eval()` again before scoring with it.  `loss_fn` is the trainer's own loss
    module (PairWiseLoss, GeneralPreferenceLoss, ...: llava_reward/models/loss.py), untouched.
    Return value: a dict on EVERY rank (the reference returns the scalar loss_mean, on rank 0 only, and logs the rest, :436-445)."""
    model = getattr(model, "module", model)              # DeepSpeed / DDP wrappers: mode switches and .device belong to the inner model
    dev = model.device
    model.eval()
    loss_sum, prob_sum, n = 0.0, 0.0, 0
    for batch in eval_dataloader:
        c_ids, c_mask, c_pix, c_size, r_ids, r_mask, r_pix, r_size = batch[:8]
        c, r, _ = concatenated_forward(model, _squeeze_to(c_ids, dev), _squeeze_to(c_mask, dev), _squeeze_to(c_pix, dev), _squeeze_to(c_size, "cpu"),
                                       _squeeze_to(r_ids, dev), _squeeze_to(r_mask, dev), _squeeze_to(r_pix, dev), _squeeze_to(r_size, "cpu"))
        loss, prob = loss_fn(c, r, margin)
        loss_sum += float(loss)
        prob_sum += float(prob)
        n += 1
    stats = {"eval_loss_mean": loss_sum / max(n, 1), "prob_mean": prob_sum / max(n, 1)}
    if all_reduce is not None:
        stats = all_reduce(stats)
    model.train()
    return stats
