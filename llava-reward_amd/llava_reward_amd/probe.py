"""The seeded probe batch behind the operand-form self-check that `RewardModel.to('cuda')` runs (model.py `_lock_operand_form`).

No reference counterpart: the reference runs one operand type (bf16 on the GPU, fp32 on the CPU).  Here the default parity form
(f16 hi + e4m3 residual passes, ~15 bits per operand) is only as good as the loaded weights let it be -- weight sets that amplify
operand rounding (massive activations, large norm gains: what trained checkpoints show) need the strict form (16-bit residual
passes, 22 bits) -- so the engine measures the distance between the two forms on the weights it was given, on rows that are a
function of (model geometry, engine capacity) ALONE: every rank, every shard and every batch size of one deployment sees the same
rows and therefore locks the same form.

Rows: `PROBE_ROWS` full-length rows in chunks of `PROBE_CHUNK` (the largest crop grid / image that fits the engine's capacity,
captions of 128 / 96 / 64 / 33 and 112 / 80 / 48 / 17 tokens: left padding and the long-sequence regime are both inside), token ids from the counter hash of synth.py, pixels filled in
HBM by the same hash (lr_op_synth_fill), std 1 like CLIP-normalised images."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import numpy as np
import torch

from . import _lib as L
from . import synth

PROBE_SEED = 0x5EED0F0A
PROBE_ROWS = 8                   # round 5: 8 rows (4 until then): the decision is the MAX over the rows, and rows are draws of one noise
PROBE_CHUNK = 4                  # scored 4 at a time: the rows (their left padding) do not depend on max_batch once it is >= 4
PROBE_CAPTIONS = (128, 96, 64, 33, 112, 80, 48, 17)


def _fill(lib, t: torch.Tensor, name: str) -> None:
    """t <- seeded noise of std 1 (fp32, device), generated in place by synth_fill_kernel."""
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    stream = torch.cuda.current_stream(t.device).cuda_stream
    rc = lib.lr_op_synth_fill(C.c_void_p(t.data_ptr()), t.numel(), C.c_uint64(PROBE_SEED), name.encode(), 1.0, 0.0, 0,
                              C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"lr_op_synth_fill failed ({rc})")


def _captions(room: int, n: int) -> List[int]:
    return [max(1, min(c, room)) for c in PROBE_CAPTIONS[:n]] + [max(1, min(17, room))] * max(0, n - len(PROBE_CAPTIONS))


def _chunks(rows: int, cap: int):
    cap = max(1, min(rows, cap))
    return [(i, min(i + cap, rows)) for i in range(0, rows, cap)]


def probe_batches(model, rows: int = PROBE_ROWS) -> List[Dict[str, object]]:
    """-> a list of custom_forward keyword dicts covering `rows` probe rows in chunks of at most max_batch rows."""
    cfg, opts, dev = model.config, model._opts, model.device
    lib = L.load()
    mb, ms = int(opts["max_batch"]), int(opts["max_seq"])
    out = []
    if model.model_type == "phi3v":
        grid = None
        for hc, wc in ((4, 4), (3, 4), (3, 3), (2, 3), (2, 2), (1, 2), (1, 1)):
            if hc * wc + 1 <= int(opts["max_crops"]) and synth.num_img_tokens(336 * hc, 336 * wc) + 5 + 1 <= ms:
                grid = (hc, wc)
                break
        if grid is None:
            return []
        room = ms - (synth.num_img_tokens(336 * grid[0], 336 * grid[1]) + 5)
        caps = _captions(room, rows)
        b = synth.synth_batch(cfg, PROBE_SEED, caps, grid, with_pixels=False)
        ncr = grid[0] * grid[1] + 1
        for lo, hi in _chunks(rows, min(mb, PROBE_CHUNK)):
            pix = torch.empty(hi - lo, ncr, 3, cfg.clip.image, cfg.clip.image, device=dev, dtype=torch.float32)
            for r in range(lo, hi):
                _fill(lib, pix[r - lo], f"probe.pixel_values.{r}")
            out.append(dict(input_ids=torch.from_numpy(b["input_ids"][lo:hi]).to(dev), attention_mask=torch.from_numpy(b["attention_mask"][lo:hi]).to(dev),
                            pixel_values=pix, image_sizes=torch.from_numpy(b["image_sizes"][lo:hi])))
        return out
    if model.model_type == "llava":
        size = None
        for h, w in ((672, 672), (336, 672), (336, 336), (200, 200), (100, 100)):
            g = synth.llava_geometry(h, w, cfg.pinpoints, cfg.clip.image, cfg.clip.grid)
            if 1 + g[0] * g[1] <= int(opts["max_crops"]) and g[6] + 5 + 1 <= ms:
                size = (h, w)
                break
        if size is None:
            return []
        g = synth.llava_geometry(size[0], size[1], cfg.pinpoints, cfg.clip.image, cfg.clip.grid)
        ncr = 1 + g[0] * g[1]
        caps = _captions(ms - (g[6] + 5), rows)
        b = synth.llava_synth_batch(cfg, PROBE_SEED, caps, [size] * rows, with_pixels=False)
        for lo, hi in _chunks(rows, min(mb, PROBE_CHUNK)):
            pix = torch.empty(hi - lo, ncr, 3, cfg.clip.image, cfg.clip.image, device=dev, dtype=torch.float32)
            for r in range(lo, hi):
                _fill(lib, pix[r - lo], f"probe.pixel_values.{r}")
            out.append(dict(inputs_batch=dict(input_ids=torch.from_numpy(b["input_ids"][lo:hi]).to(dev),
                                              attention_mask=torch.from_numpy(b["attention_mask"][lo:hi]).to(dev), pixel_values=pix,
                                              image_sizes=torch.from_numpy(b["image_sizes"][lo:hi]))))
        return out
    # qwen: one image of g x g patches per row (g even)
    v = cfg.vision
    mp = int(model.engine.max_patches) if model.engine is not None else int(opts.get("max_patches", 0))
    grid = None
    for gsz in (32, 24, 16, 12, 8, 4):
        per_row = gsz * gsz
        chunk = max(1, min(rows, mb, PROBE_CHUNK))
        if per_row * chunk <= mp and per_row // v.merge_unit + 5 + 1 <= ms:
            grid = (gsz, gsz)
            break
    if grid is None:
        return []
    caps = _captions(ms - (grid[0] * grid[1] // v.merge_unit + 5), rows)
    b = synth.qwen_synth_batch(cfg, PROBE_SEED, caps, grid, with_pixels=False)
    per_row = grid[0] * grid[1]
    for lo, hi in _chunks(rows, min(mb, PROBE_CHUNK)):
        pix = torch.empty((hi - lo) * per_row, v.patch_dim, device=dev, dtype=torch.float32)
        for r in range(lo, hi):
            _fill(lib, pix[(r - lo) * per_row:(r - lo + 1) * per_row], f"probe.pixel_values.{r}")
        out.append(dict(inputs_batch=dict(input_ids=torch.from_numpy(b["input_ids"][lo:hi]).to(dev),
                                          attention_mask=torch.from_numpy(b["attention_mask"][lo:hi]).to(dev), pixel_values=pix,
                                          image_grid_thw=torch.from_numpy(b["image_grid_thw"][lo:hi]))))
    return out
