"""Batch scoring loop and its multi-GPU sharding.

Counterpart of the reference's eval/batch_inference_rm_phi.py:79-152, batch_inference_rm_qwen.py:76-133 and
batch_inference_rm_llava.py (pairwise and non-pairwise modes), which run on one GPU (DistributedSampler(num_replicas=1, rank=0), :50-57).  Here rows are
independent units (custom_forward has no cross-sample op), so a global batch is cut into contiguous
per-rank shards and the only collective is one all-gather of the fp32 rewards per batch
(SURVEY.md §8e).  A preference pair's chosen and rejected rows go to the same rank.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .reward_adaptor_loader import preference_compute


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_rows(n: int, rank: int, world_size: int) -> slice:
    """Contiguous shard [lo, hi) of n rows for `rank`; sizes differ by at most one and concatenating the
    shards in rank order restores the input order."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return slice(lo, lo + base + (1 if rank < rem else 0))


def _all_gather(local: torch.Tensor, n_total: Optional[int], ws: int) -> torch.Tensor:
    n_local = local.shape[0]
    if n_total is None:
        out = torch.empty((ws * n_local,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    sizes = [shard_rows(n_total, r, ws) for r in range(ws)]
    cap = max(s.stop - s.start for s in sizes)
    buf = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[:n_local] = local
    outs = [torch.empty_like(buf) for _ in range(ws)]
    dist.all_gather(outs, buf)
    return torch.cat([o[: s.stop - s.start] for o, s in zip(outs, sizes)], dim=0)


_side_streams: Dict[int, "torch.cuda.Stream"] = {}


class GatherHandle:
    """An all-gather of rewards in flight on the collective stream; `.wait()` makes the caller's stream depend on it and returns
    the gathered [n_total, d] tensor."""

    def __init__(self, out, stream=None):
        self._out, self._stream = out, stream

    def wait(self) -> torch.Tensor:
        if self._stream is not None:
            cur = torch.cuda.current_stream(self._out.device)
            cur.wait_stream(self._stream)
            self._out.record_stream(cur)
            self._stream = None
        return self._out


def gather_rewards_async(local: torch.Tensor, n_total: Optional[int] = None, always_collective: bool = False) -> GatherHandle:
    """All-gather per-rank rewards [n_r, d] -> [n_total, d] on every rank; ragged shards are padded to the largest shard for the
    collective.  Device tensors under RCCL (backend "nccl"): the collective is enqueued on a dedicated stream behind an event of the
    compute stream, so the next forward (the rejected rows of a pair) is not ordered behind it.  Device tensors
    under a backend without device collectives (gloo) are staged through the host; host tensors gather as they are.
    always_collective: enter the collective even in a world of one rank (tests: the RCCL / side-stream branch on a single GPU)."""
    rank, ws = world()
    if ws == 1 and not (always_collective and dist.is_available() and dist.is_initialized()):
        return GatherHandle(local)
    if local.is_cuda and dist.get_backend() != "nccl":
        return GatherHandle(_all_gather(local.cpu(), n_total, ws).to(local.device))
    if not local.is_cuda:
        return GatherHandle(_all_gather(local, n_total, ws))
    dev = local.device
    side = _side_streams.get(dev.index)
    if side is None:
        side = _side_streams[dev.index] = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        out = _all_gather(local, n_total, ws)
    local.record_stream(side)
    return GatherHandle(out, side)


def gather_rewards(local: torch.Tensor, n_total: Optional[int] = None) -> torch.Tensor:
    return gather_rewards_async(local, n_total).wait()


def _squeeze(b: Dict[str, torch.Tensor], rows: slice, device) -> Tuple[torch.Tensor, ...]:
    def f(k):
        t = b[k]
        t = t.squeeze(1) if t.dim() > (2 if k != "pixel_values" else 5) else t      # collate adds a dim (:82-90)
        return t[rows].to(device)
    return f("input_ids"), f("attention_mask"), f("pixel_values"), f("image_sizes")


def shard_qwen_batch(b: Dict[str, torch.Tensor], rows: slice, image_token_id: int, merge_unit: int) -> Dict[str, torch.Tensor]:
    """Rows [rows] of a Qwen2.5-VL BatchFeature: pixel_values [sum t*h*w, 1176] and image_grid_thw [n_images, 3] are
    concatenated over the batch in row order, so the images of a row are found by consuming its image-slot count."""
    ids = b["input_ids"]
    grid = torch.as_tensor(b["image_grid_thw"]).cpu().long()
    slots = (ids == image_token_id).sum(dim=1).cpu().tolist()
    per_img = (grid.prod(dim=1) // merge_unit).tolist()
    first, k = [], 0
    for n in slots:                                   # first image of every row
        first.append(k)
        while n > 0:
            if k >= len(per_img) or per_img[k] > n:
                raise ValueError("Image features and image tokens do not match")
            n -= per_img[k]
            k += 1
    first.append(k)
    lo, hi = rows.start, rows.stop
    patches = grid.prod(dim=1)
    p0, p1 = int(patches[: first[lo]].sum()), int(patches[: first[hi]].sum())
    return {"input_ids": ids[lo:hi], "attention_mask": b["attention_mask"][lo:hi], "pixel_values": b["pixel_values"][p0:p1],
            "image_grid_thw": grid[first[lo]: first[hi]]}


def _forward_rows(model, b, rows: slice, device):
    """custom_forward on rows [rows] of a collated batch, whichever backbone the model is (rw_model:343-375)."""
    mt = getattr(model, "model_type", "phi3v")
    if rows.stop <= rows.start:
        # fewer rows than ranks (the last partial batch of a drop_last=False loader): this rank has nothing to score, but it must
        # still enter the all-gather; lr_forward rejects B = 0, so the empty result is made here
        return torch.empty((0, int(getattr(model, "value_head_dim", 1))), dtype=torch.float32, device=device)
    if mt == "phi3v":
        return model.custom_forward(*_squeeze(b, rows, device))[0]
    if mt == "qwen":
        sb = shard_qwen_batch(b, rows, model.config.image_token_id, model.config.vision.merge_unit)
    else:
        sb = {k: b[k][rows] for k in ("input_ids", "attention_mask", "pixel_values", "image_sizes")}
    sb = {k: (v.to(device) if k != "image_grid_thw" else v) for k, v in sb.items()}
    return model.custom_forward(inputs_batch=sb)[0]


@torch.no_grad()
def score_candidates(model, tokenizer, captions, images, num_crops: int = 16, batch_size: int = 32, pad_token_id: Optional[int] = None,
                     device=None) -> torch.Tensor:
    """Rewards of N (caption, image) candidates, [N, value_head_dim] fp32 on the model's device, in input order: the reward function
    of reward-guided sampling (the Fk-steering use of the reference's README: a population of candidate images per prompt, scored at
    every resampling step).  captions: one string for all candidates or one per candidate; images: paths, PIL images or RGB uint8
    [h, w, 3] arrays / tensors (device tensors never leave the GPU: HD transform, left-pad collate and forward all run there).
    Phi-3.5-V backbone.  With torch.distributed initialised the candidates are sharded by rows (contiguous, before any
    preprocessing) and the rewards all-gathered, bit-identical to the single-process result."""
    from .preprocess import batch_inference_process_phi3v_device
    if getattr(model, "model_type", "phi3v") != "phi3v":
        raise NotImplementedError("score_candidates builds Phi-3.5-V inputs; use the model's own processor + custom_forward(inputs_batch=...)")
    n = len(images)
    caps = [captions] * n if isinstance(captions, str) else list(captions)
    if len(caps) != n:
        raise ValueError(f"{len(caps)} captions for {n} images")
    rank, ws = world()
    device = device or model.device
    rows = shard_rows(n, rank, ws)
    d = int(getattr(model, "value_head_dim", 1))
    was_training = bool(getattr(model, "training", False))
    if was_training:
        model.eval()                     # the EOS-position convention of inference (rw_model:416-425)
    out = [torch.empty((0, d), dtype=torch.float32, device=device)]
    try:
        for lo in range(rows.start, rows.stop, batch_size):
            hi = min(lo + batch_size, rows.stop)
            batch = batch_inference_process_phi3v_device(None, tokenizer, list(zip(images[lo:hi], caps[lo:hi])), device=device,
                                                         num_crops=num_crops, pad_token_id=pad_token_id)
            out.append(model.custom_forward(**batch)[0].float().reshape(hi - lo, d))
    finally:
        if was_training:
            model.train()
    return gather_rewards(torch.cat(out, dim=0), n)


@torch.no_grad()
def score_pairwise(model, args, batches: Iterable, device=None) -> Dict[str, object]:
    """batches yields (inputs_c, inputs_r, c_rates, r_rates) as the reference's DataLoader does.
    Returns the quantities the reference prints: prob_mean, proportion (prob > 0.5), proportion w/o ties."""
    rank, ws = world()
    device = device or model.device
    cs: List[torch.Tensor] = []
    rs: List[torch.Tensor] = []
    for inputs_c, inputs_r, *_ in batches:
        n = inputs_c["input_ids"].shape[0]
        rows = shard_rows(n, rank, ws)
        c = _forward_rows(model, inputs_c, rows, device)
        hc = gather_rewards_async(c, n)                  # in flight while the rejected rows are scored
        r = _forward_rows(model, inputs_r, rows, device)
        # rewards stay on the device: nothing here waits for the GPU, so the host runs ahead and enqueues the next batch (the
        # reference's loop reads every batch back, eval/batch_inference_rm_phi.py:103-112; the engine bounds the run-ahead itself)
        cs.append(hc.wait())
        rs.append(gather_rewards(r, n))
    return _pairwise_stats(args, cs, rs)


def _pairwise_stats(args, cs: List[torch.Tensor], rs: List[torch.Tensor]) -> Dict[str, object]:
    """The quantities the reference prints (eval/batch_inference_rm_phi.py:103-121) from per-batch reward tensors: ONE transfer."""
    d = int(getattr(args, "value_head_dim", 1)) if args.is_general_preference else 1
    c = torch.cat(cs, dim=0) if cs else torch.empty(0, d)
    r = torch.cat(rs, dim=0) if rs else torch.empty(0, d)
    all_probs: List[float] = preference_compute(args, c, r).tolist() if c.shape[0] else []
    chosen_list: List[float] = [] if args.is_general_preference else c.squeeze(-1).tolist()
    reject_list: List[float] = [] if args.is_general_preference else r.squeeze(-1).tolist()
    total = len(all_probs)
    gt = sum(1 for x in all_probs if x > 0.5)
    ties = sum(1 for x in all_probs if x == 0.5)
    return {"prob_mean": sum(all_probs) / max(total, 1), "proportion": gt / max(total, 1),
            "proportion_wo_tie": gt / (total - ties) if total - ties else None, "probs": all_probs,
            "chosen_rewards": chosen_list, "reject_rewards": reject_list}


@torch.no_grad()
def score_single(model, args, batches: Iterable, cls_based: bool = False, device=None) -> Dict[str, object]:
    """Non-pairwise mode (:123-152): batches yields (inputs, labels)."""
    if args.is_general_preference:
        raise ValueError("General preference loss-based model is not supported for single image evaluation. "
                         "Please use BT model instead.")
    rank, ws = world()
    device = device or model.device
    rews: List[torch.Tensor] = []
    labels: List[int] = []
    for inputs, lab in batches:
        n = inputs["input_ids"].shape[0]
        rows = shard_rows(n, rank, ws)
        r = _forward_rows(model, inputs, rows, device)
        rews.append(gather_rewards_async(r, n))          # on the collective stream; read back once, after the loop
        labels.extend(torch.as_tensor(lab).tolist())
    rewards: List[float] = torch.cat([h.wait() for h in rews], dim=0).squeeze(-1).tolist() if rews else []
    out: Dict[str, object] = {"rewards": rewards, "labels": labels}
    if cls_based:
        pred = (1.0 / (1.0 + np.exp(-np.asarray(rewards))) >= 0.5).astype(np.int64)
        lab = np.asarray(labels).astype(np.int64)
        tp = int(((pred == 1) & (lab == 1)).sum())
        fp = int(((pred == 1) & (lab == 0)).sum())
        fn = int(((pred == 0) & (lab == 1)).sum())
        out["accuracy"] = float((pred == lab).mean()) if len(lab) else None
        out["recall"] = tp / (tp + fn) if tp + fn else 0.0
        prec = tp / (tp + fp) if tp + fp else 0.0
        out["f1"] = 2 * prec * out["recall"] / (prec + out["recall"]) if prec + out["recall"] else 0.0
    return out


# ------------------------------------------------------------------------------------------------------------------------------
# Files on disk -> rewards, with the input side overlapped (the reference's loop, eval/batch_inference_rm_phi.py:
# 58-65 DataLoader(num_workers=0-ish) + :79-94, prepares every batch on the main thread while the GPU idles)
# ------------------------------------------------------------------------------------------------------------------------------
class PrefetchingBatcher:
    """Iterates over batches of (image, caption) items as `custom_forward(**batch)` dicts on the device.  `items`: a flat list cut
    into batches of `batch_size`, or (batch_size=None) a list of ready-made chunks.  A background thread prepares batches k + 1 ..
    k + depth while batch k is scored: image files are decoded by a small thread pool (PIL releases the GIL), the uint8 pixels cross
    PCIe and go through lr_hd_transform on a SIDE stream, prompts are tokenised and left-padded; the consumer's stream only waits on
    the event recorded behind that work.  Phi-3.5-V rows (batch_inference_process_phi3v_device).
    A consumer that stops early (an exception in custom_forward, a `break`) must call close() -- or use the object as a context
    manager -- so that the producer thread ends and the prefetched device batches are released."""

    def __init__(self, items, tokenizer, batch_size: Optional[int] = 32, num_crops: int = 16, device="cuda", depth: int = 2, workers: int = 4,
                 pad_token_id: Optional[int] = None):
        import queue
        import threading
        items = list(items)
        self.chunks = [list(c) for c in items] if batch_size is None else [items[lo: lo + batch_size] for lo in range(0, len(items), batch_size)]
        self.tok, self.num_crops = tokenizer, num_crops
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.pad = pad_token_id
        self.workers = max(1, int(workers))
        self._q = queue.Queue(maxsize=max(1, int(depth)))
        self._err = None
        self._stop = threading.Event()
        self._stream = torch.cuda.Stream(device=self.device)
        self._thread = threading.Thread(target=self._produce, daemon=True)
        self._thread.start()

    def __len__(self):
        return len(self.chunks)

    def _put(self, item) -> bool:
        """queue.put that gives up when close() was called (a bounded queue nobody reads would block the producer forever)."""
        import queue
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def close(self):
        """Stop the producer thread and drop what it had prefetched.  Idempotent; called by __exit__ and by score_pairwise_files."""
        import queue
        self._stop.set()
        for _ in range(2):               # unblock a producer waiting on the full queue, release the device batches it holds
            while True:
                try:
                    self._q.get_nowait()
                except queue.Empty:
                    break
            self._thread.join(timeout=30)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _produce(self):
        from concurrent.futures import ThreadPoolExecutor
        from .preprocess import _load_rgb, batch_inference_process_phi3v_device
        try:
            torch.cuda.set_device(self.device)
            with ThreadPoolExecutor(max_workers=self.workers) as pool:
                for chunk in self.chunks:
                    if self._stop.is_set():
                        break
                    pixels = list(pool.map(lambda it: _load_rgb(it[0]), chunk))               # decode off the main thread
                    with torch.cuda.stream(self._stream):
                        batch = batch_inference_process_phi3v_device(None, self.tok, [(p, it[1]) for p, it in zip(pixels, chunk)],
                                                                     device=self.device, num_crops=self.num_crops, pad_token_id=self.pad)
                        ev = torch.cuda.Event()
                        ev.record(self._stream)
                    if not self._put((batch, ev)):
                        break
        except BaseException as e:          # surfaced on the consumer's side
            self._err = e
        finally:
            self._put(None)

    def __iter__(self):
        while not self._stop.is_set():
            item = self._q.get()
            if item is None:
                if self._err is not None:
                    raise self._err
                return
            batch, ev = item
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for v in batch.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)
            yield batch


@torch.no_grad()
def score_pairwise_files(model, args, tokenizer, pairs, batch_size: int = 32, num_crops: int = 16, depth: int = 2, workers: int = 4,
                         pad_token_id: Optional[int] = None, device=None) -> Dict[str, object]:
    """The pairwise evaluation loop of eval/batch_inference_rm_phi.py:79-121 from files on disk: `pairs` = (caption, chosen image,
    rejected image) triples (paths, PIL images or uint8 arrays).  Each rank takes a contiguous shard of the pairs BEFORE anything is
    read; decoding, H2D and the HD transform of the next batches run behind the forward of the current one (PrefetchingBatcher:
    the chosen and the rejected rows of a batch are two consecutive prefetched batches); rewards stay on the device until the end.
    Returns the same statistics as score_pairwise."""
    rank, ws = world()
    device = device or model.device
    pairs = list(pairs)
    mine = pairs[shard_rows(len(pairs), rank, ws)]
    chunks = []
    for lo in range(0, len(mine), batch_size):
        part = mine[lo: lo + batch_size]
        chunks.append([(c, cap) for cap, c, _ in part])
        chunks.append([(r, cap) for cap, _, r in part])
    d = int(getattr(model, "value_head_dim", 1))
    cs = [torch.empty((0, d), dtype=torch.float32, device=device)]
    rs = [torch.empty((0, d), dtype=torch.float32, device=device)]
    with PrefetchingBatcher(chunks, tokenizer, batch_size=None, num_crops=num_crops, device=device, depth=depth, workers=workers,
                            pad_token_id=pad_token_id) as batches:          # (closed on any exit: an exception in custom_forward included)
        for i, batch in enumerate(batches):
            (cs if i % 2 == 0 else rs).append(model.custom_forward(**batch)[0].float().reshape(-1, d))
    return _pairwise_stats(args, [gather_rewards(torch.cat(cs, dim=0), len(pairs))], [gather_rewards(torch.cat(rs, dim=0), len(pairs))])
