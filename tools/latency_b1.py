#!/usr/bin/env python3
"""Latency of ONE row (BASELINE configs[0] shape: B=1, 17 crops, S~2642) through custom_forward: wall time per call with and without
back-to-back enqueueing, to see how much of it is launch overhead (1700+ kernel launches per pass)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
import torch
from llava_reward_amd import synth
from llava_reward_amd.model import RewardModel
cfg = synth.full_config()
b = synth.synth_batch(cfg, 1234, [128], (4, 4), with_pixels=False)
ids, mask = torch.from_numpy(b["input_ids"]).cuda(), torch.from_numpy(b["attention_mask"]).cuda()
pix = torch.randn(1, 17, 3, 336, 336, device="cuda")
sizes = torch.from_numpy(b["image_sizes"])
m = RewardModel(cfg, synth_seed=1234, max_batch=1, max_seq=ids.shape[1], max_crops=17).to("cuda").eval()
for _ in range(2):
    m.engine.forward(ids, mask, pix, sizes)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    m.engine.forward(ids, mask, pix, sizes); torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(10):
    m.engine.forward(ids, mask, pix, sizes)
torch.cuda.synchronize()
t2 = time.perf_counter()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    m.engine.forward(ids, mask, pix, sizes)
e1.record(); torch.cuda.synchronize()
t_host = time.perf_counter()
for _ in range(3):
    pass
print(f"B=1: {1e2 * (t1 - t0):.2f} ms per call with a sync after each; {1e2 * (t2 - t1):.2f} ms per call enqueued back to back; GPU time {e0.elapsed_time(e1) / 10:.2f} ms")
import cProfile
t3 = time.perf_counter(); m.engine.forward(ids, mask, pix, sizes); t4 = time.perf_counter(); torch.cuda.synchronize()
print(f"host-side enqueue time of one call: {1e3 * (t4 - t3):.2f} ms")
