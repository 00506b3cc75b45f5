#!/usr/bin/env python3
"""Image hand-over throughput: lr_hd_transform on the GPU (uint8 source resident in HBM, HIP-event timed) beside the CPU
primitives the reference's processor runs (Pillow resize + torch normalise / bicubic / tiling, one thread per image as in
the reference's DataLoader worker).  python3 tools/preprocess_bench.py [h w num_crops]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from llava_reward_amd import _lib as L, synth  # noqa: E402


def main():
    h, w, nc = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (336, 336, 16)
    lib = L.load()
    a = synth.synth_image(1234, "bench", h, w)
    src = torch.from_numpy(a).cuda()
    B = 32
    out = torch.empty(B, nc + 1, 3, 336, 336, device="cuda")
    need = lib.lr_hd_transform_workspace(h, w, nc)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream()

    def batch():
        for b in range(B):
            rc = lib.lr_hd_transform(C.c_void_p(src.data_ptr()), h, w, nc, C.c_void_p(out[b].data_ptr()), None, None,
                                     C.c_void_p(ws.data_ptr()), need, C.c_void_p(st.cuda_stream))
            assert rc == 0
    batch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record(st)
    for _ in range(reps):
        batch()
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (reps * B)
    size = (C.c_int64 * 2)()
    lib.lr_hd_transform(C.c_void_p(src.data_ptr()), h, w, nc, C.c_void_p(out[0].data_ptr()), size, None, C.c_void_p(ws.data_ptr()), need,
                        C.c_void_p(st.cuda_stream))
    H, W = size[0], size[1]
    algo = (nc + 1) * 3 * 336 * 336 * 4 + h * w * 3 + H * W * 3 * 2          # fp32 out + source + resized image (written, read)
    print(f"GPU  {h}x{w} -> {H}x{W}, {nc + 1} crops: {ms * 1e3:8.1f} us/image = {1e3 / ms:9.0f} images/s, "
          f"{algo / ms / 1e6:7.1f} GB/s of algorithmic bytes ({algo / 1e6:.1f} MB/image)")
    from make_preprocess_goldens import pipeline
    torch.set_num_threads(1)
    pipeline(a, nc)
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        pipeline(a, nc)
    cpu = (time.perf_counter() - t0) / n
    print(f"CPU  Pillow + torch primitives, 1 thread: {cpu * 1e3:8.1f} ms/image = {1 / cpu:6.1f} images/s  -> GPU/CPU-thread {cpu * 1e3 / ms:.0f}x")


if __name__ == "__main__":
    main()
