"""CPU oracle of the Qwen2-VL image processor: RGB uint8 image -> pixel_values [grid_h * grid_w, 1176] fp32, image_grid_thw.
TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline); the product path (lr_qwen_image_transform, HIP) never imports it.

The processor is third party (transformers, pinned 4.50.0 by the reference's requirements.txt:9; call site
llava_reward/utils/utils.py:34-44: AutoProcessor.from_pretrained(..., min_pixels=256*28*28, max_pixels=1280*28*28), used by
eval/batch_inference_rm_qwen.py).  Restated from the published algorithm (image_processing_qwen2_vl.py: smart_resize, resize
with PIL BICUBIC, rescale = float64(u) * (1/255) -> fp32, normalize = (x - mean) / std in fp32, patchify with
temporal_patch_size 2 / merge_size 2).  Pinned: tests/golden/preq_*.json are digests of the real processor's output
(transformers 5.15 Qwen2VLImageProcessorPil = the same slow-processor arithmetic) made by tests/golden/make_preprocess_goldens.py,
and the CPU test re-runs that processor where it imports."""
from __future__ import annotations

import math

import numpy as np

from .phi3v_hd_transform_oracle import CLIP_MEAN, CLIP_STD, resize_u8


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56, max_pixels: int = 14 * 14 * 4 * 1280):
    if max(height, width) / min(height, width) > 200:
        raise ValueError("absolute aspect ratio must be smaller than 200")
    h_bar = round(height / factor) * factor
    w_bar = round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def preprocess(img: np.ndarray, min_pixels: int = 256 * 28 * 28, max_pixels: int = 1280 * 28 * 28):
    """One RGB uint8 [h, w, 3] image -> (pixel_values [gh*gw, 1176] f32, (1, gh, gw))."""
    h, w, _ = img.shape
    oh, ow = smart_resize(h, w, 28, min_pixels, max_pixels)
    r = resize_u8(img, oh, ow, "bicubic")
    x = (r.astype(np.float64) * (1 / 255)).astype(np.float32).transpose(2, 0, 1)            # rescale, channels first
    x = (x - np.array(CLIP_MEAN, dtype=np.float32)[:, None, None]) / np.array(CLIP_STD, dtype=np.float32)[:, None, None]
    gh, gw = oh // 14, ow // 14
    p = x.reshape(3, gh // 2, 2, 14, gw // 2, 2, 14).transpose(1, 4, 2, 5, 0, 3, 6)        # (gh/2, gw/2, 2, 2, C, 14, 14)
    p = np.broadcast_to(p[:, :, :, :, :, None, :, :], p.shape[:5] + (2,) + p.shape[5:])   # the frame fills both temporal slots
    return np.ascontiguousarray(p.reshape(gh * gw, 3 * 2 * 14 * 14)).astype(np.float32), (1, gh, gw)
