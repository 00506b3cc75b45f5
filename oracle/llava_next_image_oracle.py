"""CPU oracle of the LLaVA-NeXT image processor: RGB uint8 image -> pixel_values [n_crops, 3, 336, 336] fp32, image_sizes.
TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline); the product path (lr_llava_image_transform, HIP) never imports it.

The processor is third party (transformers, pinned 4.50.0 by the reference's requirements.txt:9; call site
llava_reward/utils/utils.py:46-55: LlavaNextProcessor.from_pretrained, used by eval/batch_inference_rm_llava.py).  Restated
from the published algorithm (image_processing_llava_next.py: select_best_resolution, get_patch_output_size, BICUBIC resize on
uint8 via Pillow, centred zero padding, divide_to_patches, whole image resized to 336x336 in front, rescale = float64(u) * (1/255)
-> fp32, normalize in fp32, zero crops up to the batch maximum).  Pinned: tests/golden/prel_*.json are digests of the real
processor's output (transformers 5.15 LlavaNextImageProcessorPil with llava-v1.6-mistral-7b-hf's preprocessor settings) made
by tests/golden/make_preprocess_goldens.py, and the CPU test re-runs that processor where it imports."""
from __future__ import annotations

import math

import numpy as np

from .phi3v_hd_transform_oracle import CLIP_MEAN, CLIP_STD, resize_u8

PINPOINTS = [[336, 672], [672, 336], [672, 672], [1008, 336], [336, 1008]]       # llava-v1.6-mistral-7b-hf config


def select_best_resolution(original_size, possible_resolutions):
    oh, ow = original_size
    best, max_eff, min_waste = None, 0, float("inf")
    for h, w in possible_resolutions:
        scale = min(w / ow, h / oh)
        dw, dh = int(ow * scale), int(oh * scale)
        eff = min(dw * dh, ow * oh)
        waste = w * h - eff
        if eff > max_eff or (eff == max_eff and waste < min_waste):
            max_eff, min_waste, best = eff, waste, (h, w)
    return best


def patch_output_size(h, w, target):
    th, tw = target
    scale_w, scale_h = tw / w, th / h
    if scale_w < scale_h:
        return min(math.ceil(h * scale_w), th), tw
    return th, min(math.ceil(w * scale_h), tw)


def _norm(u8_hwc: np.ndarray) -> np.ndarray:
    x = (u8_hwc.astype(np.float64) * (1 / 255)).astype(np.float32).transpose(2, 0, 1)
    return (x - np.array(CLIP_MEAN, dtype=np.float32)[:, None, None]) / np.array(CLIP_STD, dtype=np.float32)[:, None, None]


def preprocess(img: np.ndarray, pinpoints=PINPOINTS, max_crops: int = None):
    """One RGB uint8 [h, w, 3] image -> (pixel_values [max_crops or n_crops, 3, 336, 336] f32, (h, w))."""
    h, w, _ = img.shape
    bh, bw = select_best_resolution((h, w), pinpoints)
    nh, nw = patch_output_size(h, w, (bh, bw))
    hi = resize_u8(img, nh, nw, "bicubic")
    canvas = np.zeros((bh, bw, 3), dtype=np.uint8)
    top, left = (bh - nh) // 2, (bw - nw) // 2
    canvas[top:top + nh, left:left + nw] = hi
    crops = [resize_u8(img, 336, 336, "bicubic")]
    for cy in range(bh // 336):
        for cx in range(bw // 336):
            crops.append(canvas[cy * 336:(cy + 1) * 336, cx * 336:(cx + 1) * 336])
    n = len(crops)
    out = np.zeros((max_crops or n, 3, 336, 336), dtype=np.float32)
    for i, c in enumerate(crops):
        out[i] = _norm(c)
    return out, (h, w)
