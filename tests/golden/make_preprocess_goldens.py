#!/usr/bin/env python3
"""Golden vectors of the Phi-3.5-V image hand-over, made in the build container.

The reference's processor (llava_reward/models/base_mllm/phi3_v/processing_phi3_v.py) cannot be imported here: it needs
torchvision, which this image does not have.  Its two numerical primitives CAN be run: Pillow's Image.resize (what
torchvision.transforms.functional.resize calls on a PIL image) and torch.nn.functional.interpolate(bicubic).  This script
runs those real primitives, composed as processing_phi3_v.py:85-107 / :262-288 composes them (torchvision's constant
`pad` on a PIL image = ImageOps.expand), on seeded images and stores digests: the padded size, the token count, a SHA-256
of the local crops' bytes (bit-exact part) and 96 sampled values of the bicubic global view.

The Qwen2-VL and LLaVA-NeXT processors are third party (transformers); their PIL-backend classes import here, so their
goldens (preq_*.json, prel_*.json) are digests of the REAL processors' outputs on seeded images.

    python tests/golden/make_preprocess_goldens.py        # writes tests/golden/pre_*.json, preq_*.json, prel_*.json
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch
from PIL import Image, ImageOps

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
from llava_reward_amd import synth  # noqa: E402

MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)

CASES = [  # name, h, w, num_crops, smooth
    ("square336_nc16", 336, 336, 16, False),
    ("portrait640x512_nc16", 640, 512, 16, True),
    ("landscape512x640_nc4", 512, 640, 4, False),
    ("wide300x900_nc16", 300, 900, 16, True),
    ("down1500x2000_nc16", 1500, 2000, 16, False),
    ("tiny97x133_nc4", 97, 133, 4, True),
]


def pipeline(a, hd_num):
    img = Image.fromarray(a)
    width, height = img.size
    trans = False
    if width < height:
        img = img.transpose(Image.TRANSPOSE)
        trans = True
        width, height = img.size
    ratio = width / height
    scale = 1
    while scale * np.ceil(scale / ratio) <= hd_num:
        scale += 1
    scale -= 1
    new_w = int(scale * 336)
    new_h = int(new_w / ratio)
    img = img.resize((new_w, new_h), Image.BILINEAR)
    tar = int(np.ceil(new_h / 336) * 336)
    top = int((tar - new_h) / 2)
    img = ImageOps.expand(img, border=(0, top, 0, tar - new_h - top), fill=(255, 255, 255))
    if trans:
        img = img.transpose(Image.TRANSPOSE)
    t = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float().div(255)
    t = (t - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)
    g = torch.nn.functional.interpolate(t[None].float(), size=(336, 336), mode="bicubic")
    h, w = t.shape[1:]
    loc = t.reshape(1, 3, h // 336, 336, w // 336, 336).permute(0, 2, 4, 1, 3, 5).reshape(-1, 3, 336, 336)
    out = torch.cat([g, loc], 0)
    if out.shape[0] < hd_num + 1:
        out = torch.cat([out, torch.zeros(hd_num + 1 - out.shape[0], 3, 336, 336)], 0)
    return out.numpy(), (h, w)


def sample_index(n=96):
    return [(c, (37 * i + 5) % 336, (101 * i + 11) % 336) for i in range(n) for c in (i % 3,)]


QWEN_CASES = [("sq336", 336, 336, False), ("p500x375", 500, 375, True), ("big1200x1600", 1200, 1600, False),
              ("tiny97x133", 97, 133, True), ("strip30x2000", 30, 2000, True)]
LLAVA_CASES = [("sq336", 336, 336, False), ("p500x375", 500, 375, True), ("big1200x1600", 1200, 1600, False),
               ("tiny97x133", 97, 133, True), ("wide300x1100", 300, 1100, True), ("tall900x200", 900, 200, False)]
LLAVA_PINPOINTS = [[336, 672], [672, 336], [672, 672], [1008, 336], [336, 1008]]


def sample_rows(n_rows, n=64):
    return [((977 * i + 3) % n_rows, (131 * i + 7) % 1176) for i in range(n)]


def third_party_processors():
    """The real transformers image processors (PIL backend = the arithmetic of the pinned 4.50 'slow' processors)."""
    from transformers.models.llava_next.image_processing_pil_llava_next import LlavaNextImageProcessorPil
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil
    q = Qwen2VLImageProcessorPil(min_pixels=256 * 28 * 28, max_pixels=1280 * 28 * 28)              # utils/utils.py:34-44
    l = LlavaNextImageProcessorPil(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336},
                                   image_grid_pinpoints=LLAVA_PINPOINTS, resample=3, image_mean=list(MEAN), image_std=list(STD))
    return q, l


def main():
    import PIL
    import transformers
    q, l = third_party_processors()
    made = {"pillow": PIL.__version__, "transformers": transformers.__version__}
    for name, h, w, smooth in QWEN_CASES:
        a = synth.synth_image(1234, "preq." + name, h, w, smooth)
        out = q(images=[Image.fromarray(a)], return_tensors="np")
        pv = np.ascontiguousarray(out["pixel_values"], dtype=np.float32)
        g = {"name": name, "seed": 1234, "h": h, "w": w, "smooth": smooth, "min_pixels": 256 * 28 * 28, "max_pixels": 1280 * 28 * 28,
             "image_grid_thw": out["image_grid_thw"][0].tolist(), "sha256": hashlib.sha256(pv.tobytes()).hexdigest(),
             "samples": [float(pv[r, c]) for r, c in sample_rows(pv.shape[0])], "made_with": made}
        json.dump(g, open(os.path.join(HERE, f"preq_{name}.json"), "w"), indent=1)
        print("qwen", name, g["image_grid_thw"], g["sha256"][:16])
    for name, h, w, smooth in LLAVA_CASES:
        a = synth.synth_image(1234, "prel." + name, h, w, smooth)
        out = l(images=[Image.fromarray(a)], return_tensors="np")
        pv = np.ascontiguousarray(out["pixel_values"][0], dtype=np.float32)
        g = {"name": name, "seed": 1234, "h": h, "w": w, "smooth": smooth, "pinpoints": LLAVA_PINPOINTS,
             "n_crops": int(pv.shape[0]), "image_size": [int(v) for v in out["image_sizes"][0]],
             "sha256": hashlib.sha256(pv.tobytes()).hexdigest(), "made_with": made}
        json.dump(g, open(os.path.join(HERE, f"prel_{name}.json"), "w"), indent=1)
        print("llava", name, g["n_crops"], g["sha256"][:16])
    for name, h, w, nc, smooth in CASES:
        a = synth.synth_image(1234, "pre." + name, h, w, smooth)
        pv, (H, W) = pipeline(a, nc)
        n_local = (H // 336) * (W // 336)
        g = {"name": name, "seed": 1234, "h": h, "w": w, "num_crops": nc, "smooth": smooth,
             "image_size": [H, W], "num_img_tokens": int((n_local + 1) * 144 + 1 + (H // 336 + 1) * 12),
             "n_local": n_local, "local_sha256": hashlib.sha256(np.ascontiguousarray(pv[1:]).tobytes()).hexdigest(),
             "global_samples": [float(pv[0, c, y, x]) for c, y, x in sample_index()],
             "made_with": {"pillow": PIL.__version__, "torch": torch.__version__}}
        with open(os.path.join(HERE, f"pre_{name}.json"), "w") as f:
            json.dump(g, f, indent=1)
        print(name, g["image_size"], g["num_img_tokens"], g["local_sha256"][:16])


if __name__ == "__main__":
    main()
