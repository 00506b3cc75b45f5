"""GPU input hand-over for the Phi-3.5-V path: what the reference's processor does on one CPU
thread per image (processing_phi3_v.py:85-107 HD_transform, :262-288 normalise / global view / crop tiling, :407-454 text +
image-slot merge) with the pixel work on the GPU: the decoded uint8 image crosses PCIe (0.5 MB instead of 23 MB of fp32) and
`lr_hd_transform` (csrc/preprocess.hip) writes `pixel_values` in place.  No CPU fallback: without the HIP library this fails."""
from __future__ import annotations

import ctypes as C
import re
from typing import List, Sequence

import numpy as np
import torch

from . import _lib as L

CROP = 336


def hd_transform_batch(images: Sequence, num_crops: int = 16, device="cuda", out: torch.Tensor = None):
    """images: RGB uint8 [h, w, 3] arrays / tensors (host or device), one per row.
    Returns (pixel_values [B, num_crops+1, 3, 336, 336] fp32 on `device`, image_sizes [B, 2] int64 host, num_img_tokens list)."""
    lib = L.load()
    dev = torch.device(device)
    B = len(images)
    if out is None:
        out = torch.empty(B, num_crops + 1, 3, CROP, CROP, dtype=torch.float32, device=dev)
    if tuple(out.shape) != (B, num_crops + 1, 3, CROP, CROP) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out must be a contiguous fp32 [B, num_crops+1, 3, 336, 336] tensor")
    sizes = np.zeros((B, 2), dtype=np.int64)
    ntok = []
    st = torch.cuda.current_stream(dev)
    ws, ws_bytes = None, 0                                    # scratch, reused in stream order, grown as needed
    for b, im in enumerate(images):
        t = im if torch.is_tensor(im) else torch.from_numpy(np.array(im))      # (copy: PIL-backed arrays are read-only)
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise ValueError(f"image {b}: expected RGB uint8 [h, w, 3], got {tuple(t.shape)} {t.dtype}")
        t = t.to(dev, non_blocking=True).contiguous()
        h, w = int(t.shape[0]), int(t.shape[1])
        need = lib.lr_hd_transform_workspace(h, w, num_crops)
        if need == 0:
            raise ValueError(f"image {b} ({h}x{w}): " + lib.lr_last_error(None).decode())
        if need > ws_bytes:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            ws_bytes = need
        size = (C.c_int64 * 2)()
        n = C.c_int32()
        rc = lib.lr_hd_transform(C.c_void_p(t.data_ptr()), h, w, num_crops, C.c_void_p(out[b].data_ptr()), size, C.byref(n),
                                 C.c_void_p(ws.data_ptr()), ws_bytes, C.c_void_p(st.cuda_stream))
        if rc != 0:
            raise RuntimeError(lib.lr_last_error(None).decode())
        sizes[b] = (size[0], size[1])
        ntok.append(int(n.value))
    return out, torch.from_numpy(sizes), ntok


_IMAGE_TAG = re.compile(r"<\|image_\d+\|>")


def merge_text_and_image_slots(tokenizer, text: str, num_img_tokens: List[int]):
    """The token side of Phi3VProcessor._convert_images_texts_to_inputs (processing_phi3_v.py:407-454): the prompt is split
    at its <|image_k|> tags, each chunk is tokenised on its own, and tag k becomes num_img_tokens[k-1] copies of -k."""
    chunks = [tokenizer(c).input_ids for c in _IMAGE_TAG.split(text)]
    tags = [int(s.split("|")[1].split("_")[-1]) for s in _IMAGE_TAG.findall(text)]
    uniq = sorted(set(tags))
    if uniq != list(range(1, len(uniq) + 1)):
        raise AssertionError(f"image_ids must start from 1, and must be continuous int, e.g. [1, 2, 3], cannot be {uniq}")
    if len(uniq) != len(num_img_tokens):
        raise AssertionError(f"total images must be the same as the number of image tags, got {len(uniq)} image tags and "
                             f"{len(num_img_tokens)} images")
    ids: List[int] = []
    for i, chunk in enumerate(chunks):
        ids.extend(chunk)
        if i < len(tags):
            ids.extend([-tags[i]] * num_img_tokens[tags[i] - 1])
    input_ids = torch.tensor(ids, dtype=torch.long).unsqueeze(0)
    return input_ids, (input_ids > -1000000).to(torch.long)


def inference_process_phi3v_device(args, tokenizer, img_dir_list, caption, device="cuda", num_crops: int = 16):
    """inference_process_phi3v (eval/reward_adaptor_loader.py:158-173) without the CPU image processor: same prompt, same
    return value (one dict per image with input_ids, attention_mask, pixel_values, image_sizes on `device`)."""
    from PIL import Image
    prompt_messages = {"role": "user", "content": f"<|image_1|>\n{caption}"}
    prompt = tokenizer.apply_chat_template([prompt_messages], tokenize=False, add_generation_prompt=True)[:-22] + tokenizer.eos_token
    imgs = [np.asarray(Image.open(d).convert("RGB")) for d in img_dir_list]
    pix, sizes, ntok = hd_transform_batch(imgs, num_crops, device)
    out = []
    for b in range(len(imgs)):
        ids, mask = merge_text_and_image_slots(tokenizer, prompt, [ntok[b]])
        out.append({"input_ids": ids.to(device), "attention_mask": mask.to(device), "pixel_values": pix[b:b + 1],
                    "image_sizes": sizes[b:b + 1].to(device)})
    return out
