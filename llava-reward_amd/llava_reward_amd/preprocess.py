"""GPU input hand-over (Phi-3.5-V HD transform; Qwen2-VL and LLaVA-NeXT image processors) for the scoring path: what the reference's processor does on one CPU
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
        t = _as_device_u8(im, b, dev)
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


def _as_device_u8(im, b, dev):
    t = im if torch.is_tensor(im) else torch.from_numpy(np.array(im))      # (copy: PIL-backed arrays are read-only)
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise ValueError(f"image {b}: expected RGB uint8 [h, w, 3], got {tuple(t.shape)} {t.dtype}")
    return t.to(dev, non_blocking=True).contiguous()


def qwen_image_batch(images: Sequence, min_pixels: int = 256 * 28 * 28, max_pixels: int = 1280 * 28 * 28, device="cuda"):
    """The Qwen2-VL image processor on the GPU (transformers image_processing_qwen2_vl; the reference builds it with these pixel
    bounds, utils/utils.py:34-44).  images: RGB uint8 [h, w, 3].  Returns (pixel_values [sum gh*gw, 1176] fp32 on `device`,
    image_grid_thw [n, 3] int64 host) -- the two image entries of the processor's BatchFeature, bit-exact."""
    lib = L.load()
    dev = torch.device(device)
    grids = np.zeros((len(images), 3), dtype=np.int64)
    shapes = []
    for b, im in enumerate(images):
        h, w = int(im.shape[0]), int(im.shape[1])
        g = (C.c_int64 * 3)()
        if lib.lr_qwen_image_grid(h, w, min_pixels, max_pixels, g) != 0:
            raise ValueError(f"image {b} ({h}x{w}): " + lib.lr_last_error(None).decode())
        grids[b] = (g[0], g[1], g[2])
        shapes.append((h, w))
    rows = (grids[:, 1] * grids[:, 2]).astype(np.int64)
    out = torch.empty(int(rows.sum()), 1176, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev)
    ws, ws_bytes, r0 = None, 0, 0
    for b, im in enumerate(images):
        t = _as_device_u8(im, b, dev)
        h, w = shapes[b]
        need = lib.lr_qwen_image_workspace(h, w, min_pixels, max_pixels)
        if need > ws_bytes:
            ws, ws_bytes = torch.empty(need, dtype=torch.uint8, device=dev), need
        rc = lib.lr_qwen_image_transform(C.c_void_p(t.data_ptr()), h, w, min_pixels, max_pixels, C.c_void_p(out[r0:].data_ptr()), None,
                                         C.c_void_p(ws.data_ptr()), ws_bytes, C.c_void_p(st.cuda_stream))
        if rc != 0:
            raise RuntimeError(lib.lr_last_error(None).decode())
        r0 += int(rows[b])
    return out, torch.from_numpy(grids)


LLAVA_PINPOINTS = ((336, 672), (672, 336), (672, 672), (1008, 336), (336, 1008))     # llava-v1.6-mistral-7b-hf


def llava_image_batch(images: Sequence, pinpoints=LLAVA_PINPOINTS, max_crops: int = None, device="cuda"):
    """The LLaVA-NeXT image processor on the GPU (transformers image_processing_llava_next).  Returns (pixel_values
    [B, C, 3, 336, 336] fp32 on `device`, C = max_crops or the batch maximum as the processor pads, image_sizes [B, 2] int64 host
    = ORIGINAL (h, w)), bit-exact with the processor."""
    lib = L.load()
    dev = torch.device(device)
    pin = (C.c_int32 * (2 * len(pinpoints)))(*[int(v) for p in pinpoints for v in p])
    ncrops, shapes = [], []
    for b, im in enumerate(images):
        h, w = int(im.shape[0]), int(im.shape[1])
        g = (C.c_int32 * 5)()
        if lib.lr_llava_image_geometry(h, w, pin, len(pinpoints), g) != 0:
            raise ValueError(f"image {b} ({h}x{w}): " + lib.lr_last_error(None).decode())
        ncrops.append(int(g[4]))
        shapes.append((h, w))
    Cn = max_crops if max_crops is not None else max(ncrops)
    if Cn < max(ncrops):
        raise ValueError(f"max_crops={Cn} but an image needs {max(ncrops)} crops")
    out = torch.empty(len(images), Cn, 3, CROP, CROP, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev)
    ws, ws_bytes = None, 0
    for b, im in enumerate(images):
        t = _as_device_u8(im, b, dev)
        h, w = shapes[b]
        need = lib.lr_llava_image_workspace(h, w, pin, len(pinpoints))
        if need > ws_bytes:
            ws, ws_bytes = torch.empty(need, dtype=torch.uint8, device=dev), need
        rc = lib.lr_llava_image_transform(C.c_void_p(t.data_ptr()), h, w, pin, len(pinpoints), Cn, C.c_void_p(out[b].data_ptr()), None,
                                          C.c_void_p(ws.data_ptr()), ws_bytes, C.c_void_p(st.cuda_stream))
        if rc != 0:
            raise RuntimeError(lib.lr_last_error(None).decode())
    return out, torch.tensor(shapes, dtype=torch.int64)


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


def zero_pad_sequences(sequences, side: str = "left", value=0):
    """llava_reward/datasets/utils.py:5-13: rows of different length (last dim) -> one stacked tensor, padded with `value` on `side`.
    Works on host or device tensors; one output allocation, one copy per row."""
    assert side in ("left", "right")
    max_len = max(int(seq.size(-1)) for seq in sequences)
    first = sequences[0]
    out = torch.full((len(sequences),) + tuple(first.shape[:-1]) + (max_len,), value, dtype=first.dtype, device=first.device)
    for i, seq in enumerate(sequences):
        n = int(seq.size(-1))
        if side == "left":
            out[i, ..., max_len - n:] = seq
        else:
            out[i, ..., :n] = seq
    return out


def collate_rows(rows, pad_token_id: int, squeeze: bool = True):
    """The batch builder of the reference's dataset (reward_dataset.py:164-179 / :196-202) over per-row dicts as
    inference_process_phi3v[_device] returns them (input_ids / attention_mask [1, S_i], pixel_values [1, C, 3, 336, 336],
    image_sizes [1, 2]): ids LEFT-padded with the tokenizer's pad id, masks with 0, pixel tensors stacked -- on whatever device the
    rows live, so rows prepared on the GPU never go back to the host.  The reference's collate keeps the singleton dim and its
    callers squeeze it (eval/batch_inference_rm_phi.py:82-90); squeeze=True returns [B, S] / [B, C, 3, 336, 336] / [B, 2] directly."""
    ids = zero_pad_sequences([r["input_ids"] for r in rows], value=pad_token_id)
    mask = zero_pad_sequences([r["attention_mask"] for r in rows])
    pix = torch.stack([r["pixel_values"] for r in rows], dim=0)
    sizes = torch.stack([torch.as_tensor(r["image_sizes"]) for r in rows], dim=0)
    out = {"input_ids": ids, "attention_mask": mask, "pixel_values": pix, "image_sizes": sizes}
    if squeeze:
        out = {k: (v.squeeze(1) if v.dim() > 1 and v.shape[1] == 1 else v) for k, v in out.items()}
    return out


def _load_rgb(image):
    """An image of a scoring request: a path (opened as the reference does, reward_adaptor_loader.py:161), a PIL image, or RGB uint8
    [h, w, 3] pixels already in memory -- a numpy array or a tensor on any device (a sampler's decoded candidates stay on the GPU)."""
    if isinstance(image, (str, bytes)) or hasattr(image, "__fspath__"):
        from PIL import Image
        return np.asarray(Image.open(image).convert("RGB"))
    if hasattr(image, "convert") and not torch.is_tensor(image):          # PIL.Image
        return np.asarray(image.convert("RGB"))
    return image


def batch_inference_process_phi3v_device(args, tokenizer, items, device="cuda", num_crops: int = 16, pad_token_id: int = None):
    """Rows for ONE batched custom_forward from (image, caption) pairs with captions of any length -- image: a path, a PIL image or
    RGB uint8 pixels in memory (_load_rgb): every image goes through lr_hd_transform into one [B, num_crops+1, 3, 336, 336]
    tensor, every prompt is built and merged with its image slots as inference_process_phi3v does
    (eval/reward_adaptor_loader.py:163-167), and the rows are left-padded into [B, S] (collate_rows).  Returns the dict
    custom_forward(**batch) takes."""
    imgs = [_load_rgb(image) for image, _ in items]
    pix, sizes, ntok = hd_transform_batch(imgs, num_crops, device)
    rows = []
    for b, (_, caption) in enumerate(items):
        msg = {"role": "user", "content": f"<|image_1|>\n{caption}"}
        prompt = tokenizer.apply_chat_template([msg], tokenize=False, add_generation_prompt=True)[:-22] + tokenizer.eos_token
        ids, mask = merge_text_and_image_slots(tokenizer, prompt, [ntok[b]])
        rows.append({"input_ids": ids.to(device), "attention_mask": mask.to(device), "pixel_values": pix[b:b + 1], "image_sizes": sizes[b:b + 1]})
    pad = pad_token_id if pad_token_id is not None else tokenizer.pad_token_id
    return collate_rows(rows, pad)
