"""CPU tests of the image hand-over oracle (oracle/phi3v_hd_transform_oracle.py): the restated resampler against Pillow
itself (bit-exact), the restated bicubic against torch (1e-6), the whole hand-over against the committed golden digests
(tests/golden/pre_*.json, made by make_preprocess_goldens.py from the real primitives), and the token-count KATs of
SURVEY.md §8c."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from llava_reward_amd import synth
from oracle import phi3v_hd_transform_oracle as O

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "pre_*.json")))


def sample_index(n=96):
    return [(i % 3, (37 * i + 5) % 336, (101 * i + 11) % 336) for i in range(n)]


@pytest.mark.parametrize("case", [(336, 336, 1344, 1344), (512, 640, 1075, 1344), (700, 500, 300, 211), (97, 133, 97, 400),
                                  (1000, 1500, 336, 504), (50, 60, 672, 806), (33, 1, 100, 7), (640, 480, 640, 480)])
@pytest.mark.parametrize("smooth", [False, True])
def test_resampler_restatement_is_bit_exact_with_pillow(case, smooth):
    from PIL import Image
    h, w, nh, nw = case
    a = synth.synth_image(7, f"rs.{h}.{w}", h, w, smooth)
    ref = np.asarray(Image.fromarray(a).resize((nw, nh), Image.BILINEAR))
    got = O.resize_bilinear_u8(a, nh, nw)
    assert got.shape == ref.shape and np.array_equal(ref, got)


@pytest.mark.parametrize("hw", [(1344, 1344), (1008, 1344), (336, 672), (672, 336), (336, 336)])
def test_bicubic_restatement_matches_torch(hw):
    H, W = hw
    x = synth.synth_pixels(3, f"bc.{H}.{W}", (3, H, W))
    ref = torch.nn.functional.interpolate(torch.from_numpy(x)[None], size=(336, 336), mode="bicubic")[0].numpy()
    got = O.bicubic_resize_f32(x, 336, 336)
    assert np.abs(ref - got).max() < 2e-6 * max(1.0, np.abs(ref).max())


def test_token_count_kats():
    # 336^2, 512x640 and 768^2 images all become 1344x1344 at num_crops=16 -> 2509 slots; 757 at num_crops=4
    for (w, h) in [(336, 336), (512, 640), (640, 512), (768, 768)]:
        trans, new_w, new_h, top, tar = O.hd_geometry(w, h, 16)
        H, W = (new_w, tar) if trans else (tar, new_w)
        assert (H, W) == (1344, 1344) and O.num_img_tokens(H, W) == 2509
        assert synth.hd_target_size(w, h, 16)[:2] == (W, H)
    trans, new_w, new_h, top, tar = O.hd_geometry(336, 336, 4)
    assert (tar, new_w) == (672, 672) and O.num_img_tokens(672, 672) == 757


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[4:-5] for p in GOLDEN])
def test_oracle_matches_golden_digests(path):
    g = json.load(open(path))
    a = synth.synth_image(g["seed"], "pre." + g["name"], g["h"], g["w"], g["smooth"])
    pv, (H, W), ntok = O.preprocess(a, g["num_crops"])
    assert [H, W] == g["image_size"] and ntok == g["num_img_tokens"]
    assert hashlib.sha256(np.ascontiguousarray(pv[1:]).tobytes()).hexdigest() == g["local_sha256"]      # bit-exact part
    got = np.array([pv[0, c, y, x] for c, y, x in sample_index()], dtype=np.float64)
    assert np.abs(got - np.array(g["global_samples"])).max() < 2e-6                                       # fp32 bicubic
    assert (pv[1 + g["n_local"]:] == 0).all()


# ---------------------------------------------------------------- Qwen2-VL / LLaVA-NeXT image processors (third party)
from oracle import llava_next_image_oracle as LO  # noqa: E402
from oracle import qwen2vl_image_oracle as QO  # noqa: E402

GOLDEN_Q = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "preq_*.json")))
GOLDEN_L = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "prel_*.json")))


@pytest.mark.parametrize("case", [(336, 336, 448, 448), (700, 500, 300, 211), (1000, 1500, 336, 504), (64, 48, 336, 336), (40, 3, 9, 200)])
def test_bicubic_resampler_restatement_is_bit_exact_with_pillow(case):
    from PIL import Image
    h, w, nh, nw = case
    a = synth.synth_image(7, f"bc.{h}.{w}", h, w, True)
    ref = np.asarray(Image.fromarray(a).resize((nw, nh), Image.BICUBIC))
    assert np.array_equal(ref, O.resize_u8(a, nh, nw, "bicubic"))


@pytest.mark.parametrize("path", GOLDEN_Q, ids=[os.path.basename(p)[5:-5] for p in GOLDEN_Q])
def test_qwen_image_oracle_matches_real_processor_digest(path):
    g = json.load(open(path))
    a = synth.synth_image(g["seed"], "preq." + g["name"], g["h"], g["w"], g["smooth"])
    pv, grid = QO.preprocess(a, g["min_pixels"], g["max_pixels"])
    assert list(grid) == g["image_grid_thw"]
    assert hashlib.sha256(pv.tobytes()).hexdigest() == g["sha256"]          # bit-exact with the transformers processor


@pytest.mark.parametrize("path", GOLDEN_L, ids=[os.path.basename(p)[5:-5] for p in GOLDEN_L])
def test_llava_image_oracle_matches_real_processor_digest(path):
    g = json.load(open(path))
    a = synth.synth_image(g["seed"], "prel." + g["name"], g["h"], g["w"], g["smooth"])
    pv, size = LO.preprocess(a, g["pinpoints"])
    assert pv.shape[0] == g["n_crops"] and list(size) == g["image_size"]
    assert hashlib.sha256(pv.tobytes()).hexdigest() == g["sha256"]


def test_smart_resize_kats():
    # the reference's processor bounds (utils/utils.py:34-44): a 336 px image is lifted to 448x448 = 32x32 patches = 256 slots
    assert QO.smart_resize(336, 336, 28, 256 * 28 * 28, 1280 * 28 * 28) == (448, 448)
    assert QO.smart_resize(1200, 1600, 28, 256 * 28 * 28, 1280 * 28 * 28) == (840, 1148)
    assert QO.smart_resize(42, 70, 28, 56 * 56, 1280 * 28 * 28) == (56, 56)          # Python round() is half-to-even: 1.5 -> 2, 2.5 -> 2
    assert QO.smart_resize(3, 5, 28, 256 * 28 * 28, 1280 * 28 * 28) == (364, 588)
    with pytest.raises(ValueError):
        QO.smart_resize(10, 2100)


def test_oracles_against_live_processors_when_importable():
    """Where transformers' PIL-backend processors import (this image: yes), re-run them instead of trusting the digests."""
    try:
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
        from make_preprocess_goldens import third_party_processors
        q, l = third_party_processors()
    except Exception as e:                                   # pragma: no cover
        pytest.skip(f"processors not importable: {e}")
    from PIL import Image
    for (h, w) in [(123, 456), (640, 480)]:
        a = synth.synth_image(21, f"live.{h}.{w}", h, w, True)
        out = q(images=[Image.fromarray(a)], return_tensors="np")
        pv, grid = QO.preprocess(a)
        assert np.array_equal(out["pixel_values"], pv) and out["image_grid_thw"][0].tolist() == list(grid)
        out = l(images=[Image.fromarray(a)], return_tensors="np")
        assert np.array_equal(out["pixel_values"][0], LO.preprocess(a)[0])
