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
