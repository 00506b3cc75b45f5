"""Files on disk -> rewards with the input side overlapped (llava_reward_amd.scoring.PrefetchingBatcher / score_pairwise_files):
the prefetched path gives bit-identical rewards to the synchronous one (same kernels, another stream), in the reference loop's
statistics (eval/batch_inference_rm_phi.py:79-121)."""
import types

import numpy as np
import pytest
import torch

from llava_reward_amd import synth
from llava_reward_amd.model import RewardModel
from llava_reward_amd.preprocess import batch_inference_process_phi3v_device
from llava_reward_amd.reward_adaptor_loader import preference_compute
from llava_reward_amd.scoring import PrefetchingBatcher, score_pairwise_files

pytestmark = pytest.mark.gpu


def _files(tmp_path, n, seed):
    from PIL import Image
    paths = []
    for i in range(n):
        h, w = [(336, 336), (200, 300), (500, 375), (97, 133)][i % 4]
        p = str(tmp_path / f"img{seed}_{i}.png")
        Image.fromarray(synth.synth_image(seed, f"file{i}", h, w, i % 2 == 0)).save(p)
        paths.append(p)
    return paths


def test_prefetched_batches_equal_synchronous_ones(tmp_path):
    cfg = synth.tiny_config()
    tok = synth.StandInTokenizer()
    model = RewardModel(cfg, synth_seed=5, max_batch=4, max_seq=1024, max_crops=5).to("cuda").eval()
    paths = _files(tmp_path, 10, 1)
    items = [(p, "a caption " * (1 + i % 3)) for i, p in enumerate(paths)]
    want = []
    for lo in range(0, len(items), 4):
        b = batch_inference_process_phi3v_device(None, tok, items[lo: lo + 4], device="cuda", num_crops=4)
        want.append(model.custom_forward(**b)[0].clone())
    got = [model.custom_forward(**b)[0].clone() for b in PrefetchingBatcher(items, tok, batch_size=4, num_crops=4, device="cuda", depth=2, workers=3)]
    torch.cuda.synchronize()
    assert len(got) == len(want) == 3 and all(torch.equal(a, b) for a, b in zip(got, want))
    # a failing decode surfaces on the consumer's side
    with pytest.raises(Exception):
        list(PrefetchingBatcher([(str(tmp_path / "missing.png"), "x")], tok, batch_size=1, num_crops=4))


def test_score_pairwise_files_matches_direct_forwards(tmp_path):
    cfg = synth.tiny_config(is_general_preference=True, value_head_dim=2)
    tok = synth.StandInTokenizer()
    args = types.SimpleNamespace(is_general_preference=True, value_head_dim=2, general_preference_tau=0.1)
    model = RewardModel(cfg, synth_seed=6, max_batch=4, max_seq=1024, max_crops=5).to("cuda").eval()
    ch, rj = _files(tmp_path, 7, 2), _files(tmp_path, 7, 3)
    pairs = [(f"caption number {i}", c, r) for i, (c, r) in enumerate(zip(ch, rj))]
    out = score_pairwise_files(model, args, tok, pairs, batch_size=3, num_crops=4, depth=2, workers=2)
    probs = []
    for lo in range(0, 7, 3):
        part = pairs[lo: lo + 3]
        c = model.custom_forward(**batch_inference_process_phi3v_device(None, tok, [(c, cap) for cap, c, _ in part], device="cuda", num_crops=4))[0]
        r = model.custom_forward(**batch_inference_process_phi3v_device(None, tok, [(r, cap) for cap, _, r in part], device="cuda", num_crops=4))[0]
        probs.extend(preference_compute(args, c, r).tolist())
    assert out["probs"] == probs and len(probs) == 7
    assert out["proportion"] == sum(p > 0.5 for p in probs) / 7 and abs(out["prob_mean"] - float(np.mean(probs))) < 1e-7


def test_prefetching_batcher_closes_when_the_consumer_stops_early(tmp_path):
    """A consumer that leaves the loop early (an exception in custom_forward, a `break`) must not leave the producer thread blocked
    on its bounded queue with `depth` device batches alive: close() / the context manager end it."""
    cfg = synth.tiny_config()
    tok = synth.StandInTokenizer()
    paths = _files(tmp_path, 12, 4)
    items = [(p, "caption") for p in paths]
    pb = PrefetchingBatcher(items, tok, batch_size=2, num_crops=4, device="cuda", depth=1, workers=2)
    it = iter(pb)
    first = next(it)
    assert first["input_ids"].shape[0] == 2 and pb._thread.is_alive()          # 5 more batches to go, the queue holds one
    pb.close()
    assert not pb._thread.is_alive() and pb._q.empty()
    pb.close()                                                                  # idempotent
    with pytest.raises(ZeroDivisionError):
        with PrefetchingBatcher(items, tok, batch_size=2, num_crops=4, device="cuda", depth=1, workers=2) as pb2:
            for _ in pb2:
                1 / 0
    assert not pb2._thread.is_alive()
