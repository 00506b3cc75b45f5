#!/usr/bin/env python3
"""One rank of a 2-process scoring job that shares ONE GPU (test helper; started as a child process by
tests/test_gpu_multiprocess.py before it touches the GPU itself).  The real engine scores this rank's row shards; the collective runs
over gloo (RCCL refuses two ranks on one device), staged through the host by scoring.gather_rewards."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "llava-reward_amd"), ROOT):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from llava_reward_amd import synth  # noqa: E402
from llava_reward_amd.model import RewardModel  # noqa: E402
from llava_reward_amd.scoring import score_candidates, score_pairwise, score_pairwise_files  # noqa: E402


def batches(cfg, seed):
    out = []
    for k, n in enumerate((5, 3, 1)):            # ragged shards over 2 ranks; the last batch has fewer rows than ranks
        lens_c = [3 + (k + i) % 4 for i in range(n)]
        lens_r = [2 + (2 * k + i) % 5 for i in range(n)]
        bc = {a: torch.from_numpy(b) for a, b in synth.synth_batch(cfg, seed + k, lens_c, (1, 1)).items()}
        br = {a: torch.from_numpy(b) for a, b in synth.synth_batch(cfg, seed + 100 + k, lens_r, (1, 1)).items()}
        out.append((bc, br, None, None))
    return out


def main():
    out_path = sys.argv[1]
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(0)
    if ws > 1:
        dist.init_process_group("gloo", rank=rank, world_size=ws)
    cfg = synth.tiny_config(is_general_preference=True, value_head_dim=2)
    seed = 71
    model = RewardModel(cfg, synth_seed=seed, max_batch=8, max_seq=512, max_crops=3).to("cuda:0").eval()
    args = types.SimpleNamespace(is_general_preference=True, value_head_dim=2, general_preference_tau=0.1)
    res = score_pairwise(model, args, batches(cfg, seed))
    # a population of 5 candidate images for one prompt (reward-guided sampling): in memory, on the GPU, sharded 3 + 2 over two ranks
    cands = [torch.from_numpy(synth.synth_image(seed, f"cand.{i}", 336, 336)).cuda() for i in range(5)]
    cr = score_candidates(model, synth.StandInTokenizer(), "a photo of a cat", cands, num_crops=1, batch_size=2, pad_token_id=cfg.vocab_size - 1)
    # the files-to-rewards loop: 5 (caption, chosen, rejected) triples, sharded 3 + 2 BEFORE anything is decoded, prefetched batches of 2;
    # same-sized images so that every row sees the same V_max whatever its shard (the reference's SkipCA attends over padded rows)
    pairs = [(f"caption {i}", synth.synth_image(seed, f"c.{i}", 200, 300), synth.synth_image(seed, f"r.{i}", 200, 300)) for i in range(5)]
    fr = score_pairwise_files(model, args, synth.StandInTokenizer(), pairs, batch_size=2, num_crops=1, pad_token_id=cfg.vocab_size - 1)
    # The operand form: .to('cuda') locked it on the probe rows (the same on every rank; the distance is all-reduced).  calibrate() on
    # batches that DIFFER per rank must still end with one decision: every rank reports the largest distance any rank saw.
    probe = dict(model.form_info)
    kb = [{a: torch.from_numpy(b).cuda() for a, b in synth.synth_batch(cfg, seed + 500 + k, [4 + k, 2], (1, 1)).items()} for k in range(2)]
    if ws > 1:
        cal = model.calibrate(kb[rank], parity_budget=1.0)["default_vs_strict"]
    else:
        cal = [model.calibrate(b, parity_budget=1.0)["default_vs_strict"] for b in kb]
    json.dump({"rank": rank, "probs": res["probs"], "proportion": res["proportion"], "candidates": cr.cpu().tolist(), "file_probs": fr["probs"],
               "probe_form": probe["form"], "probe_distance": probe["default_vs_strict"], "calibrate_distance": cal}, open(out_path, "w"))
    if ws > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
