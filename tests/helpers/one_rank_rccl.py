#!/usr/bin/env python3
"""A world of ONE rank on RCCL (backend "nccl"): the only way a 1-GPU box can execute the device-collective branch of
scoring.gather_rewards_async -- all_gather_into_tensor / all_gather on device tensors, enqueued on the dedicated side stream behind
the compute stream, joined by GatherHandle.wait().  Started as a child process by tests/test_gpu_multiprocess.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "llava-reward_amd"), ROOT):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from llava_reward_amd import scoring, synth  # noqa: E402
from llava_reward_amd.model import RewardModel  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    cfg = synth.tiny_config(is_general_preference=True, value_head_dim=2)
    model = RewardModel(cfg, synth_seed=71, max_batch=8, max_seq=512, max_crops=3).to("cuda:0").eval()       # (the form probe all-reduces: world of one)
    b = {k: torch.from_numpy(v).cuda() for k, v in synth.synth_batch(cfg, 71, [5, 3, 4], (1, 1)).items()}
    r = model.custom_forward(**b)[0]
    # equal shards: all_gather_into_tensor; ragged shards (n_total given): padded all_gather + cat -- both on the side stream
    h1 = scoring.gather_rewards_async(r, None, always_collective=True)
    h2 = scoring.gather_rewards_async(r, 3, always_collective=True)
    assert h1._stream is not None and h1._stream is scoring._side_streams[0] and h2._stream is h1._stream
    # work enqueued on the compute stream AFTER the gather must not be ordered behind it: the handle's stream is another one
    assert h1._stream.cuda_stream != torch.cuda.current_stream().cuda_stream
    r2 = model.custom_forward(**b)[0]
    g1, g2 = h1.wait(), h2.wait()
    torch.cuda.synchronize()
    assert torch.equal(g1, r) and torch.equal(g2, r) and torch.equal(r2, r)
    assert h1._stream is None                     # joined
    dist.barrier()
    dist.destroy_process_group()
    print("ONE_RANK_RCCL_OK")


if __name__ == "__main__":
    main()
