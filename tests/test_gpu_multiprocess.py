"""The multi-process path as far as ONE GPU allows (the driver owns the 8-GPU runs): two fresh rank processes share cuda:0, the real
engine scores each rank's contiguous row shard, rewards are all-gathered (gloo: RCCL refuses two ranks on one device, so the
RCCL transport itself stays unexercised here -- DESIGN.md §6 says so), and every rank must end with the single-process result bit
for bit.  Also the bench's own N>1 branch (barrier, max-over-ranks timing, gather inside the step) with 2 ranks on one device."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPER = os.path.join(ROOT, "tests", "helpers", "two_rank_score.py")


def _env(rank, ws, port):
    e = dict(os.environ)
    e.update(RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    return e


def test_two_rank_processes_on_one_gpu_equal_single_process(tmp_path):
    port = 29600 + os.getpid() % 300
    single = str(tmp_path / "single.json")
    subprocess.run([sys.executable, HELPER, single], env=_env(0, 1, port), check=True, timeout=600)
    outs = [str(tmp_path / f"rank{r}.json") for r in range(2)]
    ps = [subprocess.Popen([sys.executable, HELPER, outs[r]], env=_env(r, 2, port)) for r in range(2)]
    for p in ps:
        assert p.wait(timeout=600) == 0
    ref = json.load(open(single))
    assert len(ref["probs"]) == 9
    for r in range(2):
        got = json.load(open(outs[r]))
        assert got["rank"] == r and got["probs"] == ref["probs"] and got["proportion"] == ref["proportion"]      # bit-identical on every rank
        assert len(got["candidates"]) == 5 and got["candidates"] == ref["candidates"]                            # score_candidates: sharded 3 + 2
        assert len(got["file_probs"]) == 5 and got["file_probs"] == ref["file_probs"]                            # score_pairwise_files: pairs sharded 3 + 2
        # the operand form locked by .to('cuda') is the single process's (same probe rows, same distance), and a calibrate() call fed
        # DIFFERENT batches on the two ranks ends with one decision: the largest distance either rank measured
        assert got["probe_form"] == ref["probe_form"] and got["probe_distance"] == ref["probe_distance"]
        assert got["calibrate_distance"] == max(ref["calibrate_distance"])


def test_rccl_side_stream_gather_in_a_world_of_one_rank():
    """RCCL refuses two ranks on one device, but a world of ONE rank is a legal communicator: the device-collective branch of
    gather_rewards_async (side stream behind the compute stream, all_gather_into_tensor and the padded ragged form, GatherHandle.wait)
    runs on the real transport library."""
    helper = os.path.join(ROOT, "tests", "helpers", "one_rank_rccl.py")
    port = 29950 + os.getpid() % 40
    r = subprocess.run([sys.executable, helper], env=_env(0, 1, port), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ONE_RANK_RCCL_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_bench_two_ranks_on_one_device(tmp_path):
    """The BARE command the driver's N = 1 run has the shape of, `python bench.py --gpus 2 ...` with no launcher around it: bench.py
    starts its own rank processes (torch.distributed.run) before touching the GPU and relays rank 0's JSON line and the exit code."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "2", "--backend", "gloo",
           "--all-ranks-on-device", "0", "--quick"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    line = [l for l in lines if l.startswith("{")][-1]
    assert line == [l for l in lines if l.strip()][-1] and len(line) < 4096        # the driver parses the LAST line of an 8 KB tail
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 4 and res["scaling"] == "weak" and res["value"] > 0
    assert res["config"]["collective"].startswith("all_gather")
    # what a SCALE record can check the collective by: ranks seen, every rank's own step time, the timed all-gather
    mg = res["multi_gpu"]
    assert mg["ranks_seen"] == 2 and len(mg["per_rank_ms"]) == 2 and all(t > 0 for t in mg["per_rank_ms"]) and mg["collective_us"] > 0
    assert mg["devices"] == [0, 0] and mg["gathered_rows"] == 4 and max(mg["per_rank_ms"]) <= res["ms_per_step"] * 1.001
    # a failing child is this command's failure
    bad = subprocess.run(cmd + ["--model", "qwen", "--config", "gpm_pairwise"], capture_output=True, text=True, timeout=600, env=env)   # rejected by the ranks
    assert bad.returncode != 0
