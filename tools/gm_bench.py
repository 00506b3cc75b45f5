#!/usr/bin/env python3
"""The tile walk's band height (GemmParams::gm: an XCD's 32 concurrent tiles form a gm x 32/gm patch) on the path's big GEMMs, default
operand form, interleaved in one process.  Band height trades A re-reads against W re-reads at the L2 (8 x 4 and 4 x 8 are the
optimum there) -- and decides whether W plus the A bands in flight fit the 256 MB Infinity Cache.
    python3 tools/gm_bench.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-reward_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from llava_reward_amd import _lib as L
from narrow_bench import mixed_case, timed      # (narrow_bench reads argv[1] = reps too)

CASES = [("llava.gate_up", 64 * 2200, 28672, 4096, L.EPI_SWIGLU_OP, 0), ("llava.down", 64 * 2200, 4096, 14336, L.EPI_RESADD_F32, 0),
         ("dec.gate_up", 84544, 16384, 3072, L.EPI_SWIGLU_OP, 32), ("dec.qkv", 84544, 9216, 3072, L.EPI_OUT_OP, 32),
         ("dec.o", 84544, 3072, 3072, L.EPI_RESADD_F32, 32), ("dec.down", 84544, 3072, 8192, L.EPI_RESADD_F32, 32),
         ("clip.qkv", 313888, 3072, 1024, L.EPI_OUT_OP, 23), ("clip.out", 313888, 1024, 1024, L.EPI_RESADD_F32, 23),
         ("clip.fc1", 313888, 4096, 1024, L.EPI_OUT_OP, 23), ("clip.fc2", 313888, 1024, 4096, L.EPI_RESADD_F32, 23)]
GMS = tuple(sys.argv[2].split(",")) if len(sys.argv) > 2 else ("8", "4", "2", "16")
print(f"{'GEMM':12s} " + " ".join(f"{'gm=' + g + ' ms':>10s}" for g in GMS))
tot = {g: 0.0 for g in GMS}
for name, M, N, K, epi, cnt in CASES:
    fn = mixed_case(M, N, K, epi)
    t = {g: 1e9 for g in GMS}
    for rnd in range(2):
        for g in (GMS if rnd == 0 else GMS[::-1]):
            os.environ["LR_GEMM_GM"] = g.rstrip("e")
            os.environ["LR_GEMM_BANDCHUNK"] = "0" if g.endswith("e") else "1"          # "4e": equal chunks (the XCDs start anywhere in a band)
            t[g] = min(t[g], timed(fn) / 1e3)
    os.environ.pop("LR_GEMM_GM", None)
    os.environ.pop("LR_GEMM_BANDCHUNK", None)
    for g in GMS:
        tot[g] += t[g] * cnt
    print(f"{name:12s} " + " ".join(f"{t[g]:10.3f}" for g in GMS), flush=True)
print(f"{'per pass':12s} " + " ".join(f"{tot[g]:10.1f}" for g in GMS))
