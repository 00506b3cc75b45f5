#!/usr/bin/env python3
"""Register / scratch usage of every kernel in a `-save-temps` gfx950 assembly file (.s): name, VGPRs, AGPRs, scratch bytes, spills.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -c csrc/gemm8.hip -save-temps=obj -o /tmp/x.o ; python tools/kernel_regs.py /tmp/x-hip-*.s [filter]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = s.split("  - .agpr_count:")[1:]
for b in blocks:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)
    name = g("name")
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dem:
        continue
    ag = re.match(r"\s*(\d+)", b).group(1)
    print(f"{dem[:110]:110s} vgpr {g('vgpr_count'):>3s} agpr {ag:>3s} scratch {g('private_segment_fixed_size'):>4s} "
          f"spill s{g('sgpr_spill_count')} v{g('vgpr_spill_count')}")
