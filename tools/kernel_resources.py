#!/usr/bin/env python3
"""Register / LDS / scratch use of the kernels of one HIP source (device-only assembly, code-object metadata).
   python tools/kernel_resources.py bayesian-inference_amd/csrc/k_front.hip [name-regex]"""
import re
import subprocess
import sys

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "."
asm = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", "-"],
                     capture_output=True, text=True, check=True).stdout
for blk in asm.split("- .agpr_count")[1:]:
    m = re.search(r"\.name:\s+(\S+)", blk)
    if not m or not re.search(pat, m.group(1)):
        continue
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    g = lambda k: re.search(r"\." + k + r":\s+(\d+)", blk).group(1)
    print(f"{name[:100]:100s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>4s} sgpr {g('sgpr_count'):>4s} "
          f"lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}")
