"""Registers / spills / LDS of every kernel in an AMDGPU assembly file (hipcc --save-temps *.s).
usage: python tools/kernel_regs.py file.s [name filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    f = {k: v for k, v in re.findall(r"\.(\w+):\s+(\S+)", blk)}
    name = f.get("name", "?")
    try:
        name = subprocess.run(["c++filt", name], stdout=subprocess.PIPE).stdout.decode().strip()
    except OSError:
        pass
    name = name.split("(")[0]
    if flt in name:
        print("%-70s vgpr %3s spill %3s scratch %4s lds %6s sgpr %3s" % (
            name[-70:], f.get("vgpr_count"), f.get("vgpr_spill_count"), f.get("private_segment_fixed_size"),
            f.get("group_segment_fixed_size"), f.get("sgpr_count")))
