#!/usr/bin/env python3
"""Register budget of every gl_engine_kernel instantiation in a hipcc -save-temps .s file.
   scripts/kernel_regs.py file.s [substring]"""
import re, sys
txt = open(sys.argv[1]).read()
key = sys.argv[2] if len(sys.argv) > 2 else "gl_"
meta = txt[txt.index("amdhsa.kernels:"):]
for block in meta.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", block).group(1)
    if key not in name:
        continue
    g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", block).group(1))
    t = re.search(r"gl_engine_kernelILi(\d)ELi(\d)ELi(\d+)ELb([01])", name)
    tp = re.search(r"gl_persist_kernelILi(\d)ELi(\d+)E", name)
    label = f"A={t.group(1)} HB={t.group(2)} MODE={t.group(3)} SC={t.group(4)}" if t else (
        f"persist HB={tp.group(1)} MODE={tp.group(2)}" if tp else name[:60])
    print(f"{label:32s} vgpr {g('vgpr_count'):4d} spill {g('vgpr_spill_count'):3d} sgpr {g('sgpr_count'):4d} scratch {g('private_segment_fixed_size'):5d} lds {g('group_segment_fixed_size')}")
