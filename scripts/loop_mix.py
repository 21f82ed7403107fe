#!/usr/bin/env python3
"""Instruction mix of the largest loop (the frame loop) of gl_engine_kernel instantiations in a
hipcc -save-temps .s file.   scripts/loop_mix.py file.s Li1ELi2ELi9ELb0 [...]"""
import re, sys, collections
txt = open(sys.argv[1]).read()
for key in sys.argv[2:]:
    for sym in [m.group(1) for m in re.finditer(r'^(_Z\S+):', txt, re.M) if key in m.group(1) and 'gl_engine' in m.group(1)]:
        start = txt.index("\n" + sym + ":"); end = txt.index(".Lfunc_end", start)
        lines = txt[start:end].splitlines()
        labels = {l.strip()[:-1]: i for i, l in enumerate(lines) if re.match(r'^\.LBB\S+:', l.strip())}
        loops = []
        for i, l in enumerate(lines):
            m = re.match(r'\s*s_c?branch\S*\s+(\.LBB\S+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        a, b = max(loops, key=lambda ab: ab[1] - ab[0])
        cnt = collections.Counter(); ops = collections.Counter()
        for l in lines[a:b]:
            l = l.strip()
            if not l or l.startswith((';', '.', '//')) or l.endswith(':'): continue
            op = l.split()[0]
            if op.startswith('v_pk_'): c = 'valu_pk'
            elif op.startswith(('v_rcp', 'v_sqrt', 'v_rsq')): c = 'trans'
            elif op.startswith('v_permlane'): c = 'permlane'
            elif op.startswith('v_mov') and 'dpp' in l: c = 'dpp_mov'
            elif op.startswith('v_mov') or op.startswith('v_accvgpr'): c = 'v_mov'
            elif op.startswith('v_'): c = 'valu_other'; ops[op] += 1
            elif op.startswith(('ds_', 'global_', 'scratch_', 'buffer_')): c = op
            elif op.startswith('s_waitcnt'): c = 's_waitcnt'
            elif op.startswith('s_'): c = 'salu'
            else: c = op
            cnt[c] += 1
        nv = sum(v for k, v in cnt.items() if k in ('valu_pk', 'trans', 'permlane', 'dpp_mov', 'v_mov', 'valu_other'))
        print(key, "loop lines", b - a, "vector", nv)
        print("  ", dict(sorted(cnt.items(), key=lambda kv: -kv[1])))
        print("   other valu:", dict(ops.most_common(14)))
