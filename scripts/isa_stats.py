#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -save-temps .s file.
   scripts/isa_stats.py file.s <substring of kernel symbol>"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
txt = open(path).read()
# kernel body: from "<sym>:" to ".Lfunc_end"
syms = [m.group(1) for m in re.finditer(r'^(_Z\S+):', txt, re.M) if key in m.group(1)]
for sym in syms:
    start = txt.index("\n" + sym + ":")
    end = txt.index(".Lfunc_end", start)
    body = txt[start:end].splitlines()
    cnt = collections.Counter()
    for line in body:
        line = line.strip()
        if not line or line.startswith((';', '.', '//')) or line.endswith(':'):
            continue
        op = line.split()[0]
        if op.startswith('v_'):
            cls = 'valu'
            if op.startswith('v_pk_'): cls = 'valu_pk'
            if op.startswith(('v_rcp','v_sqrt','v_rsq','v_exp','v_log','v_sin','v_cos')): cls='trans'
        elif op.startswith('ds_'): cls = op
        elif op.startswith(('global_','buffer_','flat_','scratch_')): cls = op
        elif op.startswith('s_waitcnt'): cls = 's_waitcnt'
        elif op.startswith('s_'): cls = 'salu'
        else: cls = op
        cnt[cls] += 1
    print(sym[:90])
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
        print(f"   {k:28s} {v}")
