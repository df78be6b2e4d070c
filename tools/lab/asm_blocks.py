#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S listing:  asm_blocks.py file.s <mangled-name-substring> [min_mfma]"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
text = open(src).read().split('\n')
on = False
stats = []
cur = None
for i, l in enumerate(text):
    if not on:
        if re.match(r'^_Z\S*' + re.escape(key) + r'\S*:', l):
            on = True
            cur = dict(name='entry', n=0, mfma=0, sst=0, sld=0, acc=0, valu=0, ds=0, vmem=0, salu=0, line=i)
        continue
    if 's_endpgm' in l:
        break
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        stats.append(cur)
        cur = dict(name=m.group(1), n=0, mfma=0, sst=0, sld=0, acc=0, valu=0, ds=0, vmem=0, salu=0, line=i)
        continue
    t = l.strip()
    if not t or t[0] in ';.':
        continue
    cur['n'] += 1
    if t.startswith('v_mfma'): cur['mfma'] += 1
    elif t.startswith('scratch_store'): cur['sst'] += 1
    elif t.startswith('scratch_load'): cur['sld'] += 1
    elif t.startswith('v_accvgpr'): cur['acc'] += 1
    elif t.startswith('ds_'): cur['ds'] += 1
    elif t.startswith('buffer_') or t.startswith('global_'): cur['vmem'] += 1
    elif t.startswith('v_'): cur['valu'] += 1
    elif t.startswith('s_'): cur['salu'] += 1
stats.append(cur)
tot = lambda k: sum(s[k] for s in stats)
print(f"blocks {len(stats)}  instrs {tot('n')}  mfma {tot('mfma')}  scratch st/ld {tot('sst')}/{tot('sld')}  accvgpr {tot('acc')}")
for s in stats:
    if s['mfma'] >= min_mfma:
        print(s)
