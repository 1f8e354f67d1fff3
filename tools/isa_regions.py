#!/usr/bin/env python3
"""Static instruction mix of the seed-filter kernel between its s_memtime phase stamps.

usage: python tools/isa_regions.py <kernel.s>   (ISA text of one kernel)
"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
regions = []; cur = None
def fresh(i): return {'start': i, 'valu': 0, 'slow': 0, 'b64': 0, 'salu': 0, 'lds': 0, 'vmem': 0, 'lane': 0, 'labels': 0}
cur = fresh(0)
for i, l in enumerate(lines):
    t = l.strip()
    if not t: continue
    op = t.split()[0]
    if op.startswith('s_memtime'):
        regions.append(cur); cur = fresh(i)
    if op.startswith('.LBB'): cur['labels'] += 1
    if op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'): cur['lane'] += 1
    elif op.startswith('v_'):
        cur['valu'] += 1
        if re.match(r'v_mul_lo_u32|v_mul_hi|v_mad_u64|v_mad_i64', op): cur['slow'] += 1
        if re.search(r'(b64|u64|i64)', op): cur['b64'] += 1
    elif op.startswith('s_'): cur['salu'] += 1
    elif op.startswith('ds_'): cur['lds'] += 1
    elif op.startswith('global_') or op.startswith('buffer_'): cur['vmem'] += 1
regions.append(cur)
for r in regions: print(r)
