#!/usr/bin/env python3
"""Instruction counts between the KVQMARK comments of a kernel's ISA text (straight-line order).

usage: python tools/isa_marks.py <file.s> <kernel name substring>
"""
import sys, collections
FAST = ('v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_not_b32', 'v_mov_b32_e32', 'v_lshrrev_b32', 'v_ashrrev_i32', 'v_bitop3_b32', 'v_cndmask_b32_e32')
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2]); i = s.index('\n', s.index('@function', i)); j = s.index('.Lfunc_end', i)
cur = 'start'; acc = collections.OrderedDict()
def slot(): return acc.setdefault(cur, collections.Counter())
for l in s[i:j].split('\n'):
    t = l.strip()
    if 'KVQMARK' in t: cur = t.split('KVQMARK', 1)[1].strip(); continue
    if not t or t[0] in '.;' or t.endswith(':'): continue
    op = t.split()[0]; c = slot()
    if op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): c['lane'] += 1; c['vslow'] += 1
    elif op.startswith('v_'):
        c['valu'] += 1
        c['vfast' if op.startswith(FAST) else 'vslow'] += 1
    elif op.startswith('s_nop'): c['nop'] += 1
    elif op.startswith('s_waitcnt'): c['wait'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    elif op.startswith('ds_'): c['lds'] += 1
    elif op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): c['vmem'] += 1
for k, c in acc.items():
    print('%-28s valu %4d (fast %4d slow %4d, lane ops %3d)  salu %4d  nop %3d  lds %3d  vmem %3d  wait %3d' % (k, c['valu'], c['vfast'], c['vslow'], c['lane'], c['salu'], c['nop'], c['lds'], c['vmem'], c['wait']))
