#!/usr/bin/env python3
"""Static census of a kernel's ISA between its KVQMARK comments: vector instructions by cost class (two-cycle / four-cycle per
wave64 on gfx950's SIMD-32, profiles/round3_valu_rate.txt; lane operations = v_readlane / v_writelane / v_readfirstlane, the
reloads of scalars spilled to VGPR lanes), scalar, LDS and vector-memory instructions.  Run where hipcc is (no GPU needed):

    hipcc -O3 -std=c++17 --offload-arch=gfx950 --offload-device-only -S -o unity.s kvarq_amd/csrc/kvq_unity.hip
    python tools/r4_isa_census.py unity.s _Z11kvq_scan_bpILi2ELi2ELb0ELb0ELi8ELb0E > profiles/round4_isa_census.txt
"""
import collections
import re
import sys

FAST = ('v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_not_b32', 'v_mov_b32', 'v_lshrrev_b32',
        'v_ashrrev_i32', 'v_bitop3_b32', 'v_cndmask_b32')


def census(path, kernel):
    s = open(path).read()
    i = s.index(kernel)
    i = s.index('\n', s.index('@function', i))
    j = s.index('.Lfunc_end', i)
    cur, acc = 'prologue + tile top', collections.OrderedDict()
    for line in s[i:j].split('\n'):
        t = line.strip()
        if 'KVQMARK' in t:
            cur = t.split('KVQMARK', 1)[1].strip()
            continue
        if not t or t[0] in '.;' or t.endswith(':'):
            continue
        op = re.sub(r'_e32$|_e64$|_sdwa$', '', t.split()[0])
        c = acc.setdefault(cur, collections.Counter())
        if op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')):
            c['lane'] += 1
        elif op.startswith('v_'):
            dpp = 'dpp' in t or 'row_' in t or 'quad_perm' in t
            c['fast' if op in FAST and not dpp else 'slow'] += 1
            c['op:' + op + ('_dpp' if dpp and not op.endswith('_dpp') else '')] += 1
        elif op.startswith('s_nop'):
            c['nop'] += 1
        elif op.startswith('s_waitcnt'):
            c['wait'] += 1
        elif op.startswith('s_barrier'):
            c['barrier'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
            c['vmem'] += 1
    return acc


if __name__ == '__main__':
    acc = census(sys.argv[1], sys.argv[2])
    print('kernel', sys.argv[2])
    print('%-24s %5s %5s %5s %5s %5s %5s %5s   %s' % ('region', '2-cyc', '4-cyc', 'lane', 'salu', 'lds', 'vmem', 'wait', 'the four-cycle ones'))
    for k, c in acc.items():
        slow = sorted(((n, o[3:]) for o, n in c.items() if o.startswith('op:') and o[3:].replace('_dpp', '') not in FAST or o.endswith('_dpp')), reverse=True)
        print('%-24s %5d %5d %5d %5d %5d %5d %5d   %s' % (k, c['fast'], c['slow'], c['lane'], c['salu'], c['lds'], c['vmem'], c['wait'],
                                                           ', '.join('%s %d' % (o, n) for n, o in slow[:8])))
