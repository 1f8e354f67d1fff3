#!/usr/bin/env python3
"""Puts the issue-time ledger together (tools/r4_ledger.sh): per phase of kvq_scan_bp the vector instructions per read by cost class,
what they cost to issue, the scalar and LDS instructions, and the phase's measured share of the kernel time.

usage: python tools/r4_ledger.py <ledger_pmc.txt> <ledger_times.txt> <isa census>
"""
import re
import sys

READS = 10_000_000
SIMDS, CUS, GHZ = 1024, 256, 2.4
C2, C4 = 2.4, 4.3            # cycles of a SIMD's issue per wave64 instruction at 8 waves per SIMD (profiles/round3_valu_rate.txt)

pmc, cur = {}, None
for line in open(sys.argv[1]):
    m = re.match(r'== KVQ_DBG=(\d+)', line)
    if m:
        cur = int(m.group(1)); pmc[cur] = {}
    elif cur is not None and line.split():
        pmc[cur][line.split()[0]] = float(line.split()[1])
times = {}
for line in open(sys.argv[2]):
    m = re.match(r'KVQ_DBG=(\d+)\s+kernel ([0-9.]+) ms', line)
    if m:
        times[int(m.group(1))] = float(m.group(2))
census = {}
for line in open(sys.argv[3]):
    f = line.rstrip('\n').split()
    m = re.match(r'(.{24}) +(\d+) +(\d+) +(\d+) +(\d+) +(\d+) +(\d+) +(\d+)', line)
    if m:
        census[m.group(1).strip()] = tuple(int(m.group(i)) for i in range(2, 9))

# phases: (name, counters of "with" minus "without", the census regions whose instruction mix stands for it)
ALL = 1024
PHASES = [
    ('front end (P0-P2, tile top/end)', 32, None, ['P0 vectors', 'P0 vectors end', 'P1b', 'P1b end / P2', 'P2 end / P3 setup', 'tile end']),
    ('  of which P0 arithmetic (static: 5 vectors x thread)', None, None, ['P0 vectors']),
    ('trim (P3)', 130, 32, ['trim']),
    ('pass scaffolding (P3 set-up, rinfo, loop)', 2, 130, ['P4 end', 'P2 end / P3 setup']),
    ('seed filter (P3)', 1, 2, ['trim end / filter']),
    ('verification (P4)', ALL, 1, ['filter end / P4a', 'P4b']),
    ('whole kernel', ALL, None, None),
]


def mix(regions):
    f2 = sum(census[r][0] for r in regions if r in census); f4 = sum(census[r][1] for r in regions if r in census); ln = sum(census[r][2] for r in regions if r in census)
    tot = max(1, f2 + f4 + ln)
    return f2 / tot, f4 / tot, ln / tot


print('issue-time ledger of kvq_scan_bp<2,2> (10 M x 150 bp reads per launch, MTBC table; instruction counts: PMC, per read = per launch / 10 M;')
print('issue cost: wave-instructions x cycles of their class / (1024 SIMDs x 2.4 GHz); SALU: one per cycle and CU; time: un-profiled launches with the phase switched off)')
print()
print('%-52s %7s %7s %7s %7s %7s | %9s %9s %9s | %8s %6s' % ('phase', 'VALU/rd', '2-cyc', '4-cyc', 'lane', 'SALU/rd', 'VALU ms', 'SALU ms', 'LDS/rd', 'time ms', 'issue%'))
whole_mix = mix([r for r in census])
for name, w, wo, regions in PHASES:
    if w is None:
        f2, f4, ln = census['P0 vectors'][0], census['P0 vectors'][1], census['P0 vectors'][2]
        per_read = (f2 + f4 + ln) * 512 / (39760 / 325.0) / 64        # a tile of 39 760 bytes holds 122.3 reads; 512 threads x the static count, per wave-instruction
        valu_ms = per_read * READS * ((f2 * C2 + (f4 + ln) * C4) / (f2 + f4 + ln)) / SIMDS / (GHZ * 1e6)
        print('%-52s %7.1f %7.1f %7.1f %7.1f %7s | %9.3f %9s %9s | %8s %6s' % (name, per_read, per_read * f2 / (f2 + f4 + ln), per_read * f4 / (f2 + f4 + ln), per_read * ln / (f2 + f4 + ln), '', valu_ms, '', '', '', ''))
        continue
    if w not in pmc or (wo is not None and wo not in pmc):
        continue
    d = {k: pmc[w].get(k, 0) - (pmc[wo].get(k, 0) if wo is not None else 0) for k in pmc[w]}
    valu = d.get('SQ_INSTS_VALU', 0) / READS; salu = d.get('SQ_INSTS_SALU', 0) / READS; lds = d.get('SQ_INSTS_LDS', 0) / READS
    m2, m4, ml = mix(regions) if regions else whole_mix
    valu_ms = valu * READS * (m2 * C2 + (m4 + ml) * C4) / SIMDS / (GHZ * 1e6)
    salu_ms = salu * READS / CUS / (GHZ * 1e6)
    t = times.get(w, 0) - (times.get(wo, 0) if wo is not None else 0)
    print('%-52s %7.1f %7.1f %7.1f %7.1f %7.1f | %9.3f %9.3f %9.1f | %8.3f %5.0f%%' % (name, valu, valu * m2, valu * m4, valu * ml, salu, valu_ms, salu_ms, lds, t, 100 * valu_ms / t if t > 0 else 0))
print()
if ALL in pmc:
    p = pmc[ALL]
    wc = p.get('SQ_WAVE_CYCLES', 1)
    print('wave time of the whole kernel (SQ counters, quad-cycles): parked at s_waitcnt / s_barrier %.0f %%, ready but not issuing %.0f %%, issuing %.0f %%'
          % (100 * p.get('SQ_WAIT_ANY', 0) / wc, 100 * p.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * (wc - p.get('SQ_WAIT_ANY', 0) - p.get('SQ_WAIT_INST_ANY', 0)) / wc))
print('reading: a phase whose "issue%" is 80 or more is bound by vector issue -- only fewer or cheaper instructions help; the classes of each')
print('phase are those of its static ISA (profiles/round4_isa_census.txt), the counts are dynamic.')
