#!/bin/bash
# Round 4: the issue-time ledger of kvq_scan_bp.  Phases are switched off one by one with the KVQ_DBG switches of the
# diagnostic instantiation (same code as the production kernel plus the switches): instruction counts per phase from PMC passes
# (SQ_INSTS_*; 10 M reads per launch), times from un-profiled runs of the same switches, the cost classes of each phase's vector
# instructions from the static census of its ISA (profiles/round4_isa_census.txt, made where hipcc is), the per-class issue cost from
# profiles/round3_valu_rate.txt (8 waves per SIMD: 2.4 cycles for the two-cycle class, 4.3 for the four-cycle one and for lane operations).
# usage (through gpurun, repo root): bash tools/r4_ledger.sh <tag>
set -u
TAG=${1:-ledger}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --preheat 0 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --pipeline 1"
: > $O/ledger_pmc.txt; : > $O/ledger_times.txt
for d in 64 1 2 130 32; do       # 64: every switch off but the kernel is the diagnostic instantiation (tile text from the first 64 tiles only would be wrong: see below)
  dd=$d; [ $d = 64 ] && dd=1024  # (1024 is no switch at all: it only selects the diagnostic instantiation, so that all five rows run the same code)
  KVQ_DBG=$dd timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/p$dd -- $B > $O/p$dd.log 2>&1
  echo "== KVQ_DBG=$dd" >> $O/ledger_pmc.txt
  python3 $R/tools/pmc_sum.py $O/p$dd kvq_scan_bp >> $O/ledger_pmc.txt
  rm -rf $O/p$dd
  KVQ_DBG=$dd timeout -k 10 120 python3 $R/tools/kernel_time.py 10000000 20 >> $O/ledger_times.txt 2>&1
done
cd $R
python3 tools/r4_ledger.py $O/ledger_pmc.txt $O/ledger_times.txt profiles/round4_isa_census.txt > $O/issue_ledger.txt
cat $O/issue_ledger.txt
