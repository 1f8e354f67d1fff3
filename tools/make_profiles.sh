#!/bin/bash
# Regenerates the measurement artifacts under profiles/ on an MI355X box (run through gpurun from the
# repo root: `gpurun --timeout 1200 -- "KVQ_GIT_HEAD=$(git rev-parse HEAD) bash tools/make_profiles.sh round4"`; round 4 added part c).  Everything is written
# under gpurun_out/<tag>/ ; copy what is to be kept into profiles/ afterwards (the script prints the cp lines).
# PMC passes are separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes.
set -u
TAG=${1:-round4}
PART=${2:-abc}           # a: traces, counters and the bench line; b: the other measurements; c (round 4): issue ledger, P4 counters, settings sweep, two-rank rehearsal (gpurun calls of <= 20 min each)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-end-to-end"
BP="$B --preheat 0"          # counter passes: every launch is counted, no need to heat the clocks
step() { echo "== $1" >> $O/progress.txt; }

if [[ $PART == *a* ]]; then
step stats;  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 5 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
step stats1; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- $B --pipeline 1 --steps 5 --warmup 1 --no-cpu-baseline > $O/stats1.log 2>&1
# (the counter passes serialise the launches: each would get every workgroup slot, as a job run alone does; the timed launches of the bench line
# run with jobs in flight, on 960 workgroups -- KVQ_GRID=960 gives the counted launches the timed ones' grid)
step fetch;  KVQ_GRID=960 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $BP --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
step write;  KVQ_GRID=960 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $BP --steps 2 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
step sq1;    timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq1 -- $BP --steps 2 --warmup 1 --no-cpu-baseline > $O/sq1.log 2>&1
step sq2;    timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq2 -- $BP --steps 2 --warmup 1 --no-cpu-baseline > $O/sq2.log 2>&1
cd $R
python3 tools/pmc_traffic.py $O/fetch $O/write 10000000 325 > $O/pmc_traffic.json
python3 tools/pmc_sum.py $O/sq1 kvq_scan_ > $O/pmc_sq_counters.txt; python3 tools/pmc_sum.py $O/sq2 kvq_scan_ >> $O/pmc_sq_counters.txt
python3 tools/trace_step.py $O/stats1 1 > $O/step_timeline.txt                 # one step at a time: the tail kernels at their own speed
python3 tools/trace_step.py $O/stats 2 > $O/step_timeline_pipelined.txt        # the default: three steps in flight
( echo '== three steps in flight (the default)'; python3 tools/kernel_agreement.py $O/stats $O/stats.log 5; echo '== one step at a time (--pipeline 1)'; python3 tools/kernel_agreement.py $O/stats1 $O/stats1.log 5 ) > $O/kernel_agreement.txt 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/rocprofv3_kernel_stats.csv 2>/dev/null
cp $(ls $O/stats1/*/*kernel_stats.csv | head -1) $O/rocprofv3_kernel_stats_pipeline1.csv 2>/dev/null
step stamps; KVQ_DBG=16 timeout -k 10 200 python3 tools/phase_stamps.py > $O/phase_stamps.txt 2>&1
step hosttimes; timeout -k 10 200 python3 tools/step_host_times.py > $O/step_host_times.txt 2>&1
step phases
for d in 0 1 2 32; do
  ( cd /tmp; KVQ_DBG=$d timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/ph$d -- $BP --reads 5000000 --steps 2 --warmup 1 --no-cpu-baseline > $O/ph$d.log 2>&1 )
  echo "== KVQ_DBG=$d (0: whole kernel, 1: no verify, 2: no filter and verify, 32: front end only); 5 M reads per launch" >> $O/pmc_phases.txt
  python3 tools/pmc_sum.py $O/ph$d kvq_scan_ >> $O/pmc_phases.txt
done
fi
if [[ $PART == *b* ]]; then
cd $R
step configs; bash tools/other_configs.sh $TAG > /dev/null 2>&1
step filerate; timeout -k 10 200 python3 tools/file_rate.py > $O/file_rate.txt 2>&1
step filepath; KVQ_TIMING=1 timeout -k 10 200 python3 tools/r3_file.py 10000000 4 8 16 2>&1 | grep 'stream_batches\|plain file\|findseqs:' > $O/file_path.txt
step long; ( timeout -k 10 200 python3 tools/realistic_bench.py 3000000; timeout -k 10 200 python3 tools/realistic_bench.py 3000000 100000; LONG_EVERY=100000 bash tools/r3_long_trace.sh ) 2>&1 | grep -v amdgpu.ids > $O/long_reads.txt
step gaps; ( timeout -k 10 200 python3 tools/r3_gap.py 10000000 40 3; timeout -k 10 200 python3 tools/r3_gap2.py; KVQ_GRID=992 timeout -k 10 200 python3 tools/r3_gap.py 10000000 40 3 | sed 's/^/KVQ_GRID=992 (round 2: 32 free slots)  /' ) 2>&1 | grep -v amdgpu.ids > $O/scan_gaps.txt
step gridsweep; bash tools/r3_grid_sweep.sh "1024 992 976 960 944" > $O/grid_sweep.txt 2>&1
step dense; ( for k in 8 16 32; do timeout -k 10 120 python3 tools/r3_dense.py $k 1000000 | tail -1; done ) > $O/dense_tables.txt 2>&1
step h2dflags; timeout -k 10 200 python3 tools/r3_h2d_flags.py 64 8 2>&1 | grep -v amdgpu.ids > $O/host_ceiling_flags.txt
step hostceiling; timeout -k 10 200 python3 tools/r3_h2d.py 64 4 8 16 2>&1 | grep -v amdgpu.ids > $O/host_ceiling.txt
step probes; bash tools/r3_probe_ablation.sh > $O/probe_ablation.txt 2>&1
step sizes; bash tools/r3_sizes.sh > $O/kernel_time_by_size.txt 2>&1
step occ; bash tools/r3_occ.sh > $O/time_by_occupancy.txt 2>&1
step clock; bash tools/clock_pmc.sh $TAG/clk > $O/clock.txt 2>&1
step ubench
( cd tools/ubench && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip 2>/dev/null && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o issue_rate issue_rate.hip 2>/dev/null
  timeout -k 10 120 ./valu_rate > $O/valu_rate.txt 2>&1 && timeout -k 10 120 ./issue_rate > $O/issue_rate.txt 2>&1 )
fi
if [[ $PART == *c* ]]; then
cd $R
step ledger; bash tools/r4_ledger.sh $TAG > /dev/null 2>&1
step p4counters
if [ -f kvarq_amd/abx/tally.so ]; then
  cp kvarq_amd/libkvarq_hip.so /tmp/lib_keep.so; cp kvarq_amd/abx/tally.so kvarq_amd/libkvarq_hip.so
  ( echo "what P4 eats (instrumented build with -DKVQ_TALLY, 2.5 M reads of the bench workload; tools/phase_stamps.py):"; KVQ_DBG=16 timeout -k 10 200 python3 tools/phase_stamps.py 2>&1 | grep "^P4" ) > $O/p4_counters.txt
  cp /tmp/lib_keep.so kvarq_amd/libkvarq_hip.so
fi
step settings
( echo "engine settings next to the product's (10 M x 150 bp, MTBC table; bench.py --maxerrors / --minoverlap; kvq_seed_k picks the seed length):"
  for args in "--maxerrors 0" "--maxerrors 1" "--maxerrors 2" "--maxerrors 3" "--maxerrors 1 --minoverlap 20" "--maxerrors 3 --minoverlap 35"; do
    timeout -k 10 300 python3 bench.py $args --no-cpu-baseline --no-end-to-end --steps 5 2> $O/cfg.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-34s step %.3f ms  kernel %.3f ms  roofline %.4f  %.2f G reads/s  %s' % (sys.argv[1], d['ms_per_step'], r['avg_launch_ms'], r['frac'], d['value']/1e9, d['config']['kernel_path']))" "$args"
  done ) > $O/settings_sweep.txt 2>&1
step tworanks
# `python bench.py --gpus 2` by itself starts two ranks; on a one-GPU box they share the card and rendezvous over gloo (RCCL refuses two ranks on one device)
( KVQ_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --reads 2000000 --steps 5 --no-cpu-baseline --no-end-to-end > $O/bench_n2_gloo_rehearsal.json 2> $O/bench_n2.err; tail -2 $O/bench_n2.err >> $O/bench_n2_gloo_rehearsal.json ) 
fi
if [[ $PART == *a* ]]; then
cd $R
# the bench line last: it quotes the traffic file made above (same sources)
step bench;  cp $O/pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json; timeout -k 10 500 python3 bench.py > $O/bench_n1.json 2> $O/bench.err
timeout -k 10 400 python3 bench.py --pipeline 1 --no-cpu-baseline --no-end-to-end > $O/bench_n1_pipeline1.json 2>> $O/bench.err
fi
step done
for f in bench_n1.json kernel_agreement.txt bench_n1_pipeline1.json rocprofv3_kernel_stats_pipeline1.csv step_timeline_pipelined.txt rocprofv3_kernel_stats.csv pmc_traffic.json pmc_sq_counters.txt step_timeline.txt phase_stamps.txt step_host_times.txt pmc_phases.txt other_configs.txt file_rate.txt file_path.txt long_reads.txt scan_gaps.txt grid_sweep.txt dense_tables.txt host_ceiling.txt host_ceiling_flags.txt probe_ablation.txt kernel_time_by_size.txt time_by_occupancy.txt clock.txt valu_rate.txt issue_rate.txt issue_ledger.txt ledger_pmc.txt ledger_times.txt p4_counters.txt settings_sweep.txt bench_n2_gloo_rehearsal.json; do
  echo "cp gpurun_out/$TAG/$f profiles/${TAG}_$f"
done
[[ $PART == *a* ]] && cat $O/bench_n1.json
