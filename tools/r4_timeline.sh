#!/bin/bash
# kernel timelines of one bench step (one step at a time / three in flight) from rocprofv3 kernel traces: gpurun_out/tl/step_timeline*.txt
# usage (through gpurun, repo root): bash tools/r4_timeline.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/tl; rm -rf $O; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 $R/bench.py --no-end-to-end --pipeline 1 --steps 5 --warmup 1 --no-cpu-baseline > $O/stats1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-end-to-end --steps 5 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
cd $R; python3 tools/trace_step.py $O/stats1 1 > $O/step_timeline.txt; python3 tools/trace_step.py $O/stats 2 > $O/step_timeline_pipelined.txt
rm -rf $O/stats $O/stats1
