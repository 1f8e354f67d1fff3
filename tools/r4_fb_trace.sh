#!/bin/bash
# kernel traces of the bench with and without kvq_scan_finish_begin (--no-finish-begin): gpurun_out/fb/*_tail.txt show three scans of the
# pipelined phase, kernel by kernel with their hardware queues
# usage (through gpurun, repo root): bash tools/r4_fb_trace.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/fb; rm -rf $O; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/with -- python3 $R/bench.py --no-end-to-end --steps 6 --warmup 1 --preheat 3 --no-cpu-baseline > $O/with.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/without -- python3 $R/bench.py --no-end-to-end --steps 6 --warmup 1 --preheat 3 --no-cpu-baseline --no-finish-begin > $O/without.log 2>&1
cd $R
python3 tools/r4_trace_tail.py $(find $O/with -name "*kernel_trace.csv" | head -1) 11 > $O/with_tail.txt
python3 tools/r4_trace_tail.py $(find $O/without -name "*kernel_trace.csv" | head -1) 11 > $O/without_tail.txt
rm -rf $O/with $O/without
