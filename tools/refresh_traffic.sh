#!/bin/bash
# The two PMC passes behind profiles/<tag>_pmc_traffic.json and the bench lines that quote it -- the part of
# tools/make_profiles.sh that has to be repeated after ANY edit of kvarq_amd/csrc or include/ (bench.py only quotes a
# traffic file that carries the hash of the sources it runs on).
# usage (through gpurun, repo root): bash tools/refresh_traffic.sh round2
set -u
TAG=${1:-round2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BP="python3 $R/bench.py --preheat 0"
rm -rf $O/fetch $O/write
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $BP --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $BP --steps 2 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
cd $R
python3 tools/pmc_traffic.py $O/fetch $O/write 10000000 325 > $O/pmc_traffic.json
cp $O/pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json
timeout -k 10 400 python3 bench.py > $O/bench_n1.json 2> $O/bench.err
timeout -k 10 400 python3 bench.py --pipeline 1 --no-cpu-baseline > $O/bench_n1_pipeline1.json 2>> $O/bench.err
cat $O/bench_n1.json
