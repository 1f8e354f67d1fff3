set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcph; mkdir -p $O
for d in 0 1 2 32; do
  export KVQ_DBG=$d
  timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/d$d -- python3 bench.py --reads 5000000 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_d$d.log 2>&1
  echo "== dbg $d" >> $O/summary.txt
  python3 tools/pmc_sum.py $O/d$d >> $O/summary.txt
done
cat $O/summary.txt
