cd ${GRAFT_REPO_ROOT:-.}
cp kvarq_amd/libkvarq_hip.so /tmp/lib_orig.so
for f in kvarq_amd/ab/d*.so; do
  cp $f kvarq_amd/libkvarq_hip.so
  for args in "--maxerrors 3" "--table-scale 8 --reads 4000000" "--table-scale 32 --reads 1000000"; do
    python3 bench.py $args --no-cpu-baseline --no-end-to-end --steps 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-12s %-36s step %.3f ms kernel %.3f ms frac %.4f hits %d' % (sys.argv[2], sys.argv[1], d['ms_per_step'], r['avg_launch_ms'], r['frac'], d['config']['hits_per_step']))" "$args" $(basename $f)
  done
done
cp /tmp/lib_orig.so kvarq_amd/libkvarq_hip.so
