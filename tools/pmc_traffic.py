#!/usr/bin/env python3
"""HBM bytes per launch of the seed-filter kernel from two rocprofv3 --pmc passes.

usage: python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <reads per GPU> <record bytes> > profiles/roundN_pmc_traffic.json

Both passes run the same command (`python bench.py --steps 2 --warmup 1 --no-cpu-baseline`).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts half of the bytes of a
16-byte-per-lane streaming read, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact.
"""
import csv, glob, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def per_launch(root, counter, sub='kvq_scan_bp'):
    acc = {}
    for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if sub in r['Kernel_Name'] and r['Counter_Name'] == counter:
                    acc[r['Dispatch_Id']] = acc.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
    return [acc[k] for k in sorted(acc, key=int)]

def main():
    fdir, wdir, reads, rb = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    import bench                                 # (source_sha256: the hash bench.py checks before it quotes this file)
    kernel = 'kvq_scan_bp'
    head = os.environ.get('KVQ_GIT_HEAD')        # (the GPU box has no .git: the caller passes the commit along, tools/make_profiles.sh)
    if not head:
        try:
            head = subprocess.check_output(['git', 'rev-parse', 'HEAD'], cwd=os.path.dirname(os.path.abspath(bench.__file__)), stderr=subprocess.DEVNULL).decode().strip()
        except Exception:
            head = None                          # (the source hash is what ties the file to the code)
    f = per_launch(fdir, 'FETCH_SIZE'); w = per_launch(wdir, 'WRITE_SIZE')
    steps = 3 + 5                                # --steps 2 --warmup 1, and the five steps bench.py runs one at a time behind them for `all_kernels_ms_per_step` (round 4)
    launches_per_step = len(f) // steps
    fm = sum(f) / len(f); wm = sum(w) / len(w)
    alg = reads * rb / launches_per_step
    rd = 2.0 * fm * 1024.0; wr = wm * 1024.0
    out = {
        'command': 'KVQ_GRID=960 rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --preheat 0 --steps 2 --warmup 1 --no-cpu-baseline (separate passes; tools/make_profiles.sh; KVQ_GRID=960: the grid of the bench line\'s timed launches, which run with jobs in flight -- the counter passes serialise the launches, and a launch alone takes all 1024 workgroup slots: 1.56 x there, more tiles in flight per L2)',
        'kernel': kernel, 'reads_per_gpu': reads, 'launches_per_step': launches_per_step,
        'source_sha256': bench.source_sha256(), 'git_head': head,
        'note': 'gfx950: FETCH_SIZE counts 1/2 of the bytes of a 16-B-per-lane streaming read (MI355X_MICROARCH.md, HBM section), '
                'so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact. Per-launch values are the mean over the launches of the run. '
                'The text itself is streamed once (1.005 x the algorithmic bytes with every tile stopped behind the front end, profiles/round2_fetch_by_phase.txt); the rest are the two single-byte probes per record, the \'@\' and the \'+\' (workhorse.c:1037-1048), two in three of which miss L2 and count a whole line each (profiles/round3_probe_ablation.txt: without them 1.02 x; profiles/round4_probe_ablation.txt: the time of the kernel is the same with and without them).',
        'FETCH_SIZE_KB_per_launch_raw': f, 'FETCH_SIZE_KB_mean': fm,
        'WRITE_SIZE_KB_per_launch_raw': w, 'WRITE_SIZE_KB_mean': wm,
        'algorithmic_bytes_per_launch': alg,
        'hbm_read_bytes_per_launch': rd, 'hbm_write_bytes_per_launch': wr, 'hbm_bytes_per_launch': rd + wr,
        'traffic_over_algorithmic': (rd + wr) / alg,
    }
    print(json.dumps(out, indent=1))

if __name__ == '__main__':
    main()
