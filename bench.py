#!/usr/bin/env python3
"""
bench.py -- FastQ reads/sec scanned on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (engine.findseqs' scan: record split,
quality trim, match against the MTBC-shaped table, hits + per-sequence counters
+ coverage/mutation counts back on the host) over one batch of synthetic FastQ
that is already resident in HBM when the timed region starts.  Workload at N=1:
BASELINE.json configs[2], 10 M x 150 bp reads vs the MTBC table (132 templates,
both strands = 264 sequences) -- the configuration the north-star target is
quoted on; it fits one GPU (3.25 GB).  With N > 1 every rank scans its own
`--reads` records (weak scaling, reads shard without any data-path collective)
and the counter arrays are summed with one RCCL all-reduce per step.

    python bench.py                                   # N=1, 10 M reads
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N

Before the W warmup steps the setup runs `--preheat` (30) untimed scans of the same batch: in a
fresh process the GPU clocks take about ten steps (20 ms) to settle, and without it the first timed
steps of a short run measure the ramp (1.63 ms per launch instead of 1.58); `--preheat 0` switches it off.

Prints ONE JSON line on rank 0 (contract in the task description): value =
whole-job reads/s, roofline = algorithmic bytes (2L+25 per record) of the
dominant kernel / its HIP-event time vs 8 TB/s HBM peak, cpu_baseline = the
CPU engine timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def analytic_chunk_offsets(n_records, rb, L):
    """chunk cuts of fastq_read (workhorse.c:737-956) for fixed-size synthetic
    records: the cut falls on the last record whose '+' line starts inside the
    1 MiB buffer (verified against the real chunker on a prefix, below)"""
    total = n_records * rb
    offs, cs = [0], 0
    while True:
        end = cs + (1 << 20)
        if end > total or total - cs < (1 << 20):
            break
        if end == total:
            pass
        r = (end - 1 - (L + 22)) // rb
        cut = r * rb
        if cut <= cs:
            raise RuntimeError('record larger than the scan buffer')
        offs.append(cut)
        cs = cut
    offs.append(total)
    return np.array(offs, dtype=np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--reads', type=int, default=10_000_000, help='records per GPU')
    ap.add_argument('--readlen', type=int, default=150)
    ap.add_argument('--table', default='MTBC', choices=['MTBC', 'MTBC+barcodes'])
    ap.add_argument('--table-scale', type=int, default=1)
    ap.add_argument('--maxerrors', type=int, default=2, help='engine setting (product default 2; kvarq/cli.py:410-427 exposes it)')
    ap.add_argument('--minoverlap', type=int, default=25, help='engine setting (product default 25)')
    ap.add_argument('--exhaustive', action='store_true', help='force the exhaustive kernel for every sequence')
    ap.add_argument('--no-finish-begin', action='store_true', help='(measurement) do not enqueue a job\'s tail ahead of its finish')
    ap.add_argument('--pipeline', type=int, default=0,
                    help='scans in flight: step k+1 is enqueued (on its own scan object and stream) before the results of step k '
                         'are waited for, so the GPU goes from one scan kernel to the next while the host collects a step '
                         '(1: strictly one after the other; default: 3 when a step is one batch, else 1 -- the batches of a step already follow each other on one stream)')
    ap.add_argument('--batch-bytes', type=int, default=(1 << 32) - (1 << 20), help='largest batch handed to kvq_scan_device')
    ap.add_argument('--preheat', type=int, default=30, help='untimed scans before the warmup steps (the GPU clocks take ~10 steps = 20 ms to settle in a fresh process)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-end-to-end', action='store_true', help='skip the file -> engine.findseqs -> Python result measurement behind the timed region')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='target duration of the CPU baseline sample')
    ap.add_argument('--total-reads', type=int, default=0,
                    help='records of the WHOLE job, split evenly over the ranks (strong scaling): `--gpus 8 --total-reads 40000000` is '
                         'BASELINE.json configs[3]; without it every rank scans --reads records (weak scaling)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` by itself: this process becomes the launcher of N ranks (the join of the reference's
        # worker threads, workhorse.c:1375-1447, is a join of processes here) and never touches the GPU
        raise SystemExit(spawn_ranks(args.gpus))
    if os.environ.get('KVQ_BENCH_CHILD_PROBE') == 'fail' and os.environ.get('RANK') == '1':
        raise SystemExit(3)                         # (tests: a rank that dies takes the launcher's exit code with it)
    if os.environ.get('KVQ_BENCH_CHILD_PROBE'):
        # (tests/test_host_logic.py: what a rank is started with, printed before anything is imported)
        print(json.dumps({'rank': int(os.environ.get('RANK', '0')), 'local_rank': int(os.environ.get('LOCAL_RANK', '0')),
                          'world': int(os.environ.get('WORLD_SIZE', '1')), 'master': '%s:%s' % (os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT')),
                          'gpu_modules_loaded': [m for m in ('torch', 'kvarq_amd._lib') if m in sys.modules]}))
        return

    # (RCCL and gloo print banners on stdout when communicators are made; the contract is ONE JSON line there: everything the
    # setup prints goes to stderr instead -- file descriptor 1 is pointed at 2 until the line is due)
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # backend nccl = RCCL over xGMI; KVQ_BENCH_BACKEND=gloo only exists to rehearse the
        # multi-rank flow on a one-GPU box (ranks then share the device)
        backend = os.environ.get('KVQ_BENCH_BACKEND', 'nccl')
        local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)

    from kvarq_amd import _lib, scan, synth
    from kvarq_amd import dist as kdist
    L_ = _lib.lib()
    if L_.kvq_device_count() <= 0:
        raise SystemExit('bench.py needs an MI355X: libkvarq_hip has no CPU path')
    if L_.kvq_set_device(local):
        raise SystemExit('cannot select device %d' % local)

    # ---- workload: resident in HBM before anything is timed -------------------
    L = args.readlen
    rb = synth.record_bytes(L)
    if args.total_reads:
        if args.total_reads % world:
            raise SystemExit('--total-reads must be a multiple of the number of ranks')
        args.reads = args.total_reads // world
    n = args.reads
    first = rank * n
    g = synth.genome()
    plus = synth.table(g, args.table, scale=args.table_scale)
    seqs = synth.both_strands(plus)
    cfg = dict(maxerrors=args.maxerrors, minoverlap=args.minoverlap, minreadlength=25, Amin='.')     # kvarq/config.py:2-10
    d_genome = scan.DeviceBuffer(g.nbytes)
    d_genome.upload(g)
    d_data = scan.DeviceBuffer(n * rb)
    if L_.kvq_synth_reads_device(d_data.ptr, first, n, L, synth.SEED, d_genome.ptr, g.nbytes):
        raise SystemExit('synthetic generator failed: %s' % (_lib.last_error(),))
    fpos0 = first * rb
    offs = analytic_chunk_offsets(n, rb, L)
    # the analytic cuts must be the real chunker's cuts (checked on a prefix of the device data)
    k = min(len(offs) - 1, 8)
    prefix = d_data.download(int(offs[k]))
    real = scan.chunk_offsets(prefix) if offs[k] < n * rb else None
    if real is not None and k > 1:
        assert list(real[:k]) == list(offs[:k]), 'analytic chunk cuts disagree with the chunker'

    # equal batches of at most --batch-bytes (the library takes 4 GiB - 1 MiB per call), cut at chunk
    # boundaries; the base pointer is aligned down to 16 bytes
    nb = max(1, -(-(n * rb) // args.batch_bytes))
    nch = len(offs) - 1
    cuts = [round(k * nch / nb) for k in range(nb + 1)]
    batches = []
    for k in range(nb):
        i, j = cuts[k], cuts[k + 1]
        if j <= i:
            continue
        base = int(offs[i]) & ~15
        batches.append((base, int(offs[j]) - base, (offs[i:j + 1] - base).copy()))

    table = scan.Table(seqs, **cfg)
    depth = args.pipeline if args.pipeline > 0 else (3 if len(batches) == 1 else 1)
    ctrs = None
    reduce_by = None
    join_checked = None
    scanners = []
    if world > 1:
        # the join of the ranks: libkvarq_hip.so's own RCCL communicator (include/kvarq_hip.h, "several GPUs") --
        # `finish` then sums the counter arrays of all ranks with one all-reduce on the scan's stream.  Every scan in
        # flight has a communicator of its own (collectives of one communicator must not overlap).  Should the
        # library fail to make its communicators, the same sum is taken by torch.distributed (also RCCL), one scan
        # at a time, and the JSON line says so.
        try:
            for _ in range(depth):
                comm = kdist.NativeComm.from_torch(dist, device='cuda' if backend == 'nccl' else None)
                sc = scan.Scanner(table)
                sc.set_comm(comm)
                scanners.append(sc)
            reduce_by = 'libkvarq_hip (RCCL all-reduce inside kvq_scan_finish)'
        except Exception as e:                       # noqa: BLE001
            sys.stderr.write('bench.py: native communicator unavailable (%s): reducing with torch.distributed\n' % e)
            depth = 1
            ctrs = [torch.zeros(table.counters_len, dtype=torch.int64, device='cuda')]
            scanners = [scan.Scanner(table, ctrs[0].data_ptr())]
            reduce_by = 'torch.distributed all_reduce (%s)' % ('RCCL' if backend == 'nccl' else backend)
    else:
        scanners = [scan.Scanner(table) for _ in range(depth)]
    if args.exhaustive:
        for sc in scanners:
            sc.force_exhaustive(True)
    ctr = ctrs[0] if ctrs else None

    # One step = reset + scan of the whole resident text + finish (ordered hits, hit bytes and summed counters on the
    # host).  Steps are independent jobs; `depth` of them are in flight: a step's scan is enqueued, then the step that
    # was enqueued depth - 1 calls earlier is finished.  Every step is finished inside the timed region.
    in_flight = []

    def begin(k):
        sc = scanners[k % depth]
        sc.reset()
        for base, nbytes, co in batches:
            sc.scan_device(d_data.ptr + base, nbytes, co, fpos_base=fpos0 + base)
        if depth > 1 and not args.no_finish_begin:
            sc.finish_begin()                    # (the job's tail behind its own kernels, not behind the host's wait two jobs later)
        in_flight.append(sc)

    def end():
        sc = in_flight.pop(0)
        r = sc.finish(hits=False, stats=False)          # hits, hit bytes and counters are on the host (C arrays); no Python tuples or dicts here
        if ctr is not None:
            kdist.reduce_counters(ctr, dist)                            # hit/coverage arrays over xGMI (one sum all-reduce)
            torch.cuda.current_stream().synchronize()                   # the next step zeroes ctr on the scan's own stream
        # every step is checked, not only the last one (the counters are a view of the scan's host array: read them now)
        want = n * (world if ctr is None else 1)
        got = int(r['counters'][_lib.CTR_RECORDS])
        assert got == want, 'records lost: %d of %d' % (got, want)
        r['records'] = got
        return r

    def run(nsteps):
        """nsteps whole steps; -> their results"""
        out_ = []
        for k in range(nsteps):
            begin(k)
            if len(in_flight) == depth:
                out_.append(end())
        while in_flight:
            out_.append(end())
        return out_

    def sync():
        L_.kvq_device_synchronize()
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    run(max(0, args.preheat))                    # part of the setup: brings the clocks up, not counted as steps
    run(args.warmup)
    sync()
    t0 = time.perf_counter()
    kern_ms = main_ms = 0.0
    launches = 0
    results = run(args.steps)
    r = results[-1]
    assert len(results) == args.steps and len(set((x['records'], x['n_hits']) for x in results)) == 1, 'steps disagree'
    for x in results:
        kern_ms += x['kernel_ms']
        main_ms += x['main_kernel_ms']
        launches += x['main_kernel_launches']
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        if ctr is not None:
            total_records = int(ctr[_lib.CTR_RECORDS].item())
            total_hits = int(ctr[_lib.CTR_HITS].item())
            assert int(r['counters'][_lib.CTR_RECORDS]) == n, 'records lost: %d of %d' % (int(r['counters'][_lib.CTR_RECORDS]), n)
            # the fallback reducer against an independent sum: every rank's own hit and record count of the last step, summed as plain numbers
            own = torch.tensor([int(r['n_hits']), int(r['counters'][_lib.CTR_RECORDS])], dtype=torch.int64, device='cuda')
            dist.all_reduce(own, op=dist.ReduceOp.SUM)
            same = int(own[0].item()) == total_hits and int(own[1].item()) == total_records
            join_checked = ('hits and records of the last step in the summed counter array == sum over the ranks of their own counts (fallback reducer: %s)' % reduce_by
                            if same else 'MISMATCH: the summed counter array disagrees with the ranks\' own counts')
        else:                                        # (the library's host copy of the counters is the sum over all ranks)
            total_records = int(r['counters'][_lib.CTR_RECORDS])
            total_hits = int(r['counters'][_lib.CTR_HITS])
            # the library's sum against an independent one (outside the timed region): every rank's OWN counters of
            # the last step, summed by torch.distributed, must equal what kvq_scan_finish left on this rank
            # (a failure of this CHECK is reported in the line, not raised: the measurement above stands either way;
            # every rank takes part in its collectives whatever it finds)
            import numpy as np
            import ctypes as C
            try:
                own = np.zeros(table.counters_len, dtype=np.int64)
                sc_last = scanners[(args.steps - 1) % depth]
                rc_own = L_.kvq_memcpy_d2h(own.ctypes.data_as(C.c_void_p), L_.kvq_scan_device_counters_own(sc_last.h), own.nbytes)
                chk = torch.from_numpy(own).to('cuda')
                longest = chk[_lib.CTR_LONGEST].clone()
                dist.all_reduce(chk, op=dist.ReduceOp.SUM); dist.all_reduce(longest, op=dist.ReduceOp.MAX)
                chk[_lib.CTR_LONGEST] = longest
                mine = torch.from_numpy(np.array(r['counters'], dtype=np.int64))
                same = rc_own == 0 and bool((chk.cpu() == mine).all())
                join_checked = ('nseqhits, coverage and every other counter of the last step == torch.distributed sum of the ranks\' own counters'
                                if same else 'MISMATCH: the library\'s sum over the ranks differs from torch.distributed\'s sum of the ranks\' own counters')
            except Exception as e:                   # noqa: BLE001
                join_checked = 'check not completed: %s' % e
            if join_checked.startswith(('MISMATCH', 'check not')):
                sys.stderr.write('bench.py: rank %d: %s\n' % (rank, join_checked))
        assert total_records == world * n, 'records lost: %d of %d' % (total_records, world * n)
    else:
        total_records = int(r['counters'][_lib.CTR_RECORDS])
        total_hits = r['n_hits']
        assert total_records == n, 'records lost: %d of %d' % (total_records, n)

    # SURVEY 8(d) defines t_kernel as the sum of ALL device kernels of a scan: HIP events of the library around everything a
    # step enqueues (table upload, tile tables, scan, validation, redo chain, fold, ordering, copies) only mean that when
    # one step runs at a time, so five more steps are run strictly one after the other behind the timed region
    all_ms = None
    if args.steps > 0:
        saved_depth, depth = depth, 1
        try:
            extra = run(5)
            all_ms = sum(x['kernel_ms'] for x in extra) / len(extra)
        finally:
            depth = saved_depth
        sync()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    value = world * n * args.steps / dt
    main_avg_ms = main_ms / max(1, launches)
    bytes_per_launch = n * rb / max(1, launches // args.steps) if launches else 0
    achieved = (n * rb * args.steps) / (main_ms * 1e-3) / 1e9 if main_ms > 0 else 0.0
    out = {
        'metric': 'fastq_reads_per_sec_scanned', 'value': value, 'unit': 'reads/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'strong' if args.total_reads else 'weak', 'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
        'config': {'workload': '%s%d x %d bp synthetic FastQ per GPU (%d B/record, resident in HBM) vs %s table '
                               '(%d templates, both strands = %d sequences, %d bases); e=%d, minoverlap=%d, minreadlength=25, Amin=\'.\''
                               % ('%d x %d bp in all, read-sharded over %d GPUs = ' % (args.total_reads, L, world) if args.total_reads else '',
                                  n, L, rb, args.table, len(plus), len(seqs), sum(map(len, seqs)), args.maxerrors, args.minoverlap),
                   'baseline_config': (None if (args.maxerrors, args.minoverlap) != (2, 25) else
                                       'configs[3]' if args.total_reads == 40_000_000 and world == 8 and L == 150 and args.table == 'MTBC' else
                                       'configs[2]' if n == 10_000_000 and L == 150 and args.table == 'MTBC' and args.table_scale == 1 else
                                       'configs[1]' if n == 1_000_000 and L == 150 and args.table == 'MTBC' and args.table_scale == 1 else None),
                   'total_reads': world * n,
                   'reads_per_gpu': n, 'readlen': L, 'table': args.table, 'table_scale': args.table_scale,
                   'parallelism': 'read-shard x%d, counter arrays summed by %s' % (world, reduce_by) if world > 1 else 'single GPU',
                   'kernel_path': 'exhaustive' if args.exhaustive or not any(table.seeded) else
                                  'seed-filter k=%d (%d of %d sequences)' % (table.seed_k, sum(table.seeded), table.nseq),
                   'join_checked': join_checked,
                   'hits_per_step': total_hits, 'records_per_step': total_records,
                   'preheat_steps': max(0, args.preheat), 'steps_in_flight': depth},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': None,
                     'kernel': 'kvq_scan_bp' if any(table.seeded) and not args.exhaustive else 'kvq_match_all',
                     'launches_per_step': launches // max(1, args.steps), 'avg_launch_ms': main_avg_ms,
                     'algorithmic_bytes_per_launch': bytes_per_launch,
                     # (HIP events around everything a step enqueues; with several steps in flight that span also holds the
                     # wait for the scan kernel in front, so it is only quoted for --pipeline 1)
                     # every kernel and copy of a step (SURVEY 8d's t_kernel), from five steps run one at a time behind the timed region
                     'all_kernels_ms_per_step': all_ms,
                     'frac_all_kernels': (n * rb / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if all_ms else None},
    }
    # HBM bytes per launch of the dominant kernel from the PMC passes of this same command
    # (profiles/, collected with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE as MI355X_MICROARCH.md prescribes)
    # -- the newest round's file that matches this workload
    import glob
    import re
    cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'round*_pmc_traffic.json')),
                   key=lambda q: int(re.search(r'round(\d+)_', os.path.basename(q)).group(1)), reverse=True)
    for path in cands:
        try:
            with open(path) as f:
                pmc = json.load(f)
        except (IOError, ValueError):
            continue
        if (pmc.get('reads_per_gpu') == n and pmc.get('kernel') == out['roofline']['kernel'] and
                pmc.get('launches_per_step') == out['roofline']['launches_per_step']):
            # counters measured on OTHER code say nothing about this run: the file carries the hash of the
            # kernel sources it was measured on, and only the same sources may quote it
            if pmc.get('source_sha256') == source_sha256():
                out['roofline']['traffic'] = pmc['hbm_bytes_per_launch']
                out['roofline']['traffic_source'] = 'profiles/' + os.path.basename(path)
            else:
                out['roofline']['traffic_note'] = 'profiles/%s was measured on other kernel sources (%s): not quoted' % (
                    os.path.basename(path), str(pmc.get('source_sha256'))[:12])
            break
    if world == 1 and not args.no_end_to_end and not args.exhaustive:
        out['end_to_end'] = end_to_end(d_data, n, rb, seqs, cfg)
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(g, seqs, cfg, L, rb, args.cpu_seconds)
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    print(json.dumps(out))
    sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one process per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, as torch.distributed.run would), relay what they print, return the first
    non-zero exit code.  Runs BEFORE torch or libkvarq_hip are imported: the launcher never initialises the GPU (a process that
    has must not be replaced or forked around on this pool), and it starts fresh interpreters rather than forking."""
    import socket
    import subprocess
    loaded = [m for m in ('torch', 'kvarq_amd._lib') if m in sys.modules]
    assert not loaded, 'the launcher must not have touched the GPU: %s' % loaded
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    sys.stderr.write('bench.py: launched %d ranks (pids %s) before any HIP call; rendezvous 127.0.0.1:%d\n' % (n, ' '.join(str(p.pid) for p in procs), port))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)      # (rank 0 prints the JSON line)
    reader.start()
    rc = 0
    deadline = time.time() + 3600
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code and not rc:
                rc = code
                for q in pending:               # a rank has failed: the others would wait for it in a collective for ever
                    q.terminate()
        if pending:
            if time.time() > deadline:
                for q in pending:
                    q.kill()
                rc = rc or 124
            time.sleep(0.05)
    reader.join(10)
    lines = b''.join(out0).decode(errors='replace').splitlines()
    for line in lines:                               # rank 0's JSON line to stdout, anything else it printed to stderr
        (sys.stdout if line.startswith('{') else sys.stderr).write(line + '\n')
    sys.stdout.flush()
    return rc


def source_sha256():
    """hash of the sources libkvarq_hip.so is built from (kvarq_amd/csrc, include/): what a PMC measurement under
    profiles/ is tied to"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, 'kvarq_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'kvarq_amd', 'csrc', '*.inc')) +
                       glob.glob(os.path.join(ROOT, 'kvarq_amd', 'csrc', '*.h')) + glob.glob(os.path.join(ROOT, 'include', '*.h'))):
        h.update(os.path.basename(path).encode())
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def end_to_end(d_data, n, rb, seqs, cfg, nthreads=8):
    """the same records as a page-cache-warm FastQ file through engine.findseqs to the Python result (hits, hit
    bytes, stats): host read + PCIe + kernels + result -- SURVEY 8d's "end-to-end" next to the HBM-resident `value`
    (never `value` itself).  Outside the timed region; the best of three calls."""
    from kvarq_amd import engine
    path = '/tmp/kvarq_bench_e2e.fastq'
    try:
        d_data.download().tofile(path)
        engine.config(nthreads=nthreads, **cfg)
        best, hits = None, 0
        for _ in range(3):
            t0 = time.perf_counter()
            r = engine.findseqs(path, seqs)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
            hits = len(r['hits'])
            assert r['stats']['records_parsed'] == n, 'end to end: records lost'
        return {'reads_per_s': n / best, 'GB_per_s': n * rb / best / 1e9, 'file_bytes': n * rb, 'nthreads': nthreads, 'hits': hits,
                'seconds': best, 'what': 'engine.findseqs on a page-cache-warm plain FastQ file of the same records: pread into pinned buffers, PCIe, kernels, ordered hits and stats as Python objects'}
    except Exception as e:              # noqa: BLE001 -- a full /tmp must not cost the bench line
        return {'error': str(e)[:200]}
    finally:
        if os.path.exists(path):
            os.remove(path)


def cpu_baseline(g, seqs, cfg, L, rb, seconds):
    """the CPU engine on this host's cores, on a bounded sample of the same
    workload: the reference's own engine (oracle/_ref) where it loads, else this
    repo's C restatement of it (oracle/kvarq_oracle.c); both with one worker per core"""
    from kvarq_amd import synth
    from oracle import oracle as O
    # the GPU box gives one GPU a share of 16 host cores: that is the baseline's thread count
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))
    probe = synth.reads(g, 0, 4000 * min(cores, 16), L)
    t0 = time.perf_counter()
    O.scan_memory(probe, seqs, nthreads=cores, **cfg)
    rate = (probe.nbytes // rb) / (time.perf_counter() - t0)
    n = int(max(20000, min(3_000_000, rate * seconds)))
    data = synth.reads(g, 0, n, L)
    t0 = time.perf_counter()
    r = O.scan_memory(data, seqs, nthreads=cores, **cfg)
    port = n / (time.perf_counter() - t0)
    out = {'value': port, 'unit': 'reads/s', 'cores': cores, 'kind': 'port',
           'sample': 'first %d records of the same synthetic stream (%d hits), %d worker threads' % (n, len(r['hits']), cores)}
    try:
        if O.ref_engine() is not None:
            path = '/tmp/kvarq_bench_sample.fastq'
            with open(path, 'wb') as f:
                f.write(data.tobytes())
            t0 = time.perf_counter()
            rr = O.ref_findseqs(path, seqs, nthreads=cores, **cfg)
            ref = n / (time.perf_counter() - t0)
            os.remove(path)
            assert len(rr['hits']) == len(r['hits'])
            out.update({'value': ref, 'kind': 'reference', 'port_value': port})
    except Exception as e:          # the reference build is optional on the GPU box
        out['reference_error'] = str(e)[:200]
    return out


if __name__ == '__main__':
    main()
