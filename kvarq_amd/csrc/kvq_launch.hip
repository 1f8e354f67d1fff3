// kvarq_amd/csrc/kvq_launch.hip -- enqueueing the fused seed-filter scan of one batch (whichever kernel walks the text:
// kvq_scan_bp)
#include "kvq_host.h"

// ---------------------------------------------------------------------------
// launch (both scan kernels)
// ---------------------------------------------------------------------------

int kvq_seeded_launch(kvq_scan *s, const KvqParams &P, const uint8_t *d_data, int64_t nbytes,
                      const uint32_t *d_chunk_off, int64_t nchunks, int64_t fpos_base, uint32_t max_chunk_bytes)
{
    (void)nbytes; (void)max_chunk_bytes;
    SeedIndex *ix = s->t->index;
    // tiles per chunk; the tables live in the scan's pool so that nothing here waits for the GPU
    const std::vector<int64_t> &co = s->cur_chunk_off;
    if (s->tile_bytes == 0) {
        // first batch of a device-resident scan: look at the head of the text once (the choice is kept
        // across kvq_scan_reset; host batches are sized on the host, kvq_scan_host_async)
        const size_t n = (size_t)std::min<int64_t>(nbytes, 128 << 10);
        std::vector<uint8_t> head(n);
        KVQ_HIP(hipStreamSynchronize(s->stream));
        KVQ_HIP(hipMemcpy(head.data(), d_data, n, hipMemcpyDeviceToHost));
        s->tile_bytes = kvq_tile_for_text(head.data(), n, &s->rec_bytes);
    }
    const uint32_t TILE = s->tile_bytes;
    uint64_t nt = 0;
    for (int64_t c = 0; c < nchunks; c++) {
        const uint32_t a = (uint32_t)co[c], b = (uint32_t)co[c + 1];
        nt += b > a ? (uint32_t)(((uint64_t)b - (a & ~15u) + TILE - 1) / TILE) : 0u;
    }
    s->cur_ntiles = 0;                  // (run_batch leaves the redo chain out behind a launch that scanned no tile: its counts are only zeroed by kvq_expand_tiles)
    if (nt == 0) {                      // (nothing to scan: the pair of events of the batch is recorded all the same)
        if (!s->ev_main.empty()) { KVQ_HIP(hipEventRecord(s->ev_main.back().first, s->stream)); KVQ_HIP(hipEventRecord(s->ev_main.back().second, s->stream)); }
        return KVQ_OK;
    }
    // workgroups per launch: what the CUs hold at once (four per CU)
    static const uint32_t grid_env = (uint32_t)(getenv("KVQ_GRID") ? atoi(getenv("KVQ_GRID")) : 0);
    // (a process that keeps several scan objects is taken to overlap their work -- the next job's scan with the
    // last one's fold, ordering and copy: one CU in FOUR then keeps a workgroup slot free, so that those small
    // kernels run beside the persistent workgroups of the scan instead of behind them.  Round 2 left one in eight: the
    // kernels that stand in FRONT of the next scan (reset, table upload, kvq_expand_tiles) then queue for the same 32 slots
    // behind the last step's ordering kernels and the next scan starts 140 us late (tools/r3_gap.py: 137 -> 20 us idle
    // between scans; tools/r3_grid_sweep.sh: 992 workgroups 1.26 ms per step, 976 1.18, 960 1.18, 944 1.20, 928 1.20 --
    // the scan kernel itself pays 1.5 % for the 64 slots))
    const uint32_t cus = s->cus ? s->cus : (s->cus = kvq_device_cu_count());      // (of the device this scan object lives on: a process may use several)
    const uint32_t per_cu = 4u;
    const uint32_t grid_full = cus * per_cu, grid_shared = grid_full - cus / 4u;
    // (round 4: ... and only while another scan of the process is actually on the device as this one is enqueued -- jobs in flight.  A caller
    // that runs its jobs one at a time gets every slot: the free ones bought it nothing, the scan alone is 2 % faster with all of them)
    const uint32_t grid_cap = grid_env ? grid_env : (kvq_live_scans() > 1 && kvq_chain_busy(s)) ? grid_shared : grid_full;
    if (s->pool.used + ((size_t)nchunks + 1) * 4 + (size_t)nt * 24 + 24576 + KVQ_SKIP_CAP * sizeof(KvqSkippedTile) > s->pool.cap) {       // run_batch made the room
        kvq_set_error(KVQ_ERR_RUNTIME, "batch tables outgrew their reservation"); return KVQ_ERR_RUNTIME;
    }
    // first tile of every chunk, then the parameter block: one copy
    const size_t first_b = (((size_t)nchunks + 1) * 4 + 255) & ~(size_t)255;
    const size_t ctr_b = 256 + 4 * BP_SHARDS * BP_SHARD_STRIDE;                  // the tile counters
    const size_t first_at = s->pool.take(first_b + sizeof(BpArgs) + ctr_b);      // ... and the tile counters behind it
    const size_t chunk_at = s->pool.take((size_t)nt * 16), report_at = s->pool.take((size_t)nt * 8), skip_at = s->pool.take(KVQ_SKIP_CAP * sizeof(KvqSkippedTile));      // (report: a word per tile, then a word per tile for the records a skipping tile kept)
    s->cur_skip_at = skip_at; s->cur_first_at = first_at; s->cur_ntiles = (uint32_t)nt;
    uint32_t *first = reinterpret_cast<uint32_t *>(s->pool.h + first_at);
    uint64_t acc = 0;
    for (int64_t c = 0; c < nchunks; c++) {
        const uint32_t a = (uint32_t)co[c], b = (uint32_t)co[c + 1];
        first[c] = (uint32_t)acc;
        acc += b > a ? (uint32_t)(((uint64_t)b - (a & ~15u) + TILE - 1) / TILE) : 0u;
    }
    first[nchunks] = (uint32_t)acc;
    static const uint32_t dbg = (uint32_t)(getenv("KVQ_DBG") ? atoi(getenv("KVQ_DBG")) : 0);
    const uint32_t grid_seeded = (uint32_t)std::min<uint64_t>(nt, grid_cap);
    uint32_t *d_first = reinterpret_cast<uint32_t *>(s->pool.d + first_at);
    const BpArgs *d_args = reinterpret_cast<const BpArgs *>(s->pool.d + first_at + first_b);
    uint32_t *d_tchunk = reinterpret_cast<uint32_t *>(s->pool.d + chunk_at);
    uint32_t *d_report = reinterpret_cast<uint32_t *>(s->pool.d + report_at);
    const size_t ctr_at = first_at + first_b + ((sizeof(BpArgs) + 127) & ~(size_t)127);
    unsigned int *d_tile_ctr = reinterpret_cast<unsigned int *>(s->pool.d + ctr_at);
    {
        // the argument block of the scan kernel
        BpArgs a;
        memset(&a, 0, sizeof(a));
        a.P = P; a.X = ix->dev; a.data = d_data; a.fpos_base = fpos_base;
        a.tiles = reinterpret_cast<const uint4 *>(d_tchunk); a.tile_report = d_report; a.tile_ctr = d_tile_ctr;
        static const uint32_t stagger = (uint32_t)(getenv("KVQ_STAGGER") ? atoi(getenv("KVQ_STAGGER")) : 0);
        a.ntiles = (uint32_t)nt; a.tile_bytes = TILE; a.dbg = dbg; a.pad_ = stagger;
        a.redo = s->d_redo.p; a.redo_cap = KVQ_REDO_CAP - KVQ_LONG_CAP; a.fail = s->cur_fail;
        static const bool surv_off = getenv("KVQ_SURVIVORS") && getenv("KVQ_SURVIVORS")[0] == '0';      // (tests: 0 = every work item is verified where it is found)
        a.surv = surv_off ? nullptr : s->d_surv.p; a.surv_cap = s->surv_cap;
        memcpy(s->pool.h + first_at + first_b, &a, sizeof(a));
    }
    {
        unsigned int *hc = reinterpret_cast<unsigned int *>(s->pool.h + ctr_at);
        memset(hc, 0, 4 * BP_SHARDS * BP_SHARD_STRIDE);
        for (uint32_t sh = 0; sh < BP_SHARDS; sh++) hc[sh * BP_SHARD_STRIDE] = bp_shard_begin(sh, (uint32_t)nt);
    }
    // chunk offsets (run_batch put them right in front), first tiles, arguments, tile counters: one transfer
    KVQ_HIP(hipMemcpyAsync(s->pool.d + s->cur_co_at, s->pool.h + s->cur_co_at, ctr_at + 4 * BP_SHARDS * BP_SHARD_STRIDE - s->cur_co_at,
                           hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(kvq_expand_tiles, dim3((uint32_t)((nchunks + 255) / 256)), dim3(256), 0, s->stream, (uint32_t)nchunks, d_chunk_off, d_first, reinterpret_cast<uint4 *>(d_tchunk),
                       s->d_redo.p ? KvqRedo(s->d_redo.p).count : (unsigned int *)nullptr, s->d_surv.p ? KvqSurvivors(s->d_surv.p).count : (unsigned int *)nullptr);

    // the scan kernel alone between the pair of events its time is read from (bench.py's roofline figure; rocprofv3 --kernel-trace
    // gives the same duration): the table upload, kvq_expand_tiles and kvq_validate_tiles stand outside
    bool pipelined = false;
    { const int rcw = kvq_chain_wait(s, &pipelined); if (rcw) return rcw; }     // behind the last scan kernel of this process (any scan object's)
    const bool timed = !s->ev_main.empty();
    if (timed) KVQ_HIP(hipEventRecord(s->ev_main.back().first, s->stream));
    typedef void (*BpKernel)(const BpArgs *);
    const int si = ix->stride == 8 ? 2 : ix->stride == 4 ? 1 : 0, st = (dbg & 16u) ? 3 : 0;
    {
        // the lane group of a read: four lanes, fixed at compile time, when that is the widest power of two
        // that gives every read of a full tile its own lanes (records of 100 to 250 bases); otherwise the
        // kernel that works the width out per tile
        // (with a KVQ_DBG switch set: the instantiations that honour them -- the general kernel and the four-lane one)
        static const BpKernel kernels_diag[6] = { kvq_scan_bp<2, -1, false, true>, kvq_scan_bp<4, -1, false, true>, kvq_scan_bp<8, -1, false, true>,
                                                  kvq_scan_bp<2, 2, false, true>, kvq_scan_bp<4, 2, false, true>, kvq_scan_bp<8, 2, false, true> };
        static const BpKernel kernels_bp[18] = { kvq_scan_bp<2, -1, false>, kvq_scan_bp<4, -1, false>, kvq_scan_bp<8, -1, false>,
                                                 kvq_scan_bp<2, -1, true>, kvq_scan_bp<4, -1, true>, kvq_scan_bp<8, -1, true>,
                                                 kvq_scan_bp<2, 2, false>, kvq_scan_bp<4, 2, false>, kvq_scan_bp<8, 2, false>,
                                                 kvq_scan_bp<2, 2, true>, kvq_scan_bp<4, 2, true>, kvq_scan_bp<8, 2, true>,
                                                 kvq_scan_bp<2, 3, false>, kvq_scan_bp<4, 3, false>, kvq_scan_bp<8, 3, false>,
                                                 kvq_scan_bp<2, 1, false>, kvq_scan_bp<4, 1, false>, kvq_scan_bp<8, 1, false> };
        static const int lg_env = getenv("KVQ_LG") ? atoi(getenv("KVQ_LG")) : -2;
        int lg = -1;
        if (s->rec_bytes >= 40u) {
            const uint32_t n_full = TILE / s->rec_bytes + 1u;                    // records a full tile can own
            if (n_full <= 128u && n_full > 64u) lg = 2;
            else if (n_full <= 64u && n_full > 32u) lg = 3;                      // (eight lanes a read: records of 250 to 550 bases)
            else if (n_full <= 256u && n_full > 128u) lg = 1;                    // (two lanes a read: records of 50 to 125 bases)
        }
        if (lg_env >= -1) lg = lg_env >= 1 && lg_env <= 3 ? lg_env : -1;
        if ((lg == 3 || lg == 1) && st) lg = -1;                                // (no instrumented build of those)
        // (seeds shorter than 8 -- (maxerrors + 1) * 8 above the shortest accepted overlap: kvq_seed_k -- have the general and the four-lane kernel)
        static const BpKernel kernels_k5[6] = { kvq_scan_bp<2, -1, false, true, 5>, kvq_scan_bp<4, -1, false, true, 5>, kvq_scan_bp<8, -1, false, true, 5>,
                                                kvq_scan_bp<2, 2, false, true, 5>, kvq_scan_bp<4, 2, false, true, 5>, kvq_scan_bp<8, 2, false, true, 5> };
        static const BpKernel kernels_k6[6] = { kvq_scan_bp<2, -1, false, true, 6>, kvq_scan_bp<4, -1, false, true, 6>, kvq_scan_bp<8, -1, false, true, 6>,
                                                kvq_scan_bp<2, 2, false, true, 6>, kvq_scan_bp<4, 2, false, true, 6>, kvq_scan_bp<8, 2, false, true, 6> };
        static const BpKernel kernels_k7[6] = { kvq_scan_bp<2, -1, false, true, 7>, kvq_scan_bp<4, -1, false, true, 7>, kvq_scan_bp<8, -1, false, true, 7>,
                                                kvq_scan_bp<2, 2, false, true, 7>, kvq_scan_bp<4, 2, false, true, 7>, kvq_scan_bp<8, 2, false, true, 7> };
        // (8-base seeds on a table that is dense in the code space: the draining kernels)
        static const BpKernel kernels_dense[6] = { kvq_scan_bp<2, -1, false, true, 8, true>, kvq_scan_bp<4, -1, false, true, 8, true>, kvq_scan_bp<8, -1, false, true, 8, true>,
                                                   kvq_scan_bp<2, 2, false, true, 8, true>, kvq_scan_bp<4, 2, false, true, 8, true>, kvq_scan_bp<8, 2, false, true, 8, true> };
        static const int dense_env = getenv("KVQ_DENSE") ? atoi(getenv("KVQ_DENSE")) : -1;      // (tests: 1 forces the draining kernels, 0 the halving ones)
        const bool dense = ix->k == 8 && !st && (dense_env >= 0 ? dense_env != 0 : ix->dense);
        const bool diag = ((dbg & ~16u) && !st) || ix->k != 8 || dense;
        if (diag && lg != 2) lg = -1;
        const BpKernel *const dk = ix->k == 5 ? kernels_k5 : ix->k == 6 ? kernels_k6 : ix->k == 7 ? kernels_k7 : dense ? kernels_dense : kernels_diag;
        const BpKernel kern = diag ? dk[(lg == 2 ? 3 : 0) + si] : kernels_bp[lg == 3 ? 12 + si : lg == 1 ? 15 + si : (lg == 2 ? 6 : 0) + si + st];
        hipLaunchKernelGGL(kern, dim3(grid_seeded), dim3(ST_THREADS), 0, s->stream, d_args);
    }
    if (timed) KVQ_HIP(hipEventRecord(s->ev_main.back().second, s->stream));
    // What passed the scan kernel's 16-base test (0.02 work items per read of the bench workload), byte-exact: a lane each, count on the device.
    // One job at a time: 256 workgroups of 1024, and the next scan of the process waits for them (31 us).  Several jobs in flight (the
    // scan in front of this one was still running when this one was enqueued): the next scan starts right behind this one and the
    // survivors are verified BESIDE it by 64 small workgroups that fit the slots it leaves free -- 120 us of their own, none of the
    // step's (step 1.086 -> 1.063 ms; with more or larger workgroups beside the scan the step gets longer, tools/r4_sv_mode.sh).
    static const int sv_env = getenv("KVQ_SV_MODE") ? atoi(getenv("KVQ_SV_MODE")) : -1;      // (tests: 0 always behind the scan, 1 always beside the next)
    const bool beside = sv_env >= 0 ? sv_env != 0 : pipelined;
    if (beside) { const int rcp = kvq_chain_publish(s); if (rcp) return rcp; }
    static const int sv_grid = getenv("KVQ_SV_GRID") ? atoi(getenv("KVQ_SV_GRID")) : 64;      // (experiments: workgroups of the kernel beside a scan)
    if (s->d_surv.p)
        hipLaunchKernelGGL(kvq_verify_survivors, dim3(beside ? sv_grid : 256), dim3(beside ? 256 : 1024), 0, s->stream, P, d_data, fpos_base, (const void *)s->d_surv.p,
                           (const unsigned int *)s->cur_fail, ix->k, ix->stride, ix->pitch);
    if (!beside) { const int rcp = kvq_chain_publish(s); if (rcp) return rcp; }            // (kvq_validate_tiles and what follows run beside the next scan)
    if (!(dbg & 64u))          // (diagnostic 64 scans the wrong text on purpose: nothing to validate)
    hipLaunchKernelGGL(kvq_validate_tiles, dim3((uint32_t)((nchunks + 255) / 256)), dim3(256), 0, s->stream, (uint32_t)nchunks,
                       d_first, d_report, s->cur_fail, reinterpret_cast<KvqSkippedTile *>(s->pool.d + skip_at), d_chunk_off, TILE);
    KVQ_HIP(hipGetLastError());
    if (getenv("KVQ_DBG_REPORT")) {
        // diagnostic: replay kvq_validate_tiles on the host and name the tiles it rejects
        KVQ_HIP(hipStreamSynchronize(s->stream));
        std::vector<uint32_t> rep((size_t)nt);
        KVQ_HIP(hipMemcpy(rep.data(), d_report, (size_t)nt * 4, hipMemcpyDeviceToHost));
        for (int64_t c = 0; c < nchunks; c++) {
            uint32_t seen = 0, total = 0;
            for (uint32_t g = first[c]; g < first[c + 1]; g++) total += rep[g] & 0xFFFFu;
            for (uint32_t g = first[c]; g < first[c + 1]; g++) {
                if (kvq_tile_report_bad(rep[g], g == first[c], seen, total))
                    fprintf(stderr, "tile %u (chunk %lld [%lld, %lld), tile %u of it): report %08x n_owned %u jn %u seen %u of %u\n",
                            g, (long long)c, (long long)co[c], (long long)co[c + 1], g - first[c], rep[g], rep[g] & 0xFFFFu, (rep[g] >> 16) & 0xFFu, seen, total);
                seen += rep[g] & 0xFFFFu;
            }
        }
    }
    return KVQ_OK;
}
