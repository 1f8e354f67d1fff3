"""ctypes binding of libkvarq_hip.so (include/kvarq_hip.h).  There is no CPU
fallback: if the library is missing, import fails loudly; if no GPU is present,
every compute entry point reports KVQ_ERR_DEVICE."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, 'libkvarq_hip.so')

MAX_READLENGTH = 1024
OK, ERR_FORMAT, ERR_IO, ERR_MEMORY, ERR_RUNTIME, ERR_TYPE, ERR_DEVICE, ERR_RESCAN = range(8)
CTR_RECORDS, CTR_LONGEST, CTR_HITS, CTR_READLENGTHS = 0, 1, 2, 4


class Config(C.Structure):
    _fields_ = [('maxerrors', C.c_int32), ('minoverlap', C.c_int32), ('minreadlength', C.c_int32),
                ('nthreads', C.c_int32), ('Amin', C.c_int8), ('Azero', C.c_int8)]


class LiveStats(C.Structure):
    _fields_ = [('records_parsed', C.c_int64), ('parsed', C.c_int64), ('total', C.c_int64),
                ('rls_longest', C.c_int64), ('nseq', C.c_int32), ('running', C.c_int32),
                ('sigints', C.c_int32), ('stop_requested', C.c_int32)]


i32, i64, u64, vp, cp = C.c_int32, C.c_int64, C.c_uint64, C.c_void_p, C.c_char_p
P = C.POINTER

# name -> (restype, argtypes): every symbol include/kvarq_hip.h declares
PROTOTYPES = {
    'kvq_config_set': (None, [P(Config)]),
    'kvq_config_get': (None, [P(Config)]),
    'kvq_last_error': (i32, [C.c_char_p, C.c_size_t]),
    'kvq_table_create': (vp, [P(cp), P(i32), i32, P(Config)]),
    'kvq_table_destroy': (None, [vp]),
    'kvq_table_nseq': (i32, [vp]),
    'kvq_table_bases': (i64, [vp]),
    'kvq_table_seq_is_seeded': (i32, [vp, i32]),
    'kvq_table_seed_k': (i32, [vp]),
    'kvq_counters_len': (i64, [vp]),
    'kvq_counters_off_nseqhits': (i64, [vp]),
    'kvq_counters_off_nseqbasehits': (i64, [vp]),
    'kvq_counters_off_coverage': (i64, [vp]),
    'kvq_counters_off_mutations': (i64, [vp]),
    'kvq_table_seq_offset': (i64, [vp, i32]),
    'kvq_scan_create': (vp, [vp, vp]),
    'kvq_scan_destroy': (None, [vp]),
    'kvq_chunk_offsets': (i64, [vp, i64, P(i64), i64]),
    'kvq_scan_device': (i32, [vp, vp, i64, P(i64), i64, i64]),
    'kvq_scan_host': (i32, [vp, vp, i64, P(i64), i64, i64]),
    'kvq_scan_host_async': (i32, [vp, vp, i64, P(i64), i64, i64]),
    'kvq_scan_host_copied': (i32, [vp]),
    'kvq_scan_host_drain': (i32, [vp]),
    'kvq_scan_finish': (i32, [vp]),
    'kvq_scan_finish_begin': (i32, [vp]),
    'kvq_scan_n_hits': (i64, [vp]),
    'kvq_scan_hit_seq_nr': (P(i32), [vp]),
    'kvq_scan_hit_file_pos': (P(i64), [vp]),
    'kvq_scan_hit_seq_pos': (P(i32), [vp]),
    'kvq_scan_hit_length': (P(i32), [vp]),
    'kvq_scan_hit_readlength': (P(i32), [vp]),
    'kvq_scan_hitseq_blob': (vp, [vp]),
    'kvq_scan_hitseq_offsets': (P(i64), [vp]),
    'kvq_scan_counters': (P(i64), [vp]),
    'kvq_scan_device_counters': (vp, [vp]),
    'kvq_scan_device_counters_own': (vp, [vp]),
    'kvq_scan_parsed': (i64, [vp]),
    'kvq_scan_total': (i64, [vp]),
    'kvq_scan_kernel_ms': (C.c_double, [vp]),
    'kvq_scan_main_kernel_ms': (C.c_double, [vp]),
    'kvq_scan_gap_ms': (C.c_double, [vp, vp]),
    'kvq_scan_main_kernel_launches': (i64, [vp]),
    'kvq_scan_reset': (i32, [vp]),
    'kvq_scan_path': (i32, [vp]),
    'kvq_scan_force_exhaustive': (None, [vp, i32]),
    'kvq_comm_unique_id': (i32, [vp]),
    'kvq_comm_create': (vp, [i32, i32, vp]),
    'kvq_comm_destroy': (None, [vp]),
    'kvq_comm_nranks': (i32, [vp]),
    'kvq_comm_rank': (i32, [vp]),
    'kvq_scan_set_comm': (i32, [vp, vp]),
    'kvq_scan_gather_hits': (i32, [vp, vp]),
    'kvq_comm_allreduce_counters': (i32, [vp, vp, i64, vp]),
    'kvq_comm_create_local': (vp, [i32, i32, C.c_uint64]),
    'kvq_gather_plan': (i32, [i32, vp, vp, vp, vp]),
    'kvq_gather_host': (i32, [i32, vp, vp, vp]),
    'kvq_result_layout_words': (None, [C.c_uint64, C.c_uint64, vp]),
    'kvq_findseqs': (vp, [P(cp), i32, P(cp), P(i32), i32]),
    'kvq_findseqs_free': (None, [vp]),
    'kvq_host_chunk_plan': (i64, [P(cp), i32, P(i64), P(i64), i64, P(i64), P(i64), i64]),
    'kvq_poll_stats': (None, [P(LiveStats), P(i64), P(i64), P(i64), i32]),
    'kvq_request_stop': (None, []),
    'kvq_count_sigint': (None, []),
    'kvq_sigint_counter_install': (C.c_int, []),
    'kvq_sigint_counter_remove': (None, []),
    'kvq_device_count': (i32, []),
    'kvq_set_device': (i32, [i32]),
    'kvq_device_alloc': (vp, [i64]),
    'kvq_device_free': (None, [vp]),
    'kvq_memcpy_h2d': (i32, [vp, vp, i64]),
    'kvq_memcpy_d2h': (i32, [vp, vp, i64]),
    'kvq_memset_d': (i32, [vp, i32, i64]),
    'kvq_device_synchronize': (i32, []),
    'kvq_release_cached': (None, []),
    'kvq_synth_reads_device': (i32, [vp, i64, i64, i32, u64, vp, i64]),
    'kvq_synth_reads_host': (None, [vp, i64, i64, i32, u64, vp, i64]),
    'kvq_synth_genome_host': (None, [vp, i64, u64]),
    'kvq_version': (cp, []),
}

_lib = None


def build():
    """compile the library in-tree (hipcc, gfx950)"""
    import subprocess
    subprocess.check_call(['make', '-s', '-C', os.path.join(HERE, 'csrc')])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(PATH):
            raise ImportError('%s is missing: build it with `make -C kvarq_amd/csrc` '
                              '(there is no CPU fallback for the scan)' % PATH)
        L = C.CDLL(PATH)
        for name, (res, args) in PROTOTYPES.items():
            f = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def last_error():
    buf = C.create_string_buffer(1024)
    code = lib().kvq_last_error(buf, 1024)
    return code, buf.value.decode('latin-1')
