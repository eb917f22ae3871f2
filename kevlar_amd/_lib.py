"""ctypes binding of libkvsketch_hip.so (include/kvsketch.h).

There is no CPU fallback: if the shared object is missing, or no MI355X is visible when a
sketch is first touched, the caller gets an exception -- never a silently different path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBNAME = 'libkvsketch_hip.so'
# KV_LIB_PATH: another build of the same library (the A/B variants scratch/ab_build.py makes); same ABI, same checks
LIBPATH = os.environ.get('KV_LIB_PATH') or os.path.join(_HERE, LIBNAME)

KV_OK = 0
KV_ERR_ARG, KV_ERR_IO, KV_ERR_TYPE, KV_ERR_HIP, KV_ERR_NOTIMPL, KV_ERR_CAPACITY = -1, -2, -3, -4, -5, -6
KV_MAX_TABLES = 16
KV_BAND_NONE, KV_BAND_RANGE, KV_BAND_REFQUIRK = 0, 1, 2

u8p = ctypes.POINTER(ctypes.c_uint8)
u32p = ctypes.POINTER(ctypes.c_uint32)
u64p = ctypes.POINTER(ctypes.c_uint64)
vp = ctypes.c_void_p
vpp = ctypes.POINTER(ctypes.c_void_p)
i32 = ctypes.c_int
u32 = ctypes.c_uint32
u64 = ctypes.c_uint64
cstr = ctypes.c_char_p


class SketchInfo(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int32), ('ksize', ctypes.c_int32), ('ntables', ctypes.c_int32),
                ('reserved', ctypes.c_int32), ('sizes', ctypes.c_uint64 * KV_MAX_TABLES),
                ('n_occupied', ctypes.c_uint64), ('n_unique', ctypes.c_uint64),
                ('bytes_device', ctypes.c_uint64)]


# every symbol include/kvsketch.h declares: name -> (restype, argtypes)
SIGNATURES = {
    'kv_last_error': (cstr, []),
    'kv_version': (cstr, []),
    'kv_device_count': (i32, [ctypes.POINTER(i32)]),
    'kv_set_device': (i32, [i32]),
    'kv_thread_device_get': (i32, [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    'kv_set_stream': (i32, [vp]),
    'kv_synchronize': (i32, []),
    'kv_stream_create': (i32, [vpp]),
    'kv_stream_destroy': (i32, [vp]),
    'kv_table_cache_trim': (i32, []),
    'kv_scratch_trim': (i32, []),
    'kv_unique_release': (i32, []),
    'kv_hits_lazy': (i32, [i32]),
    'kv_knobs_describe': (i32, [i32, ctypes.c_char_p, u64]),
    'kv_knob_get': (i32, [ctypes.c_char_p, ctypes.c_char_p, u64]),
    'kv_prof_enable': (i32, [i32]),
    'kv_prof_reset': (i32, []),
    'kv_prof_get': (i32, [cstr, ctypes.POINTER(ctypes.c_double), u64p]),
    'kv_prof_names': (i32, [ctypes.c_char_p, ctypes.c_size_t]),
    'kv_primes_below': (i32, [ctypes.c_double, i32, u64p, ctypes.POINTER(i32)]),
    'kv_hash_kmer': (i32, [i32, cstr, i32, u64p]),
    'kv_reverse_hash': (i32, [i32, u64, i32, ctypes.c_char_p]),
    'kv_band_bounds': (i32, [i32, i32, u64p, u64p]),
    'kv_sketch_create': (i32, [i32, i32, i32, u64p, vpp]),
    'kv_sketch_destroy': (i32, [vp]),
    'kv_sketch_load': (i32, [cstr, i32, vpp]),
    'kv_sketch_save': (i32, [vp, cstr]),
    'kv_sketch_info_get': (i32, [vp, ctypes.POINTER(SketchInfo)]),
    'kv_sketch_table_read': (i32, [vp, i32, u8p, u64]),
    'kv_sketch_table_write': (i32, [vp, i32, u8p, u64]),
    'kv_sketch_table_devptr': (i32, [vp, i32, vpp, u64p]),
    'kv_sketch_clear': (i32, [vp]),
    'kv_sketch_scan_hint': (i32, [vp, i32]),
    'kv_reads_create': (i32, [cstr, u64p, u64, vpp]),
    'kv_reads_create_packed': (i32, [u32p, u64, u32, vpp]),
    'kv_reads_generate': (i32, [u64, u64, i32, u64, u64, u32, ctypes.c_double, vpp]),
    'kv_reads_words_read': (i32, [vp, u64, u64, u32p]),
    'kv_reads_destroy': (i32, [vp]),
    'kv_fastx_open': (i32, [cstr, vpp]),
    'kv_fastx_next': (i32, [vp, u64, i32, vpp, u64p]),
    'kv_fastx_batch_text': (i32, [vp, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(u64p), ctypes.POINTER(ctypes.c_void_p),
                                  ctypes.POINTER(u64p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(u64p),
                                  ctypes.POINTER(u8p)]),
    'kv_fastx_num_reads': (i32, [vp, u64p]),
    'kv_fastx_from_cache': (i32, [vp, ctypes.POINTER(ctypes.c_int)]),
    'kv_format_augmented': (i32, [u32p, u32p, u8p, u64, i32, i32, u64p, vp, u64p, vp, u64p, vp, u64p, u8p, ctypes.POINTER(vp), u64p, u64p]),
    'kv_text_free': (i32, [vp]),
    'kv_augfastx_load': (i32, [cstr, vpp]),
    'kv_augfastx_info': (i32, [vp, u64p, u64p, ctypes.POINTER(i32), ctypes.POINTER(i32), u64p]),
    'kv_augfastx_view': (i32, [vp] + [vpp] * 13),
    'kv_augfastx_free': (i32, [vp]),
    'kv_reads_flag_other_bytes': (i32, [vp, vp, u64, vp]),
    'kv_canonical_read_hashes': (i32, [vp, vp, vp, u64, vp, vp, vp]),
    'kv_canonical_reads_equal': (i32, [vp, vp, vp, vp, u64, vp, vp]),
    'kv_format_records': (i32, [u64, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u64, vp, vp, vpp, u64p]),
    'kv_format_records_fd': (i32, [u64, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u64, vp, vp, i32, i32, u64p]),
    'kv_fastx_on_device': (i32, [vp, ctypes.POINTER(ctypes.c_int)]),
    'kv_fastx_fetch': (i32, [vp, u64p, u64]),
    'kv_fastx_record_text': (i32, [vp, u64, ctypes.c_char_p, ctypes.c_char_p]),
    'kv_fastx_close': (i32, [vp]),
    'kv_reads_count': (i32, [vp, u64p, u64p]),
    'kv_reads_num_kmers': (i32, [vp, i32, u64p]),
    'kv_consume': (i32, [vp, vp, i32, i32, vp, i32, i32, u64p]),
    'kv_unique_exact': (i32, [vp, vpp, i32, i32, i32, vp, i32, i32, u64p]),
    'kv_unique_new': (i32, [vp, vp, i32, i32, vp, i32, i32, u64p]),
    'kv_abundance_distribution': (i32, [vp, vp, vpp, i32, u64p]),
    'kv_hash_kmers': (i32, [i32, cstr, i32, u64, u64p]),
    'kv_hash_positions': (i32, [vp, i32, i32, u32p, u32p, u64, u64p]),
    'kv_get_hashes': (i32, [vp, u64p, u64, u8p]),
    'kv_add_hashes': (i32, [vp, u64p, u64, u8p]),
    'kv_novel_scan': (i32, [vpp, i32, vpp, i32, vp, u64, i32, i32, i32, i32, i32, i32, vp, u64, vpp]),
    'kv_hits_count': (i32, [vp, u64p, u64p]),
    'kv_hits_fetch': (i32, [vp, u32p, u32p, u8p, u64, u32p, u64]),
    'kv_hits_view': (i32, [vp, ctypes.POINTER(u32p), ctypes.POINTER(u32p), ctypes.POINTER(u8p), ctypes.POINTER(u32p)]),
    'kv_hits_shadow': (i32, [vp, ctypes.POINTER(u32p), ctypes.POINTER(u32p), u64p]),
    'kv_hits_destroy': (i32, [vp]),
    'kv_route_hashes': (i32, [vp, i32, i32, i32, u64, i32, vp, u64, u64p]),
    'kv_consume_hashes': (i32, [vp, vp, u64, ctypes.c_uint32, u64p]),
    'kv_bgzf_text_size': (i32, [vp, u64, u64p, u64p]),
    'kv_bgzf_inflate_host': (i32, [vp, u64, vp, u64, ctypes.POINTER(ctypes.c_double)]),
    'kv_gunzip_host': (i32, [vp, u64, vp, u64, u64, u64p, u64p, ctypes.POINTER(ctypes.c_double)]),
    'kv_route_distinct': (i32, [vp, i32, i32, i32, vp, u64, u64p]),
    'kv_consume_hashes_weighted': (i32, [vp, vp, u64, u64p]),
    'kv_pairs_pack': (i32, [vp, u64p, i32, vp, u64, u64p]),
    'kv_pairs_unpack': (i32, [vp, u64p, i32, vp, u64, u64p, u64p]),
    'kv_novel_scan_hashes': (i32, [vpp, i32, vpp, i32, vp, u64, i32, i32, vp, vp, u64, u64p]),
    'kv_novel_scan_distinct': (i32, [vpp, i32, vpp, i32, vp, u64, i32, i32, vp, vp, u64, u64p]),
    'kv_novel_scan_set': (i32, [vp, i32, i32, i32, vp, vp, u64, vpp]),
    'kv_hits_from_tagged': (i32, [vp, vp, u64, u64, i32, vpp]),
    'kv_argsort_u64': (i32, [vp, u64, vp]),
    'kv_argsort_rows': (i32, [vp, u64, ctypes.c_uint32, vp]),
    'kv_mex_plan_make': (i32, [i32, i32, u64, u32, i32, vp]),
    'kv_mex_plan_short': (i32, [vp]),
    'kv_mex_emit': (i32, [vp, vp, u64, vp, vp]),
    'kv_mex_pack': (i32, [vp, vp, vp, vp, u64p]),
    'kv_mex_emit_pack': (i32, [vp, vp, u64, vp, vp, vp, u64, u64p, ctypes.POINTER(ctypes.c_int)]),
    'kv_mex_route': (i32, [vp, i32, vp, vp, i32, i32, i32, vp, u64, u64p, u64p]),
    'kv_mex_scan_set': (i32, [i32, i32, i32, vp, vp, u64, vp, vp, u64, u64p]),
    'kv_reads_flags': (i32, [vp, vp]),
    'kv_readgraph_components': (i32, [vp, i32, u32p, u32p, u64, u32p, u32, u32, u32, u32p, u64p]),
}


class MexPlan(ctypes.Structure):
    """kv_mex_plan of include/kvsketch.h"""
    _fields_ = [('ksize', ctypes.c_int32), ('ndest', ctypes.c_int32), ('C1', u32), ('F2', u32), ('fbits', u32), ('nwg1', u32),
                ('cap1', u32), ('recw', u32), ('m', u32), ('read_len', u32), ('seg_words', u64), ('cnt_entries', u64),
                ('n_kmers_global', u64), ('n_reads_global', u64), ('c_lo', u32 * 17), ('flags', u32)]


class KvError(RuntimeError):
    """HIP / library failure that has no counterpart among the reference's exceptions."""

    def __init__(self, code, message):
        super(KvError, self).__init__('[kvsketch {}] {}'.format(code, message))
        self.code = code


_lib = None
_device_ready = False


def load():
    """Load libkvsketch_hip.so.  Raises ImportError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise ImportError(
            '{} not found: the HIP extension has not been built (run '
            '`python -c "import __graft_entry__ as g; g.build_product()"`). kevlar_amd has no CPU '
            'fallback.'.format(LIBPATH))
    lib = ctypes.CDLL(LIBPATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().kv_last_error().decode('utf-8', 'replace')


class KvCapacityError(ValueError):
    """KV_ERR_CAPACITY: a buffer the call was given (or sized for itself) did not hold the result; callers with another way to the
    same result take it (a ValueError to everybody else, as before)"""


class KvArgError(ValueError):
    """KV_ERR_ARG: the call was given arguments it does not take -- a mistake of the caller, not of the input.  The sharded step still
    turns it into an agreed fallback (no rank may leave between collectives) but counts it as UNEXPECTED (shardrun.ShardedRun)."""


def check(code):
    """Map a C return code onto the reference's exception types."""
    if code == KV_OK:
        return
    msg = last_error()
    if code == KV_ERR_CAPACITY:
        raise KvCapacityError(msg)
    if code == KV_ERR_ARG:
        raise KvArgError(msg)
    if code == KV_ERR_NOTIMPL:
        raise ValueError(msg)
    if code == KV_ERR_IO:
        raise OSError(msg)
    if code == KV_ERR_TYPE:
        raise ValueError(msg)
    raise KvError(code, msg)


def knob(name, default=None):
    """An environment switch of the wrapper, through the library's one registry (kevlar_amd/csrc/kv_knobs.h): the value if `name` is
    set and its class is honoured right now (tuning switches need KV_TUNING=1), else `default`.  An unregistered name raises."""
    buf = ctypes.create_string_buffer(256)
    rc = load().kv_knob_get(name.encode(), buf, len(buf))
    if rc < 0:
        check(rc)
    return buf.value.decode() if rc == 1 else default


def knobs_active():
    """'NAME=value ...' of every registered switch that is set ('ignored:NAME=value' when its class is not honoured)"""
    buf = ctypes.create_string_buffer(1 << 14)
    check(load().kv_knobs_describe(0, buf, len(buf)))
    return buf.value.decode()


def device_visible():
    """is there a GPU to hand the big sorts of the host stages to?  (never raises: host logic runs without one)"""
    if _device_ready:
        return True
    try:
        n = ctypes.c_int(0)
        return load().kv_device_count(ctypes.byref(n)) == 0 and n.value >= 1
    except OSError:
        return False


def require_device():
    """Make sure a GPU is there before the first table is allocated; fail loudly if not."""
    global _device_ready
    if _device_ready:
        return
    lib = load()
    n = ctypes.c_int(0)
    check(lib.kv_device_count(ctypes.byref(n)))
    if n.value < 1:
        raise KvError(KV_ERR_HIP, 'no HIP device visible: kevlar_amd runs its sketches on an '
                                  'MI355X and has no CPU fallback')
    dev = int(os.environ.get('LOCAL_RANK', '0')) % n.value
    check(lib.kv_set_device(dev))
    _device_ready = True
