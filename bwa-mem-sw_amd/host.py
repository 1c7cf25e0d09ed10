"""Host-side mirror of the reference's operator interface, over the C ABI (ctypes).

Names follow the reference's domain: a *task batch* of seeds goes in, a *result batch*
comes out (batch_manager.v / tbb.v / rbb.v); PARAMS = global words G0/G1, TASK = header
H0..H7, RESULT = record R0..R4 (sw_pe_array_proc_element.v:807-933, :1662-1665).
numpy structured dtypes below are byte-for-byte the C structs of include/bwa_sw_mi355.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("BSW_LIB_PATH") or os.path.join(_HERE, "libbwasw_mi355.so")   # BSW_LIB_PATH: an experimental build (tools only)

PARAMS = np.dtype([("mat", "i1", (25,)), ("_pad", "i1", (3,)), ("o_del", "<i4"), ("e_del", "<i4"),
                   ("o_ins", "<i4"), ("e_ins", "<i4"), ("w", "<i4"), ("pen_clip5", "<i4"),
                   ("pen_clip3", "<i4"), ("zdrop", "<i4"), ("max_band_try", "<i4"), ("variant", "<i4")])
TASK = np.dtype([("lquery", "<u8"), ("ltarget", "<u8"), ("rquery", "<u8"), ("rtarget", "<u8"),
                 ("lqlen", "<i4"), ("ltlen", "<i4"), ("rqlen", "<i4"), ("rtlen", "<i4"),
                 ("h0", "<i4"), ("init_score", "<i4"), ("qbeg", "<i4"), ("tag", "<u4"),
                 ("wlim_l", "<i4"), ("wlim_r", "<i4")])
EXT = np.dtype([("score", "<i4"), ("qle", "<i4"), ("tle", "<i4"), ("gtle", "<i4"), ("gscore", "<i4"),
                ("max_off", "<i4"), ("aw", "<i4"), ("cells", "<u4")])
RESULT = np.dtype([("tag", "<u4"), ("qb", "<i4"), ("qe", "<i4"), ("rb", "<i4"), ("re", "<i4"),
                   ("score", "<i4"), ("truesc", "<i4"), ("w", "<i4"), ("left", EXT), ("right", EXT)])
EXT_TASK = np.dtype([("query", "<u8"), ("target", "<u8"), ("qlen", "<i4"), ("tlen", "<i4"),
                     ("w", "<i4"), ("end_bonus", "<i4"), ("h0", "<i4"), ("_pad", "<i4")])
SYNTH = np.dtype([("seed", "<u8"), ("read_len", "<i4"), ("seed_len_min", "<i4"), ("seed_len_max", "<i4"),
                  ("seed_at_start", "<i4"), ("sub_rate", "<f8"), ("indel_rate", "<f8"), ("n_rate", "<f8"),
                  ("junk_frac", "<f8"), ("a", "<i4"), ("w", "<i4"), ("o", "<i4"), ("e", "<i4")])
SEED = np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4")])
ALNREG = np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("score", "<i4"), ("truesc", "<i4"),
                   ("w", "<i4"), ("_pad", "<i4")])
GTASK = np.dtype([("query", "<u8"), ("target", "<u8"), ("qlen", "<i4"), ("tlen", "<i4"), ("w", "<i4"), ("_pad", "<i4")])
GRESULT = np.dtype([("score", "<i4"), ("n_cigar", "<i4")])
ATASK = np.dtype([("query", "<u8"), ("target", "<u8"), ("qlen", "<i4"), ("tlen", "<i4"), ("xtra", "<i4"), ("_pad", "<i4")])
KSWR = np.dtype([("score", "<i4"), ("te", "<i4"), ("qe", "<i4"), ("score2", "<i4"), ("te2", "<i4"), ("tb", "<i4"), ("qb", "<i4")])
KSW_XBYTE, KSW_XSTOP, KSW_XSUBO, KSW_XSTART = 0x10000, 0x20000, 0x40000, 0x80000
REF_TASK = np.dtype([("query", "<u8"), ("l_query", "<i4"), ("init_score", "<i4"), ("seed", SEED),
                     ("rmax0", "<i8"), ("rmax1", "<i8"), ("tag", "<u4"), ("_pad", "<u4")])
MAX_DEVICES = 16
CONFIG = np.dtype([("device", "<i4"), ("kernel", "<i4"), ("streams", "<i4"), ("pack_threads", "<i4"),
                   ("chunk_tasks", "<u8"), ("n_devices", "<i4"), ("devices", "<i4", (MAX_DEVICES,)),
                   ("timeout_ms", "<i4"), ("result_format", "<i4"), ("pin_threads", "<i4")])
STATS = np.dtype([("slot_cpu_ns", "<u8"), ("helper_cpu_ns", "<u8"), ("seeds", "<u8"), ("chunks", "<u8"), ("submits", "<u8"),
                  ("h2d_bytes", "<u8"), ("d2h_bytes", "<u8"), ("slot_threads", "<u8")])
MAX_INFLIGHT = 4
PAIR = np.dtype([("tag", "<u4"), ("qb", "<i4"), ("qe", "<i4"), ("rb", "<i4"), ("re", "<i4"),
                 ("score", "<i4"), ("truesc", "<i4"), ("w", "<i4")])       # the RTL's 5-word record: the first 32 bytes of RESULT
RESULT_FULL, RESULT_PAIR = 0, 1
assert PARAMS.itemsize == 68 and TASK.itemsize == 72 and EXT.itemsize == 32 and RESULT.itemsize == 96
assert ATASK.itemsize == 32 and KSWR.itemsize == 28
assert EXT_TASK.itemsize == 40 and SYNTH.itemsize == 72 and CONFIG.itemsize == 104 and PAIR.itemsize == 32 and REF_TASK.itemsize == 56

REFBATCH_IN_WORDS, REFBATCH_OUT_WORDS, REFBATCH_MAX_TASKS = 65536, 4096, 819
KERNEL_AUTO, KERNEL_WAVE, KERNEL_LANE = 0, 1, 2
LANE_AUTO_MIN = 40000          # seeds of the 150 bp single bin (131-base sides) from which BSW_KERNEL_AUTO uses the lane bins: LANE_WORK_MIN = 5 M query bases per launched side (bsw_internal.h)
LANE_WORK_MIN, GROUP_WORK_MIN = 5_000_000, 1_500_000   # per launched side: lane kernels / the group kernel (bsw_lane2g_kernel) from this many query bases
VARIANT_H, VARIANT_M = 0, 1

ERRORS = {0: "BSW_OK", -1: "BSW_E_NODEVICE", -2: "BSW_E_INVAL", -3: "BSW_E_LIMIT", -4: "BSW_E_HIP",
          -5: "BSW_E_NOMEM", -6: "BSW_E_BUSY"}


class BswError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("%s (%d) %s" % (ERRORS.get(code, "?"), code, msg))
        self.code = code


def lib_path():
    return _LIB


def build_library(force=False):
    """Compile csrc/ for gfx950 into libbwasw_mi355.so (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(_LIB):
        subprocess.check_call(["make", "-j6", "-C", os.path.join(_HERE, "csrc")], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    """The C-ABI library.  Raises if it is missing: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise RuntimeError("libbwasw_mi355.so is not built (run __graft_entry__.build()); "
                               "this package has no CPU/PyTorch fallback")
        L = C.CDLL(_LIB)
        vp, sz = C.c_void_p, C.c_size_t
        sig = {
            "bsw_default_params": (None, [vp]), "bsw_default_config": (None, [vp]),
            "bsw_device_count": (C.c_int, []),
            "bsw_create": (C.c_int, [vp, C.POINTER(vp)]), "bsw_destroy": (None, [vp]),
            "bsw_create_sized": (C.c_int, [vp, sz, C.POINTER(vp)]), "bsw_abi_version": (C.c_int, []),
            "bsw_chain_timeouts": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
            "bsw_device_placement": (C.c_int, [vp, C.c_int, C.c_char_p, sz, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
            "bsw_last_error": (C.c_char_p, [vp]),
            "bsw_submit": (C.c_int, [vp, vp, vp, sz, vp]), "bsw_wait": (C.c_int, [vp]),
            "bsw_submit_packed": (C.c_int, [vp, vp, vp, sz, vp]),
            "bsw_submit_t": (C.c_int, [vp, vp, vp, sz, vp, C.POINTER(C.c_uint64)]),
            "bsw_submit_packed_t": (C.c_int, [vp, vp, vp, sz, vp, C.POINTER(C.c_uint64)]),
            "bsw_submit_ref_t": (C.c_int, [vp, vp, vp, vp, sz, vp, C.POINTER(C.c_uint64)]),
            "bsw_wait_ticket": (C.c_int, [vp, C.c_uint64]), "bsw_test": (C.c_int, [vp, C.c_uint64]),
            "bsw_inflight": (C.c_int, [vp]), "bsw_host_stats": (C.c_int, [vp, vp, sz]),
            "bsw_upload_packed": (C.c_int, [vp, vp, vp, sz, C.POINTER(vp)]),
            "bsw_pack_tasks": (C.c_int64, [vp, sz, vp, sz, vp]), "bsw_pack_tasks_bound": (sz, [vp, sz]),
            "bsw_extend_batch": (C.c_int, [vp, vp, vp, sz, vp]),
            "bsw_upload": (C.c_int, [vp, vp, vp, sz, C.POINTER(vp)]),
            "bsw_run": (C.c_int, [vp, vp]), "bsw_sync": (C.c_int, [vp]),
            "bsw_upload_raw": (C.c_int, [vp, vp, vp, sz, C.POINTER(vp)]), "bsw_run_staged": (C.c_int, [vp, vp]),
            "bsw_run_history2": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]),
            "bsw_effective_timeout_ms": (C.c_int, [vp]),
            "bsw_download": (C.c_int, [vp, vp, vp]),
            "bsw_batch_info": (C.c_int, [vp] + [C.POINTER(C.c_uint64)] * 4),
            "bsw_last_run_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
            "bsw_free_batch": (None, [vp, vp]),
            "bsw_run_history": (C.c_int, [vp, C.POINTER(C.c_float), C.c_int]),
            "bsw_refbatch_encode": (C.c_int, [vp, vp, sz, vp]),
            "bsw_refbatch_decode": (C.c_int, [vp, vp, vp, sz, vp, sz]),
            "bsw_refbatch_encode_results": (C.c_int, [vp, sz, vp]),
            "bsw_refbatch_decode_results": (C.c_int, [vp, sz, vp]),
            "bsw_refbatch_run": (C.c_int, [vp, vp, vp, C.c_int, C.c_int]),
            "bsw_refbatch_submit": (C.c_int, [vp, vp, vp]),
            "bsw_refbatch_wait": (C.c_int, [vp, C.c_int, C.c_int]),
            "bsw_host_alloc": (vp, [sz]), "bsw_host_free": (None, [vp]),
            "bsw_host_register": (C.c_int, [vp, sz]), "bsw_host_unregister": (C.c_int, [vp]),
            "bsw_batch_order": (C.c_int, [vp, vp, vp, vp]),
            "bsw_scalar_stats": (None, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
            "bsw_global_batch": (C.c_int, [vp, vp, vp, sz, C.c_int, vp, vp]),
            "bsw_align_batch": (C.c_int, [vp, vp, vp, sz, vp]),
            "ksw_global2": (C.c_int, [C.c_int, vp, C.c_int, vp, C.c_int, vp] + [C.c_int] * 5 + [vp, vp]),
            "ksw_global": (C.c_int, [C.c_int, vp, C.c_int, vp, C.c_int, vp] + [C.c_int] * 3 + [vp, vp]),
            "bsw_ref_upload": (C.c_int, [vp, vp, C.c_int64, C.POINTER(vp)]),
            "bsw_ref_free": (None, [vp, vp]),
            "bsw_upload_ref": (C.c_int, [vp, vp, vp, vp, sz, C.POINTER(vp)]),
            "bsw_extend_ref": (C.c_int, [vp, vp, vp, vp, sz, vp]),
            "bsw_submit_ref": (C.c_int, [vp, vp, vp, vp, sz, vp]),
            "bsw_plan_batch": (C.c_int64, [vp, vp, sz, C.c_int, C.c_int, vp, vp]),
            "bsw_pack_bases": (C.c_int, [vp, C.c_int, vp]),
            "bsw_cal_max_gap": (C.c_int, [vp, C.c_int]),
            "bsw_chain_window": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int64, vp]),
            "bsw_seed_scratch_bytes": (sz, [vp, C.c_int64]),
            "bsw_seed_to_task": (C.c_int, [vp, vp, C.c_int, vp, C.c_int64, C.c_int64, vp, vp, sz, C.c_uint32, vp]),
            "bsw_result_to_alnreg": (C.c_int, [vp, vp, vp]),
            "bsw_pac_get_seq": (C.c_int64, [C.c_int64, vp, C.c_int64, C.c_int64, vp]),
            "bsw_synth_generate": (C.c_int64, [vp, sz, vp, vp, sz]),
            "bsw_synth_arena_bound": (sz, [vp, sz]),
            "bsw_synth_ref_generate": (C.c_int64, [vp, vp, C.c_int64, vp, sz, vp, vp, sz]),
            "bsw_set_default_variant": (None, [C.c_int]),
            "ksw_extend2": (C.c_int, [C.c_int, vp, C.c_int, vp, C.c_int, vp] + [C.c_int] * 8 + [vp] * 5),
            "ksw_extend": (C.c_int, [C.c_int, vp, C.c_int, vp, C.c_int, vp] + [C.c_int] * 6 + [vp] * 5),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


EXPORTS = ["ksw_global2", "ksw_global", "bsw_global_batch", "bsw_align_batch", "ksw_align2", "ksw_align", "ksw_extend2", "ksw_extend", "bsw_set_default_variant", "bsw_scalar_stats", "bsw_host_alloc", "bsw_host_free",
           "bsw_host_register", "bsw_host_unregister", "bsw_batch_order", "bsw_refbatch_submit", "bsw_refbatch_wait", "bsw_default_params", "bsw_default_config",
           "bsw_device_count", "bsw_create", "bsw_create_sized", "bsw_abi_version", "bsw_chain_timeouts", "bsw_device_placement", "bsw_destroy", "bsw_last_error", "bsw_submit", "bsw_wait",
           "bsw_submit_packed", "bsw_upload_packed", "bsw_pack_tasks", "bsw_pack_tasks_bound",
           "bsw_submit_t", "bsw_submit_packed_t", "bsw_submit_ref_t", "bsw_wait_ticket", "bsw_test", "bsw_inflight", "bsw_host_stats",
           "bsw_upload_raw", "bsw_run_staged", "bsw_run_history2", "bsw_effective_timeout_ms",
           "bsw_extend_batch", "bsw_upload", "bsw_run", "bsw_sync", "bsw_download", "bsw_batch_info",
           "bsw_last_run_ms", "bsw_run_history", "bsw_free_batch", "bsw_refbatch_encode", "bsw_refbatch_decode",
           "bsw_refbatch_encode_results", "bsw_refbatch_decode_results", "bsw_refbatch_run",
           "bsw_ref_upload", "bsw_ref_free", "bsw_upload_ref", "bsw_extend_ref", "bsw_submit_ref",
           "bsw_plan_batch", "bsw_pack_bases", "bsw_cal_max_gap", "bsw_chain_window", "bsw_seed_scratch_bytes", "bsw_seed_to_task",
           "bsw_result_to_alnreg", "bsw_pac_get_seq", "bsw_synth_generate", "bsw_synth_arena_bound", "bsw_synth_ref_generate"]


def default_params(**over):
    p = np.zeros(1, dtype=PARAMS)
    lib().bsw_default_params(p.ctypes.data)
    for k, v in over.items():
        p[k] = v
    return p


def pack_tasks(tasks, arena=None):
    """Byte-per-base tasks -> (tasks whose pointers address 4-bit packed words, the uint64 arena holding them).  `arena`: a
    uint64 view of registered memory (HostArena.view(np.uint64, ...)) to pack into, or None for a fresh numpy array."""
    need = int(lib().bsw_pack_tasks_bound(tasks.ctypes.data, len(tasks)))
    if arena is None:
        arena = np.zeros(need // 8 + 1, dtype=np.uint64)
    assert arena.dtype == np.uint64 and arena.nbytes >= need
    out = np.zeros(len(tasks), dtype=TASK)
    used = int(lib().bsw_pack_tasks(tasks.ctypes.data, len(tasks), arena.ctypes.data, arena.nbytes, out.ctypes.data))
    if used < 0:
        raise RuntimeError("bsw_pack_tasks failed (%d)" % used)
    return out, arena


def bwa_matrix(a=1, b=4, n=-1):
    m = np.full((5, 5), -b, dtype=np.int8)
    for i in range(4):
        m[i, i] = a
    m[4, :] = n
    m[:, 4] = n
    return m.reshape(25)


def synth_arena_bound(n, **spec):
    s = np.zeros(1, dtype=SYNTH)
    d = dict(seed=1, read_len=150, seed_len_min=19, seed_len_max=19, seed_at_start=1, sub_rate=0.01,
             indel_rate=0.001, n_rate=0.0, junk_frac=0.0, a=1, w=100, o=6, e=1)
    d.update(spec)
    for k, v in d.items():
        s[k] = v
    return int(lib().bsw_synth_arena_bound(s.ctypes.data, n))


def synth_tasks(n, arena=None, **spec):
    """Generate n synthetic seeds.  Returns (tasks, arena); keep arena alive.
    arena: optional uint8 buffer to generate into (e.g. HostArena(...).u8 for DMA-direct submits)."""
    s = np.zeros(1, dtype=SYNTH)
    d = dict(seed=1, read_len=150, seed_len_min=19, seed_len_max=19, seed_at_start=1, sub_rate=0.01,
             indel_rate=0.001, n_rate=0.0, junk_frac=0.0, a=1, w=100, o=6, e=1)
    d.update(spec)
    for k, v in d.items():
        s[k] = v
    L = lib()
    bound = L.bsw_synth_arena_bound(s.ctypes.data, n)
    if arena is None:
        arena = np.zeros(bound, dtype=np.uint8)
    elif arena.size < bound:
        raise ValueError("arena too small: need %d bytes" % bound)
    tasks = np.zeros(n, dtype=TASK)
    used = L.bsw_synth_generate(s.ctypes.data, n, tasks.ctypes.data, arena.ctypes.data, arena.size)
    if used < 0:
        raise BswError(int(used), "bsw_synth_generate")
    return tasks, arena


def synth_ref_tasks(n, l_pac, params, arena=None, **spec):
    """Synthetic genome + reads against it (bsw_synth_ref_generate).  Returns (pac, rtasks, arena)."""
    s = np.zeros(1, dtype=SYNTH)
    d = dict(seed=1, read_len=150, seed_len_min=19, seed_len_max=19, seed_at_start=1, sub_rate=0.01,
             indel_rate=0.001, n_rate=0.0, junk_frac=0.0, a=1, w=100, o=6, e=1)
    d.update(spec)
    for k, v in d.items():
        s[k] = v
    need = int(d["read_len"]) * n
    if arena is None:
        arena = np.zeros(need + 64, dtype=np.uint8)
    elif arena.size < need:
        raise ValueError("arena too small: need %d bytes" % need)
    pac = np.zeros((l_pac + 3) // 4, dtype=np.uint8)
    rt = np.zeros(n, dtype=REF_TASK)
    used = lib().bsw_synth_ref_generate(s.ctypes.data, params.ctypes.data, l_pac, pac.ctypes.data, n, rt.ctypes.data, arena.ctypes.data, arena.size)
    if used < 0:
        raise BswError(int(used), "bsw_synth_ref_generate")
    return pac, rt, arena


def make_tasks(seeds):
    """Build a TASK array from python dicts {lq, lt, rq, rt, h0, [init_score, qbeg, tag]} (sequences =
    iterables of base codes).  Returns (tasks, arena)."""
    total = sum(len(s.get(k, ())) for s in seeds for k in ("lq", "lt", "rq", "rt"))
    arena = np.zeros(total + 8, dtype=np.uint8)
    tasks = np.zeros(len(seeds), dtype=TASK)
    off = 0
    base = arena.ctypes.data
    for i, s in enumerate(seeds):
        for key, pf, lf in (("lq", "lquery", "lqlen"), ("lt", "ltarget", "ltlen"),
                            ("rq", "rquery", "rqlen"), ("rt", "rtarget", "rtlen")):
            a = np.asarray(s.get(key, ()), dtype=np.uint8)
            arena[off:off + len(a)] = a
            tasks[i][pf] = base + off
            tasks[i][lf] = len(a)
            off += len(a)
        tasks[i]["h0"] = s["h0"]
        tasks[i]["init_score"] = s.get("init_score", -1)
        tasks[i]["qbeg"] = s.get("qbeg", len(s.get("lq", ())))
        tasks[i]["tag"] = s.get("tag", i)
    return tasks, arena


class DeviceBatch:
    def __init__(self, ctx, handle, n):
        self.ctx, self.handle, self.n = ctx, handle, n

    def info(self):
        v = [C.c_uint64(0) for _ in range(4)]
        lib().bsw_batch_info(self.handle, *[C.byref(x) for x in v])
        return dict(n_tasks=v[0].value, in_bytes=v[1].value, out_bytes=v[2].value, launches=v[3].value)

    def free(self):
        if self.handle:
            lib().bsw_free_batch(self.ctx.handle, self.handle)
            self.handle = None


class BswContext:
    """One GPU context = one of the reference's PE arrays behind its batch manager."""

    def __init__(self, device=0, kernel=KERNEL_AUTO, streams=4, pack_threads=8, chunk_tasks=0, devices=None,
                 timeout_ms=0, result_format=RESULT_FULL, pin_threads=None):
        cfg = np.zeros(1, dtype=CONFIG)
        lib().bsw_default_config(cfg.ctypes.data)
        cfg["device"], cfg["kernel"], cfg["streams"] = device, kernel, streams
        cfg["pack_threads"], cfg["chunk_tasks"] = pack_threads, chunk_tasks
        if devices is not None:                 # one context over several GPUs: chunk k -> devices[k % len(devices)]
            cfg["n_devices"] = len(devices)
            cfg["devices"][0, :len(devices)] = devices
        if timeout_ms:
            cfg["timeout_ms"] = timeout_ms
        cfg["result_format"] = result_format
        if pin_threads is not None:
            cfg["pin_threads"] = 1 if pin_threads else -1
        self.out_dtype = PAIR if result_format == RESULT_PAIR else RESULT      # what the submit calls hand back
        h = C.c_void_p()
        rc = lib().bsw_create(cfg.ctypes.data, C.byref(h))
        if rc:
            raise BswError(rc, "bsw_create")
        self.handle = h
        self._keep = None
        self.last_ticket = 0

    def _chk(self, rc, what):
        if rc:
            raise BswError(rc, "%s: %s" % (what, lib().bsw_last_error(self.handle).decode()))

    def placement(self, k=0):
        """Where device k of the context sits and whether its slot threads are pinned next to it."""
        buf = C.create_string_buffer(64)
        node, ncpu = C.c_int(-1), C.c_int(0)
        self._chk(lib().bsw_device_placement(self.handle, k, buf, 64, C.byref(node), C.byref(ncpu)), "bsw_device_placement")
        return dict(bdf=buf.value.decode(), numa_node=node.value, pinned_cpus=ncpu.value)

    def chain_timeouts(self):
        """Waits of the launch chain that ended at their deadline (0 unless kernels are being run one at a time)."""
        v = C.c_uint64(0)
        self._chk(lib().bsw_chain_timeouts(self.handle, C.byref(v)), "bsw_chain_timeouts")
        return v.value

    def close(self):
        if self.handle:
            lib().bsw_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # streaming path: host buffers in, host buffers out.  Up to MAX_INFLIGHT submits per context; every submit has a ticket.
    def _submit(self, fn, what, params, tasks, out, *front):
        if out is None:
            out = np.zeros(len(tasks), dtype=self.out_dtype)
        assert out.dtype == self.out_dtype and len(out) >= len(tasks)
        t = C.c_uint64(0)
        self._chk(fn(self.handle, params.ctypes.data, *front, tasks.ctypes.data, len(tasks), out.ctypes.data, C.byref(t)), what)
        if self._keep is None:
            self._keep = {}
        self._keep[t.value] = (params, tasks, out)          # the arrays must outlive the submit
        self.last_ticket = t.value
        return out

    def submit(self, params, tasks, out=None):
        return self._submit(lib().bsw_submit_t, "bsw_submit", params, tasks, out)

    def wait(self):
        """every submit in flight; raises the first failure in submit order"""
        try:
            self._chk(lib().bsw_wait(self.handle), "bsw_wait")
        finally:
            self._keep = None

    def wait_ticket(self, ticket):
        try:
            self._chk(lib().bsw_wait_ticket(self.handle, ticket), "bsw_wait_ticket")
        finally:
            if self._keep:
                self._keep.pop(ticket, None)

    def test(self, ticket):
        """non-blocking: True once the submit is complete (collect it with wait_ticket / wait)"""
        rc = lib().bsw_test(self.handle, ticket)
        if rc < 0:
            self._chk(rc, "bsw_test")
        return bool(rc)

    def inflight(self):
        return int(lib().bsw_inflight(self.handle))

    def host_stats(self):
        """bsw_host_stats as a dict: CPU ns of the slot / gather threads, seeds, chunks, submits, bytes moved"""
        s = np.zeros(1, dtype=STATS)
        self._chk(lib().bsw_host_stats(self.handle, s.ctypes.data, STATS.itemsize), "bsw_host_stats")
        return {k: int(s[k][0]) for k in STATS.names}

    def submit_packed(self, params, ptasks, out=None):
        """bsw_submit_packed: `ptasks` from pack_tasks() (sequence pointers address 4-bit packed words)."""
        return self._submit(lib().bsw_submit_packed_t, "bsw_submit_packed", params, ptasks, out)

    def extend_pairs_packed(self, params, ptasks, out=None):
        out = self.submit_packed(params, ptasks, out)
        self.wait()
        return out[:len(ptasks)]

    def upload_packed(self, params, ptasks):
        h = C.c_void_p()
        self._chk(lib().bsw_upload_packed(self.handle, params.ctypes.data, ptasks.ctypes.data, len(ptasks), C.byref(h)), "bsw_upload_packed")
        return DeviceBatch(self, h, len(ptasks))

    def extend_pairs(self, params, tasks, out=None):
        """Streaming path, synchronous.  Pass a reused `out` array to keep page faults of a fresh one out of timings."""
        out = self.submit(params, tasks, out)
        self.wait()
        return out[:len(tasks)]

    def extend_batch(self, params, etasks):
        out = np.zeros(len(etasks), dtype=EXT)
        self._chk(lib().bsw_extend_batch(self.handle, params.ctypes.data, etasks.ctypes.data, len(etasks), out.ctypes.data), "bsw_extend_batch")
        return out

    # device-resident path
    def upload(self, params, tasks):
        h = C.c_void_p()
        self._chk(lib().bsw_upload(self.handle, params.ctypes.data, tasks.ctypes.data, len(tasks), C.byref(h)), "bsw_upload")
        return DeviceBatch(self, h, len(tasks))

    def upload_raw(self, params, tasks):
        """bsw_upload_raw: the byte-per-base sequences stay in HBM beside the packed form (for run_staged)."""
        h = C.c_void_p()
        self._chk(lib().bsw_upload_raw(self.handle, params.ctypes.data, tasks.ctypes.data, len(tasks), C.byref(h)), "bsw_upload_raw")
        return DeviceBatch(self, h, len(tasks))

    def run(self, batch):
        self._chk(lib().bsw_run(self.handle, batch.handle), "bsw_run")

    def run_staged(self, batch):
        """pack + bin + DP kernels of a batch from upload_raw (the device side of one bsw_submit chunk)."""
        self._chk(lib().bsw_run_staged(self.handle, batch.handle), "bsw_run_staged")

    def run_history2(self, cap=4096):
        """[(total_ms, staging_ms)] of every run since the last call."""
        a, b = (C.c_float * cap)(), (C.c_float * cap)()
        n = lib().bsw_run_history2(self.handle, a, b, cap)
        if n < 0:
            self._chk(n, "bsw_run_history2")
        return [(a[i], b[i]) for i in range(n)]

    def sync(self):
        self._chk(lib().bsw_sync(self.handle), "bsw_sync")

    def last_run_ms(self):
        ms = C.c_float(0)
        self._chk(lib().bsw_last_run_ms(self.handle, C.byref(ms)), "bsw_last_run_ms")
        return ms.value

    def run_history(self, cap=4096):
        buf = (C.c_float * cap)()
        n = lib().bsw_run_history(self.handle, buf, cap)
        if n < 0:
            self._chk(n, "bsw_run_history")
        return [buf[i] for i in range(n)]

    def download(self, batch):
        out = np.zeros(batch.n, dtype=RESULT)
        self._chk(lib().bsw_download(self.handle, batch.handle, out.ctypes.data), "bsw_download")
        return out

    def global_batch(self, params, gtasks, max_cigar=64, want_cigar=True):
        """Batched ksw_global2.  Returns (GRESULT array, cigars uint32[n, max_cigar] or None)."""
        res = np.zeros(len(gtasks), dtype=GRESULT)
        cig = np.zeros((len(gtasks), max_cigar), dtype=np.uint32) if want_cigar else None
        self._chk(lib().bsw_global_batch(self.handle, params.ctypes.data, gtasks.ctypes.data, len(gtasks), max_cigar,
                                         res.ctypes.data, cig.ctypes.data if want_cigar else None), "bsw_global_batch")
        return res, cig

    def align_batch(self, params, atasks):
        """Batched ksw_align2 (bwa's local alignment of mate rescue).  Returns a KSWR array."""
        res = np.zeros(len(atasks), dtype=KSWR)
        self._chk(lib().bsw_align_batch(self.handle, params.ctypes.data, atasks.ctypes.data, len(atasks), res.ctypes.data), "bsw_align_batch")
        return res

    def batch_order(self, batch):
        """Launch order the device-side binning produced (order, seg) — same layout as plan_batch."""
        order = np.zeros(4 * batch.n + 16, dtype=np.uint32)
        seg = np.zeros(PLAN_SEGS + 1, dtype=np.uint32)
        self._chk(lib().bsw_batch_order(self.handle, batch.handle, order.ctypes.data, seg.ctypes.data), "bsw_batch_order")
        return order, seg

    def refbatch_submit(self, in_words, out_words):
        """Queue one 256 KiB task batch; both arrays must stay alive until refbatch_wait()."""
        assert in_words.dtype == np.uint32 and in_words.size == REFBATCH_IN_WORDS and in_words.flags.c_contiguous
        assert out_words.dtype == np.uint32 and out_words.size == REFBATCH_OUT_WORDS and out_words.flags.c_contiguous
        self._chk(lib().bsw_refbatch_submit(self.handle, in_words.ctypes.data, out_words.ctypes.data), "bsw_refbatch_submit")

    def refbatch_wait(self, variant=VARIANT_H, zdrop=0):
        rc = lib().bsw_refbatch_wait(self.handle, variant, zdrop)
        if rc < 0:
            self._chk(rc, "bsw_refbatch_wait")
        return rc

    # seeds against a device-resident 2-bit reference (F3)
    def ref_upload(self, pac, l_pac):
        pac = np.ascontiguousarray(pac, dtype=np.uint8)
        h = C.c_void_p()
        self._chk(lib().bsw_ref_upload(self.handle, pac.ctypes.data, l_pac, C.byref(h)), "bsw_ref_upload")
        return h

    def ref_free(self, ref):
        lib().bsw_ref_free(self.handle, ref)

    def extend_ref(self, params, ref, rtasks):
        out = np.zeros(len(rtasks), dtype=RESULT)
        self._chk(lib().bsw_extend_ref(self.handle, params.ctypes.data, ref, rtasks.ctypes.data, len(rtasks), out.ctypes.data), "bsw_extend_ref")
        return out

    def submit_ref(self, params, ref, rtasks, out=None):
        """Streaming form of extend_ref (finish with wait()): only the reads cross PCIe."""
        return self._submit(lib().bsw_submit_ref_t, "bsw_submit_ref", params, rtasks, out, ref)

    def refbatch_run(self, in_words, variant=VARIANT_H, zdrop=0):
        in_words = np.ascontiguousarray(in_words, dtype=np.uint32)
        assert in_words.size == REFBATCH_IN_WORDS
        out = np.zeros(REFBATCH_OUT_WORDS, dtype=np.uint32)
        rc = lib().bsw_refbatch_run(self.handle, in_words.ctypes.data, out.ctypes.data, variant, zdrop)
        if rc < 0:
            self._chk(rc, "bsw_refbatch_run")
        return out, rc


class HostArena:
    """Pinned (DMA-able) host memory from bsw_host_alloc, viewed as a numpy array.  Sequences and result
    arrays kept in such memory cross PCIe without a host copy (bsw_submit's direct path)."""

    def __init__(self, nbytes):
        self.ptr = lib().bsw_host_alloc(nbytes)
        if not self.ptr:
            raise BswError(-5, "bsw_host_alloc(%d)" % nbytes)
        self.nbytes = nbytes
        self.u8 = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_uint8)), shape=(nbytes,))

    def view(self, dtype, count, offset=0):
        return self.u8[offset:offset + count * np.dtype(dtype).itemsize].view(dtype)

    def free(self):
        if self.ptr:
            self.u8 = None
            lib().bsw_host_free(self.ptr)
            self.ptr = None


def host_register(arr):
    """Pin an existing numpy array for DMA (bsw_host_register); returns the rc."""
    return lib().bsw_host_register(arr.ctypes.data, arr.nbytes)


def host_unregister(arr):
    return lib().bsw_host_unregister(arr.ctypes.data)


def scalar_stats():
    a, b = C.c_uint64(0), C.c_uint64(0)
    lib().bsw_scalar_stats(C.byref(a), C.byref(b))
    return a.value, b.value


def refbatch_encode(params, tasks):
    words = np.zeros(REFBATCH_IN_WORDS, dtype=np.uint32)
    n = lib().bsw_refbatch_encode(params.ctypes.data, tasks.ctypes.data, len(tasks), words.ctypes.data)
    if n < 0:
        raise BswError(n, "bsw_refbatch_encode")
    return words, n


def refbatch_decode(words):
    words = np.ascontiguousarray(words, dtype=np.uint32)
    p = np.zeros(1, dtype=PARAMS)
    tasks = np.zeros(REFBATCH_MAX_TASKS, dtype=TASK)
    seqbuf = np.zeros(REFBATCH_IN_WORDS * 8 + 64, dtype=np.uint8)
    n = lib().bsw_refbatch_decode(words.ctypes.data, p.ctypes.data, tasks.ctypes.data, len(tasks), seqbuf.ctypes.data, seqbuf.size)
    if n < 0:
        raise BswError(n, "bsw_refbatch_decode")
    return p, tasks[:n], seqbuf


def refbatch_encode_results(res):
    words = np.zeros(REFBATCH_OUT_WORDS, dtype=np.uint32)
    n = lib().bsw_refbatch_encode_results(res.ctypes.data, len(res), words.ctypes.data)
    if n < 0:
        raise BswError(n, "bsw_refbatch_encode_results")
    return words


def refbatch_decode_results(words, n):
    words = np.ascontiguousarray(words, dtype=np.uint32)
    res = np.zeros(n, dtype=RESULT)
    rc = lib().bsw_refbatch_decode_results(words.ctypes.data, n, res.ctypes.data)
    if rc < 0:
        raise BswError(rc, "bsw_refbatch_decode_results")
    return res


PLAN_SEGS = 26


def plan_batch(params, tasks, kernel=KERNEL_AUTO, pack_threads=1):
    """Batch manager's launch plan for a task batch (host only; the order is the host replay of the device's
    binning rules).  Returns (order, seg, seq_words)."""
    order = np.zeros(5 * len(tasks) + 32, dtype=np.uint32)
    seg = np.zeros(PLAN_SEGS + 1, dtype=np.uint32)
    w = lib().bsw_plan_batch(params.ctypes.data, tasks.ctypes.data, len(tasks), kernel, pack_threads, order.ctypes.data, seg.ctypes.data)
    if w < 0:
        raise BswError(int(w), "bsw_plan_batch")
    return order, seg, int(w)


def pack_bases(bases):
    """Device sequence format of one sequence (see bsw_pack_bases).  Returns (uint64 words, has_N)."""
    b = np.ascontiguousarray(bases, dtype=np.uint8)
    words = np.zeros((len(b) + 15) // 16, dtype=np.uint64)
    rc = lib().bsw_pack_bases(b.ctypes.data if len(b) else None, len(b), words.ctypes.data if len(words) else None)
    if rc < 0:
        raise BswError(rc, "bsw_pack_bases")
    return words, bool(rc)


def pac_get_seq(pac, l_pac, beg, end):
    """bns_get_seq on a 2-bit packed reference (numpy uint8 array)."""
    dst = np.zeros(max(abs(end - beg), 1), dtype=np.uint8)
    n = lib().bsw_pac_get_seq(l_pac, pac.ctypes.data, beg, end, dst.ctypes.data)
    if n < 0:
        raise BswError(int(n), "bsw_pac_get_seq")
    return dst[:n]


def pack_pac(bases):
    """ACGT codes -> bwa .pac layout (4 bases per byte, first base in the top two bits)."""
    b = np.asarray(bases, dtype=np.uint8)
    pad = (-len(b)) % 4
    b4 = np.concatenate([b, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    return (b4[:, 0] << 6 | b4[:, 1] << 4 | b4[:, 2] << 2 | b4[:, 3]).astype(np.uint8)


def seeds_to_tasks(params, pac, l_pac, reads, seeds):
    """mem_chain2aln glue for single-seed chains: reads = list of base arrays, seeds = SEED array (one per read).
    Returns (tasks, keepalive) with the sequences laid out by bsw_seed_to_task."""
    L = lib()
    n = len(reads)
    tasks = np.zeros(n, dtype=TASK)
    keep = []
    for i in range(n):
        q = np.ascontiguousarray(reads[i], dtype=np.uint8)
        rmax = np.zeros(2, dtype=np.int64)
        rc = L.bsw_chain_window(params.ctypes.data, seeds[i:i + 1].ctypes.data, 1, len(q), l_pac, rmax.ctypes.data)
        if rc:
            raise BswError(rc, "bsw_chain_window")
        rseq = pac_get_seq(pac, l_pac, int(rmax[0]), int(rmax[1]))
        rseq = np.ascontiguousarray(rseq)
        scratch = np.zeros(L.bsw_seed_scratch_bytes(seeds[i:i + 1].ctypes.data, int(rmax[0])) + 8, dtype=np.uint8)
        rc = L.bsw_seed_to_task(params.ctypes.data, seeds[i:i + 1].ctypes.data, len(q), q.ctypes.data, int(rmax[0]), int(rmax[1]),
                                rseq.ctypes.data, scratch.ctypes.data, scratch.size, i, tasks[i:i + 1].ctypes.data)
        if rc:
            raise BswError(rc, "bsw_seed_to_task")
        keep.append((q, rseq, scratch))
    return tasks, keep


def results_to_alnregs(seeds, results):
    out = np.zeros(len(seeds), dtype=ALNREG)
    for i in range(len(seeds)):
        lib().bsw_result_to_alnreg(seeds[i:i + 1].ctypes.data, results[i:i + 1].ctypes.data, out[i:i + 1].ctypes.data)
    return out


def shard_indices(n, world, rank, chunk=65536):
    """Per-read task shard of rank `rank`: task k -> rank (k // chunk) % world.
    Tasks are independent, so there is no exchange step and no data-path collective."""
    idx = np.arange(n, dtype=np.int64)
    return idx[(idx // chunk) % world == rank]


def task_seq(tasks, i, field, lenfield):
    """Read one sequence of task i back as a numpy array (for tests / fixtures)."""
    n = int(tasks[i][lenfield])
    if n == 0:
        return np.zeros(0, dtype=np.uint8)
    return np.ctypeslib.as_array(C.cast(int(tasks[i][field]), C.POINTER(C.c_uint8)), shape=(n,)).copy()
