"""ctypes binding of libplume_hip.so (include/plume_hip.h) — the only way this package computes anything.

There is deliberately no fallback: if the shared library is missing, or no gfx950 GPU is visible, constructing
an Engine raises PlumeHipError.  torch is used only as plumbing (device buffers / streams) by the *_device
methods; the host-pointer methods need numpy only.
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_u8p = C.POINTER(C.c_uint8)
_u64p = C.POINTER(C.c_uint64)


# the library's shipped defaults (plume_capi.hip, struct plume_ctx) -- tests that turn a knob restore these
DEFAULT_CHUNK = 1 << 20
DEFAULT_HOST_PIECE = 1 << 19
DEFAULT_HOST_FIRST_PIECE = 1 << 16
DEFAULT_HOST_TAIL_PIECE = 1 << 16
DEFAULT_SUB_BATCHES = 1


class PlumeHipError(RuntimeError):
    pass


def library_path() -> Path:
    """the in-tree build; PLUME_HIP_LIB selects another build of the same library (A/B tuning runs only)"""
    alt = os.environ.get("PLUME_HIP_LIB")
    return Path(alt).resolve() if alt else _HERE / "libplume_hip.so"


_lib = None


def _load():
    # The runtime's hardware-queue pool (4 per priority level by default): streams that share a queue run their kernels one after the other, which is what defeats
    # plume_set_in_flight for callers with two streams (include/plume_hip.h).  Effective only if set before the process's first HIP call; harmless after it.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    global _lib
    if _lib is not None:
        return _lib
    p = library_path()
    if not p.exists():
        raise PlumeHipError(f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            f"(or `make -C zk-nullifier-sig_amd/csrc`). There is no CPU fallback.")
    # torch ships its own libamdhip64.so (same SONAME as /opt/rocm's).  Whichever copy is loaded first serves the
    # whole process; loading ours first leaves torch without a usable device ("No HIP GPUs are available").  So if
    # torch is installed, let it load its runtime first; the library itself has no torch dependency.
    if not os.environ.get("PLUME_NO_TORCH_PRELOAD"):     # experiments only: run on /opt/rocm's runtime instead of the one torch bundles (tests/gpu_debug/d2h_in_library.py)
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(str(p))
    lib.plume_last_error.restype = C.c_char_p
    lib.plume_version.restype = C.c_char_p
    lib.plume_microbench.restype = C.c_double
    lib.plume_microbench.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.plume_microbench_last_ticks.restype = C.c_double
    lib.plume_microbench_last_ticks.argtypes = [C.POINTER(C.c_float)]
    lib.plume_init.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.plume_init_multi.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int]
    lib.plume_num_shards.argtypes = [C.c_void_p]
    lib.plume_shard_numa_node.argtypes = [C.c_void_p, C.c_int]
    lib.plume_set_host_first_piece.argtypes = [C.c_void_p, C.c_size_t]
    lib.plume_set_host_register_min.argtypes = [C.c_void_p, C.c_size_t]
    lib.plume_set_host_tail_piece.argtypes = [C.c_void_p, C.c_size_t]
    lib.plume_host_alloc.restype = C.c_void_p
    lib.plume_host_alloc.argtypes = [C.c_size_t]
    lib.plume_host_free.argtypes = [C.c_void_p]
    lib.plume_host_register.argtypes = [C.c_void_p, C.c_size_t]
    lib.plume_host_unregister.argtypes = [C.c_void_p]
    lib.plume_destroy.argtypes = [C.c_void_p]
    lib.plume_set_chunk.argtypes = [C.c_void_p, C.c_size_t]
    lib.plume_set_sub_batches.argtypes = [C.c_void_p, C.c_int]
    lib.plume_set_in_flight.argtypes = [C.c_void_p, C.c_int]
    # later entry points: an older build selected through PLUME_HIP_LIB (A/B runs against an earlier round) may lack them; the in-tree library must have them all (checked below)
    for name, args in (("plume_set_sign_uniform", [C.c_void_p, C.c_int]), ("plume_get_sign_uniform", [C.c_void_p]), ("plume_set_host_lanes", [C.c_void_p, C.c_int]),
                       ("plume_set_eq1_short", [C.c_void_p, C.c_int]), ("plume_set_stage_timing", [C.c_void_p, C.c_int])):
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.argtypes = args
    lib.plume_get_eq1_short.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    lib.plume_last_msm_clock.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.plume_last_msm_kernel.restype = C.c_char_p
    lib.plume_last_msm_kernel.argtypes = [C.c_void_p]
    lib.plume_set_host_piece.argtypes = [C.c_void_p, C.c_size_t]
    lib.plume_last_redo_tasks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.plume_last_stage_times.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    lib.plume_verify_batch.argtypes = [vp, i, sz] + [vp] * 9
    lib.plume_verify_batch_sec1.argtypes = [vp, i, sz] + [vp] * 9
    lib.plume_verify_non_zk_batch.argtypes = [vp, i, sz] + [vp] * 9
    lib.plume_verify_non_zk_batch_device.argtypes = [vp, i, sz, vp, vp, sz] + [vp] * 8
    lib.plume_verify_batch_sec1_device.argtypes = [vp, i, sz, vp, vp, sz] + [vp] * 8
    lib.plume_sign_batch.argtypes = [vp, i, sz] + [vp] * 12
    lib.plume_sign_batch_sec1.argtypes = [vp, i, sz] + [vp] * 12
    lib.plume_sign_batch_sec1_device.argtypes = [vp, i, sz, vp, vp, sz] + [vp] * 11
    lib.plume_nullifier_first_occurrence.argtypes = [vp, sz, vp, vp, vp, vp, C.POINTER(C.c_uint64)]
    lib.plume_nullifier_first_occurrence_device.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp]
    lib.plume_hash_to_curve_batch.argtypes = [vp, sz] + [vp] * 4
    lib.plume_verify_batch_device.argtypes = [vp, i, sz, vp, vp, sz] + [vp] * 8
    lib.plume_sign_batch_device.argtypes = [vp, i, sz, vp, vp, sz] + [vp] * 11
    lib.plume_hash_to_curve_batch_device.argtypes = [vp, sz, vp, vp, sz, vp, vp, vp]
    lib.plume_h2c_intermediates_batch.argtypes = [vp, sz, vp, vp, vp, i, vp, vp, vp, vp]
    lib.plume_h2c_intermediates_batch_device.argtypes = [vp, sz, vp, vp, sz, vp, i, vp, vp, vp, vp, vp]
    lib.plume_h2c_hints_batch.argtypes = [vp, sz, vp, vp, vp, i, vp]
    lib.plume_h2c_hints_batch_device.argtypes = [vp, sz, vp, vp, sz, vp, i, vp, vp]
    lib.plume_registers_from_be.argtypes = [sz, vp, vp]
    lib.plume_scalars_to_sec1_der_batch.argtypes = [vp, sz, vp, vp, vp]
    lib.plume_scalars_to_sec1_der_batch_device.argtypes = [vp, sz, vp, vp, vp, vp]
    lib.plume_sec1_der_to_scalars.argtypes = [sz, vp, vp, vp]
    lib.plume_sec1_der_to_scalars_checked.argtypes = [vp, sz, vp, vp, vp]
    lib.plume_registers_from_be_device.argtypes = [vp, sz, vp, vp, vp]
    lib.plume_aggregate_check.argtypes = [vp, i, i, sz] + [vp] * 11
    lib.plume_aggregate_check_device.argtypes = [vp, i, i, sz, vp, vp, sz] + [vp] * 7 + [C.c_uint64, vp, vp, vp]
    _lib = lib
    ver = tuple(int(x) for x in lib.plume_version().decode().split()[1].split(".")[:2])
    if ver < (0, 6) and not os.environ.get("PLUME_HIP_LIB"):
        _lib = None
        raise PlumeHipError(f"{p} is {lib.plume_version().decode()}: this module needs plume_hip >= 0.6 (rebuild: make -C zk-nullifier-sig_amd/csrc)")
    return lib


def exported_symbols():
    """every entry point include/plume_hip.h declares (used by the CPU-side ABI test)"""
    return ["plume_set_stage_timing", "plume_set_in_flight", "plume_set_sign_uniform", "plume_get_sign_uniform", "plume_set_host_lanes", "plume_set_eq1_short", "plume_get_eq1_short", "plume_last_msm_kernel", "plume_last_msm_clock", "plume_last_redo_tasks", "plume_h2c_hints_batch", "plume_h2c_hints_batch_device", "plume_shard_numa_node", "plume_set_sub_batches", "plume_aggregate_check", "plume_aggregate_check_device", "plume_init_multi", "plume_num_shards", "plume_set_host_first_piece", "plume_set_host_register_min", "plume_set_host_tail_piece", "plume_host_alloc", "plume_host_free", "plume_host_register",
            "plume_host_unregister", "plume_verify_non_zk_batch", "plume_verify_non_zk_batch_device", "plume_h2c_intermediates_batch", "plume_h2c_intermediates_batch_device",
            "plume_registers_from_be", "plume_registers_from_be_device", "plume_scalars_to_sec1_der_batch", "plume_scalars_to_sec1_der_batch_device", "plume_sec1_der_to_scalars", "plume_sec1_der_to_scalars_checked",
            "plume_init", "plume_destroy", "plume_last_error", "plume_version", "plume_set_chunk", "plume_set_host_piece", "plume_verify_batch", "plume_verify_batch_sec1", "plume_verify_batch_sec1_device", "plume_sign_batch", "plume_sign_batch_sec1", "plume_sign_batch_sec1_device",
            "plume_hash_to_curve_batch", "plume_nullifier_first_occurrence", "plume_nullifier_first_occurrence_device", "plume_verify_batch_device", "plume_sign_batch_device", "plume_hash_to_curve_batch_device",
            "plume_last_stage_times", "plume_microbench", "plume_microbench_last_ticks"]


def pack_messages(msgs):
    """list of bytes -> (packed uint8 array, uint64 offsets[n+1])"""
    off = np.zeros(len(msgs) + 1, dtype=np.uint64)
    if len(msgs):
        off[1:] = np.cumsum([len(m) for m in msgs], dtype=np.uint64)
    buf = np.frombuffer(b"".join(msgs) + b"\0" * 16, dtype=np.uint8).copy()
    return buf, off


def _np(a, width, n, name):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.size != width * n:
        raise ValueError(f"{name}: expected {n} records of {width} bytes, got {a.size} bytes")
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


AGG_RESULT_BYTES = 72


def parse_aggregate_record(rec):
    """the 72-byte result of plume_aggregate_check (include/plume_hip.h)"""
    rec = np.asarray(rec, dtype=np.uint8)
    return dict(all_ok=bool(rec[0]), identity=bool(rec[1]), n_bad=int.from_bytes(rec[4:8].tobytes(), "little"), point=rec[8:72].tobytes())


def registers_from_be(values):
    """(.., 32) uint8 big-endian values -> (.., 4) uint64 little-endian 64-bit registers (circuits/circom/utils.ts:11-17)"""
    lib = _load()
    v = np.ascontiguousarray(values, dtype=np.uint8)
    if v.shape[-1] != 32:
        raise ValueError("values must be 32-byte records")
    out = np.zeros(v.shape[:-1] + (4,), dtype=np.uint64)
    rc = lib.plume_registers_from_be(v.size // 32, _ptr(v), _ptr(out))
    if rc != 0:
        raise PlumeHipError(f"plume_registers_from_be failed ({rc}): {lib.plume_last_error().decode()}")
    return out


def sec1_der_to_scalars(der109):
    """(n, 109) SEC1-DER secret-key records (the wasm layer's `s` / `digest_private`) -> (scalars (n, 32), ok (n,)); the STRUCTURE half of SecretKey::from_sec1_der only
    (shape + scalar range, no GPU): Engine.sec1_der_to_scalars also checks the embedded public key against scalar * G, as the reference does"""
    lib = _load()
    d = np.ascontiguousarray(der109, dtype=np.uint8).reshape(-1, 109)
    sc, ok = np.zeros((len(d), 32), dtype=np.uint8), np.zeros(len(d), dtype=np.uint8)
    rc = lib.plume_sec1_der_to_scalars(len(d), _ptr(d), _ptr(sc), _ptr(ok))
    if rc != 0:
        raise PlumeHipError(f"plume_sec1_der_to_scalars failed ({rc}): {lib.plume_last_error().decode()}")
    return sc, ok


def pinned_empty(shape, dtype=np.uint8):
    """numpy array in page-locked host memory (plume_host_alloc): the copy engines read / write it directly, so the host-pointer calls
    overlap every transfer with the neighbouring pieces' kernels.  Freed when the array (and every view of it) is gone."""
    lib = _load()
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(x) for x in shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
    p = lib.plume_host_alloc(max(nbytes, 1))
    if not p:
        raise PlumeHipError(f"plume_host_alloc({nbytes}) failed: {lib.plume_last_error().decode()}")

    class _Owner:
        def __init__(self, ptr):
            self.ptr = ptr

        def __del__(self):
            try:
                lib.plume_host_free(self.ptr)
            except Exception:
                pass

    buf = (C.c_uint8 * max(nbytes, 1)).from_address(p)
    buf._plume_owner = _Owner(p)   # the ctypes buffer is the base object of the numpy array: it keeps the owner alive
    return np.frombuffer(buf, dtype=np.uint8, count=nbytes).view(dtype).reshape(shape)


def pinned_copy(a):
    out = pinned_empty(a.shape, a.dtype)
    out[...] = a
    return out


class Engine:
    """One context (include/plume_hip.h: plume_ctx).  Engine(k) = one GPU (plume_init; one Engine per process rank / device);
    Engine([k0, k1, ..]) = a multi-device context (plume_init_multi): the host-pointer calls shard every batch over the devices."""

    def __init__(self, device_id=None):
        lib = _load()
        if device_id is None:
            device_id = int(os.environ.get("LOCAL_RANK", "0"))
        self._lib = lib
        self._ctx = C.c_void_p()
        if isinstance(device_id, (list, tuple)):
            ids = (C.c_int * len(device_id))(*[int(d) for d in device_id])
            rc = lib.plume_init_multi(C.byref(self._ctx), ids, len(device_id))
            what = f"plume_init_multi(devices {list(device_id)})"
            self.device_ids = [int(d) for d in device_id]
            self.device_id = self.device_ids[0] if self.device_ids else 0
        else:
            rc = lib.plume_init(C.byref(self._ctx), int(device_id))
            what = f"plume_init(device {device_id})"
            self.device_id = int(device_id)
            self.device_ids = [self.device_id]
        if rc != 0:
            self._ctx = None
            raise PlumeHipError(f"{what} failed ({rc}): {lib.plume_last_error().decode()}")

    def num_shards(self):
        return int(self._lib.plume_num_shards(self._ctx))

    def shard_numa_nodes(self):
        """per shard: the NUMA node its worker thread was bound to (-1 = not bound)"""
        return [int(self._lib.plume_shard_numa_node(self._ctx, d)) for d in range(self.num_shards())]

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.plume_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise PlumeHipError(f"{what} failed ({rc}): {self._lib.plume_last_error().decode()}")

    def version(self):
        return self._lib.plume_version().decode()

    def set_chunk(self, n):
        self._chk(self._lib.plume_set_chunk(self._ctx, int(n)), "plume_set_chunk")

    def set_sub_batches(self, k):
        """device-resident verify / sign: number of overlapped sub-batches per call; 1 = strictly serial launch order (per-kernel stage times)"""
        self._chk(self._lib.plume_set_sub_batches(self._ctx, int(k)), "plume_set_sub_batches")

    def set_in_flight(self, k):
        """batches in flight (plume_set_in_flight): with k = 2 the device-resident calls go in turn to two lanes of the context, so that calls issued on different streams run
        side by side (2^20 verifies: about 1 % per batch; calls of fewer than 2^17 items stay on the first lane); default 1; results do not depend on it"""
        self._chk(self._lib.plume_set_in_flight(self._ctx, int(k)), "plume_set_in_flight")

    def set_sign_uniform(self, on):
        """the signer's uniform schedule (plume_set_sign_uniform): level 1 (or True) = no branch on a digit of sk or r, level 2 = and no table address derived from
        one (every row of a window's table is read); outputs unchanged"""
        self._chk(self._lib.plume_set_sign_uniform(self._ctx, int(on)), "plume_set_sign_uniform")

    def sign_uniform(self):
        """the level this context signs at (plume_get_sign_uniform): 1 by default since library 0.4"""
        rc = self._lib.plume_get_sign_uniform(self._ctx)
        if rc < 0:
            self._chk(rc, "plume_get_sign_uniform")
        return rc

    def set_eq1_short(self, mode):
        """the verifier's first equation where R is given: 1 = short form for calls of at least eq1_short()[1] items (csrc/plume_eis.h; the default), 3 = short form whatever the
        size, 0 = long form always, 2 = test mode (every item through the scalar stage's fallback)"""
        self._chk(self._lib.plume_set_eq1_short(self._ctx, int(mode)), "plume_set_eq1_short")

    def eq1_short(self):
        """(mode, min_items) in force on this context (plume_get_eq1_short): mode as set_eq1_short takes it, min_items = the smallest call that takes the short form in mode 1"""
        m = C.c_size_t(0)
        rc = self._lib.plume_get_eq1_short(self._ctx, C.byref(m))
        if rc < 0:
            self._chk(rc, "plume_get_eq1_short")
        return rc, int(m.value)

    def last_msm_clock_ghz(self):
        """the shader clock (GHz) the multi-scalar kernel of the last verify call ran at, sampled inside the kernel (plume_last_msm_clock; stage timing must be on), or None"""
        g = C.c_double(0.0)
        rc = self._lib.plume_last_msm_clock(self._ctx, C.byref(g))
        return float(g.value) if rc == 0 else None

    def last_msm_kernel(self):
        """the multi-scalar kernel the last verify call on this context launched (plume_last_msm_kernel): 'k_verify_msm', 'k_verify_msm_s', 'k_verify_msm_pair', or None"""
        r = self._lib.plume_last_msm_kernel(self._ctx)
        return r.decode() if r else None

    def set_host_lanes(self, lanes):
        """host-pointer calls: 1 = every piece on the context itself, 2 (default) = pieces alternate between the context and a second lane"""
        self._chk(self._lib.plume_set_host_lanes(self._ctx, int(lanes)), "plume_set_host_lanes")

    def set_host_piece(self, n):
        """host-pointer calls: items per pipelined piece (upload / compute / download overlap across pieces)"""
        self._chk(self._lib.plume_set_host_piece(self._ctx, int(n)), "plume_set_host_piece")

    def set_host_first_piece(self, n):
        self._chk(self._lib.plume_set_host_first_piece(self._ctx, int(n)), "plume_set_host_first_piece")

    def set_host_tail_piece(self, n):
        self._chk(self._lib.plume_set_host_tail_piece(self._ctx, int(n)), "plume_set_host_tail_piece")

    def set_host_register_min(self, nbytes):
        """host-pointer calls: page-lock pageable caller arrays of at least nbytes for the duration of the call (0 = never)"""
        self._chk(self._lib.plume_set_host_register_min(self._ctx, int(nbytes)), "plume_set_host_register_min")

    # ------------------------------------------------------------------ host-pointer API (numpy in, numpy out)
    def verify_non_zk_batch(self, version, msgs, msg_off, pk, nullifier, s, r_point, hashed_to_curve_r, digest_private, out=None):
        """plume_arkworks' verify_non_zk (rust-arkworks/src/tests.rs:28-78), batched: 1 Ok(true), 0 Ok(false), 2 Err(HashToCurveError)"""
        n = len(msg_off) - 1
        msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
        msg_off = np.ascontiguousarray(msg_off, dtype=np.uint64)
        pk, nullifier, s = _np(pk, 64, n, "pk"), _np(nullifier, 64, n, "nullifier"), _np(s, 32, n, "s")
        r_point, hashed_to_curve_r = _np(r_point, 64, n, "r_point"), _np(hashed_to_curve_r, 64, n, "hashed_to_curve_r")
        digest_private = _np(digest_private, 32, n, "digest_private")
        ok = np.zeros(n, dtype=np.uint8) if out is None else out
        self._chk(self._lib.plume_verify_non_zk_batch(self._ctx, int(version), n, _ptr(msgs), _ptr(msg_off), _ptr(pk), _ptr(nullifier), _ptr(s), _ptr(r_point),
                                                      _ptr(hashed_to_curve_r), _ptr(digest_private), _ptr(ok)), "plume_verify_non_zk_batch")
        return ok

    def aggregate_check(self, version, msgs, msg_off, pk, nullifier, c, s, r_point, hashed_to_curve_r, seed=None, mode=0):
        """Aggregate random-linear-combination pre-filter (include/plume_hip.h plume_aggregate_check): all-or-nothing and
        probabilistic, never a replacement for verify_batch.  mode 0: PlumeSignature::verify of V1 signatures; mode 1: verify_non_zk (c = digest_private).
        seed: 32 bytes the producer of the batch cannot predict (default: os.urandom).  Returns dict(all_ok, identity, n_bad, point, hash_ok, seed)."""
        import os
        n = len(msg_off) - 1
        msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
        msg_off = np.ascontiguousarray(msg_off, dtype=np.uint64)
        pk, nullifier, c, s = _np(pk, 64, n, "pk"), _np(nullifier, 64, n, "nullifier"), _np(c, 32, n, "c"), _np(s, 32, n, "s")
        r_point, hashed_to_curve_r = _np(r_point, 64, n, "r_point"), _np(hashed_to_curve_r, 64, n, "hashed_to_curve_r")
        seed = os.urandom(32) if seed is None else bytes(seed)
        if len(seed) != 32:
            raise ValueError("seed must be 32 bytes")
        sd = np.frombuffer(seed, dtype=np.uint8).copy()
        hash_ok = np.zeros(n, dtype=np.uint8)
        rec = np.zeros(AGG_RESULT_BYTES, dtype=np.uint8)
        self._chk(self._lib.plume_aggregate_check(self._ctx, int(version), int(mode), n, _ptr(msgs), _ptr(msg_off), _ptr(pk), _ptr(nullifier), _ptr(c), _ptr(s), _ptr(r_point),
                                                  _ptr(hashed_to_curve_r), _ptr(sd), _ptr(hash_ok), _ptr(rec)), "plume_aggregate_check")
        out = parse_aggregate_record(rec)
        out["hash_ok"] = hash_ok
        out["seed"] = seed
        return out

    def verify_batch(self, version, msgs, msg_off, pk, nullifier, c, s, r_point=None, hashed_to_curve_r=None, out=None):
        n = len(msg_off) - 1
        msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
        msg_off = np.ascontiguousarray(msg_off, dtype=np.uint64)
        pk, nullifier, c, s = _np(pk, 64, n, "pk"), _np(nullifier, 64, n, "nullifier"), _np(c, 32, n, "c"), _np(s, 32, n, "s")
        if version == 1:
            if r_point is None or hashed_to_curve_r is None:
                raise ValueError("V1 verification needs r_point and hashed_to_curve_r")
            r_point, hashed_to_curve_r = _np(r_point, 64, n, "r_point"), _np(hashed_to_curve_r, 64, n, "hashed_to_curve_r")
        else:
            r_point = hashed_to_curve_r = None
        ok = np.zeros(n, dtype=np.uint8) if out is None else out
        self._chk(self._lib.plume_verify_batch(self._ctx, int(version), n, _ptr(msgs), _ptr(msg_off), _ptr(pk), _ptr(nullifier), _ptr(c), _ptr(s),
                                               _ptr(r_point), _ptr(hashed_to_curve_r), _ptr(ok)), "plume_verify_batch")
        return ok

    def verify_batch_sec1(self, version, msgs, msg_off, pk33, nullifier33, c, s, r_point33=None, hashed_to_curve_r33=None):
        """verify with 33-byte SEC1-compressed points (decompressed and validated on the GPU)"""
        n = len(msg_off) - 1
        msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
        msg_off = np.ascontiguousarray(msg_off, dtype=np.uint64)
        pk33, nullifier33, c, s = _np(pk33, 33, n, "pk33"), _np(nullifier33, 33, n, "nullifier33"), _np(c, 32, n, "c"), _np(s, 32, n, "s")
        if version == 1:
            if r_point33 is None or hashed_to_curve_r33 is None:
                raise ValueError("V1 verification needs r_point and hashed_to_curve_r")
            r_point33, hashed_to_curve_r33 = _np(r_point33, 33, n, "r_point33"), _np(hashed_to_curve_r33, 33, n, "hashed_to_curve_r33")
        else:
            r_point33 = hashed_to_curve_r33 = None
        ok = np.zeros(n, dtype=np.uint8)
        self._chk(self._lib.plume_verify_batch_sec1(self._ctx, int(version), n, _ptr(msgs), _ptr(msg_off), _ptr(pk33), _ptr(nullifier33), _ptr(c), _ptr(s),
                                                    _ptr(r_point33), _ptr(hashed_to_curve_r33), _ptr(ok)), "plume_verify_batch_sec1")
        return ok

    def sign_batch(self, version, msgs, msg_off, sk, r, pk_in=None, out=None):
        """out: optional dict of preallocated arrays (pk, nullifier, c, s, r_point, hashed_to_curve_r, status), e.g. page-locked ones"""
        n = len(msg_off) - 1
        msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
        msg_off = np.ascontiguousarray(msg_off, dtype=np.uint64)
        sk, r = _np(sk, 32, n, "sk"), _np(r, 32, n, "r")
        pk_in = None if pk_in is None else _np(pk_in, 64, n, "pk_in")
        o = out if out is not None else {k: np.zeros((n, w), dtype=np.uint8) for k, w in
                                         [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
        status = o["status"] if out is not None else np.zeros(n, dtype=np.uint8)
        self._chk(self._lib.plume_sign_batch(self._ctx, int(version), n, _ptr(msgs), _ptr(msg_off), _ptr(sk), _ptr(r), _ptr(pk_in), _ptr(o["pk"]),
                                             _ptr(o["nullifier"]), _ptr(o["c"]), _ptr(o["s"]), _ptr(o["r_point"]), _ptr(o["hashed_to_curve_r"]),
                                             _ptr(status)), "plume_sign_batch")
        o["status"] = status
        return o

    def sign_batch_sec1(self, version, msgs, msg_off, sk, r, pk_in=None):
        """sign_batch with pk, nullifier, r_point, hashed_to_curve_r as 33-byte SEC1-compressed records"""
        n = len(msg_off) - 1
        msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
        msg_off = np.ascontiguousarray(msg_off, dtype=np.uint64)
        sk, r = _np(sk, 32, n, "sk"), _np(r, 32, n, "r")
        pk_in = None if pk_in is None else _np(pk_in, 64, n, "pk_in")
        o = {k: np.zeros((n, w), dtype=np.uint8) for k, w in
             [("pk", 33), ("nullifier", 33), ("c", 32), ("s", 32), ("r_point", 33), ("hashed_to_curve_r", 33)]}
        status = np.zeros(n, dtype=np.uint8)
        self._chk(self._lib.plume_sign_batch_sec1(self._ctx, int(version), n, _ptr(msgs), _ptr(msg_off), _ptr(sk), _ptr(r), _ptr(pk_in), _ptr(o["pk"]),
                                                  _ptr(o["nullifier"]), _ptr(o["c"]), _ptr(o["s"]), _ptr(o["r_point"]), _ptr(o["hashed_to_curve_r"]),
                                                  _ptr(status)), "plume_sign_batch_sec1")
        o["status"] = status
        return o

    def hash_to_curve_batch(self, msgs, msg_off, pk=None):
        n = len(msg_off) - 1
        msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
        msg_off = np.ascontiguousarray(msg_off, dtype=np.uint64)
        pk = None if pk is None else _np(pk, 64, n, "pk")
        h = np.zeros((n, 64), dtype=np.uint8)
        self._chk(self._lib.plume_hash_to_curve_batch(self._ctx, n, _ptr(msgs), _ptr(msg_off), _ptr(pk), _ptr(h)), "plume_hash_to_curve_batch")
        return h

    def h2c_intermediates_batch(self, msgs, msg_off, pk=None, registers=False):
        """circuit witness hints from the GPU hash_to_curve: dict u (n,2,·), mapped (n,4,·), q (n,4,·), h (n,2,·); values are 32
        big-endian bytes (uint8, last axis 32) or, with registers=True, 4 little-endian 64-bit registers (uint64, last axis 4)"""
        n = len(msg_off) - 1
        msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
        msg_off = np.ascontiguousarray(msg_off, dtype=np.uint64)
        pk = None if pk is None else _np(pk, 64, n, "pk")
        o = {k: np.zeros((n, w, 32), dtype=np.uint8) for k, w in [("u", 2), ("mapped", 4), ("q", 4), ("h", 2)]}
        self._chk(self._lib.plume_h2c_intermediates_batch(self._ctx, n, _ptr(msgs), _ptr(msg_off), _ptr(pk), 1 if registers else 0, _ptr(o["u"]), _ptr(o["mapped"]),
                                                          _ptr(o["q"]), _ptr(o["h"])), "plume_h2c_intermediates_batch")
        if registers:
            o = {k: v.view(np.uint64) for k, v in o.items()}
        return o

    def h2c_hints_batch(self, msgs, msg_off, pk=None, registers=False):
        """the circuit's square-root hints (UNPINNED definitions, include/plume_hip.h): dict q0_gx1_sqrt, q0_gx2_sqrt, q0_y_pos, q1_... -> (n, 32) bytes or (n, 4) uint64 registers"""
        msg_off = np.ascontiguousarray(msg_off, dtype=np.uint64)
        n = len(msg_off) - 1
        msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
        pk = None if pk is None else _np(pk, 64, n, "pk")
        out = np.zeros((n, 6, 32), dtype=np.uint8)
        self._chk(self._lib.plume_h2c_hints_batch(self._ctx, n, _ptr(msgs), _ptr(msg_off), _ptr(pk), 1 if registers else 0, _ptr(out)), "plume_h2c_hints_batch")
        names = ["q0_gx1_sqrt", "q0_gx2_sqrt", "q0_y_pos", "q1_gx1_sqrt", "q1_gx2_sqrt", "q1_y_pos"]
        return {nm: (np.ascontiguousarray(out[:, k]).view(np.uint64) if registers else np.ascontiguousarray(out[:, k])) for k, nm in enumerate(names)}

    def scalars_to_sec1_der_batch(self, scalars):
        """SecretKey::from(scalar).to_sec1_der() for a batch (javascript/src/lib.rs:98-110): (n, 109) records incl. the public key scalar*G computed on the GPU,
        and a status array (2 = scalar outside [1, n-1], record zeroed)"""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        n = len(scalars)
        der, status = np.zeros((n, 109), dtype=np.uint8), np.zeros(n, dtype=np.uint8)
        self._chk(self._lib.plume_scalars_to_sec1_der_batch(self._ctx, n, _ptr(scalars), _ptr(der), _ptr(status)), "plume_scalars_to_sec1_der_batch")
        return der, status

    def sec1_der_to_scalars(self, der109):
        """SecretKey::from_sec1_der with the reference's semantics: structure, scalar range and public key == scalar * G (recomputed on the GPU) -> (scalars (n, 32), ok (n,))"""
        d = np.ascontiguousarray(der109, dtype=np.uint8).reshape(-1, 109)
        sc, ok = np.zeros((len(d), 32), dtype=np.uint8), np.zeros(len(d), dtype=np.uint8)
        self._chk(self._lib.plume_sec1_der_to_scalars_checked(self._ctx, len(d), _ptr(d), _ptr(sc), _ptr(ok)), "plume_sec1_der_to_scalars_checked")
        return sc, ok

    def nullifier_first_occurrence(self, nullifier, live=None, ids=None):
        """first[i] = 1 iff item i is live and holds the smallest id (default: position) among the live items with the same 64-byte
        nullifier; returns (first uint8[n], number of ones)"""
        nullifier = np.ascontiguousarray(nullifier, dtype=np.uint8).reshape(-1, 64)
        n = len(nullifier)
        live = None if live is None else np.ascontiguousarray(live, dtype=np.uint8).reshape(n)
        ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint64).reshape(n)
        first = np.zeros(n, dtype=np.uint8)
        cnt = C.c_uint64(0)
        self._chk(self._lib.plume_nullifier_first_occurrence(self._ctx, n, _ptr(nullifier), _ptr(live), _ptr(ids), _ptr(first), C.byref(cnt)),
                  "plume_nullifier_first_occurrence")
        return first, int(cnt.value)

    # ------------------------------------------------------------------ device-resident API (torch uint8 tensors on this GPU)
    @staticmethod
    def _dp(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def verify_batch_device(self, version, n, msgs, msg_off, msgs_bytes, pk, nullifier, c, s, r_point, hashed_to_curve_r, ok, stream=None):
        """all tensors on cuda:<device_id>; enqueues on `stream` (torch.cuda.Stream or None = current stream); does not synchronise"""
        import torch
        st = (stream or torch.cuda.current_stream(self.device_id)).cuda_stream
        d = self._dp
        self._chk(self._lib.plume_verify_batch_device(self._ctx, int(version), int(n), d(msgs), d(msg_off), int(msgs_bytes), d(pk), d(nullifier), d(c), d(s),
                                                      d(r_point), d(hashed_to_curve_r), d(ok), C.c_void_p(st)), "plume_verify_batch_device")

    def verify_non_zk_batch_device(self, version, n, msgs, msg_off, msgs_bytes, pk, nullifier, s, r_point, hashed_to_curve_r, digest_private, ok, stream=None):
        import torch
        st = (stream or torch.cuda.current_stream(self.device_id)).cuda_stream
        d = self._dp
        self._chk(self._lib.plume_verify_non_zk_batch_device(self._ctx, int(version), int(n), d(msgs), d(msg_off), int(msgs_bytes), d(pk), d(nullifier), d(s), d(r_point),
                                                             d(hashed_to_curve_r), d(digest_private), d(ok), C.c_void_p(st)), "plume_verify_non_zk_batch_device")

    def aggregate_check_device(self, version, mode, n, msgs, msg_off, msgs_bytes, pk, nullifier, c, s, r_point, hashed_to_curve_r, seed, index_base, hash_ok, result, stream=None):
        """device tensors; seed: 32 host bytes; hash_ok: uint8[n] or None; result: uint8[72] (parse_aggregate_record after synchronising)"""
        import torch
        st = (stream or torch.cuda.current_stream(self.device_id)).cuda_stream
        d = self._dp
        sd = np.frombuffer(bytes(seed), dtype=np.uint8).copy()
        self._chk(self._lib.plume_aggregate_check_device(self._ctx, int(version), int(mode), int(n), d(msgs), d(msg_off), int(msgs_bytes), d(pk), d(nullifier), d(c), d(s), d(r_point),
                                                         d(hashed_to_curve_r), _ptr(sd), int(index_base), d(hash_ok), d(result), C.c_void_p(st)), "plume_aggregate_check_device")

    def verify_batch_sec1_device(self, version, n, msgs, msg_off, msgs_bytes, pk33, nullifier33, c, s, r_point33, hashed_to_curve_r33, ok, stream=None):
        import torch
        st = (stream or torch.cuda.current_stream(self.device_id)).cuda_stream
        d = self._dp
        self._chk(self._lib.plume_verify_batch_sec1_device(self._ctx, int(version), int(n), d(msgs), d(msg_off), int(msgs_bytes), d(pk33), d(nullifier33), d(c), d(s),
                                                           d(r_point33), d(hashed_to_curve_r33), d(ok), C.c_void_p(st)), "plume_verify_batch_sec1_device")

    def sign_batch_device(self, version, n, msgs, msg_off, msgs_bytes, sk, r, pk_in, pk, nullifier, c, s, r_point, hashed_to_curve_r, status, stream=None):
        import torch
        st = (stream or torch.cuda.current_stream(self.device_id)).cuda_stream
        d = self._dp
        self._chk(self._lib.plume_sign_batch_device(self._ctx, int(version), int(n), d(msgs), d(msg_off), int(msgs_bytes), d(sk), d(r), d(pk_in), d(pk),
                                                    d(nullifier), d(c), d(s), d(r_point), d(hashed_to_curve_r), d(status), C.c_void_p(st)),
                  "plume_sign_batch_device")

    def nullifier_first_occurrence_device(self, n, nullifier, live, ids, first, n_unique=None, stream=None):
        """tensors on cuda:<device_id> (nullifier n x 64 uint8, live / first uint8[n] or None, ids int64/uint64[n] or None, n_unique one 64-bit word or None)"""
        import torch
        st = (stream or torch.cuda.current_stream(self.device_id)).cuda_stream
        d = self._dp
        self._chk(self._lib.plume_nullifier_first_occurrence_device(self._ctx, int(n), d(nullifier), d(live), d(ids), d(first), d(n_unique), C.c_void_p(st)),
                  "plume_nullifier_first_occurrence_device")

    # ------------------------------------------------------------------ measurement
    def set_stage_timing(self, on):
        """per-stage timing events inside the device pipelines (plume_set_stage_timing): OFF by default since library 0.5 -- an event between two kernels costs ~6 us of idle
        GPU, five or six per call.  Env PLUME_STAGE_TIMES=1 turns it on for new contexts."""
        fn = getattr(self._lib, "plume_set_stage_timing", None)
        if fn is None:
            return                                   # an older library (PLUME_HIP_LIB): always on
        self._chk(fn(self._ctx, 1 if on else 0), "plume_set_stage_timing")

    def last_stage_times(self):
        """(stage, ms) of the last device-resident call; needs set_stage_timing(True) (or PLUME_STAGE_TIMES=1) BEFORE that call"""
        cap = 512
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        k = self._lib.plume_last_stage_times(self._ctx, names, ms, cap)
        if k < 0:
            raise PlumeHipError(f"plume_last_stage_times failed ({k}): {self._lib.plume_last_error().decode()}")
        return [(names[i].decode(), float(ms[i])) for i in range(min(k, cap))]

    def last_redo_tasks(self):
        """multi-scalar tasks of the last verify call that met p == +-q in an unchecked addition and were redone with checked additions (0 for honest batches)"""
        c = C.c_uint64(0)
        self._chk(self._lib.plume_last_redo_tasks(self._ctx, C.byref(c)), "plume_last_redo_tasks")
        return int(c.value)

    def microbench(self, kind, iters=4096):
        v = self._lib.plume_microbench(self._ctx, int(kind), int(iters))
        if v <= 0:
            raise PlumeHipError(f"plume_microbench failed: {self._lib.plume_last_error().decode()}")
        return float(v)

    def microbench_ticks(self):
        ms = C.c_float()
        t = self._lib.plume_microbench_last_ticks(C.byref(ms))
        return float(t), float(ms.value)


_default = None


def default_engine() -> Engine:
    global _default
    if _default is None:
        _default = Engine()
    return _default
