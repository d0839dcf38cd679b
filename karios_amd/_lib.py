"""ctypes binding of libkarios_hip.so (C ABI: include/karios_hip.h).

The product path has no CPU fallback: if the shared library is missing, or no
MI355X-class HIP device is visible, every entry point raises.  Nothing in this
package imports ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KARIOS_HIP_LIB") or os.path.join(_HERE, "libkarios_hip.so")   # (override: development builds)

KM_U8, KM_U16, KM_I16, KM_F32 = 0, 1, 2, 3
_DTYPES = {np.dtype("uint8"): KM_U8, np.dtype("uint16"): KM_U16, np.dtype("int16"): KM_I16,
           np.dtype("float32"): KM_F32}


class KariosHipError(RuntimeError):
    """A libkarios_hip call returned a negative status (cv2.error equivalent)."""


class KltParams(C.Structure):
    _fields_ = [("max_corners", C.c_int32), ("block_size", C.c_int32), ("win_size", C.c_int32),
                ("max_level", C.c_int32), ("max_count", C.c_int32), ("ksize_mon", C.c_int32),
                ("ksize_ref", C.c_int32), ("invert_mon", C.c_int32), ("quality_level", C.c_double),
                ("min_distance", C.c_double), ("epsilon", C.c_double)]


class KltStats(C.Structure):
    _fields_ = [("valid_pixels", C.c_int64), ("n_candidates", C.c_int64), ("n_init", C.c_int32),
                ("n_select_batches", C.c_int32), ("min_ref", C.c_double), ("max_ref", C.c_double),
                ("min_mon", C.c_double), ("max_mon", C.c_double), ("max_eig", C.c_float),
                ("emitted_ratio", C.c_float), ("path_flags", C.c_int32), ("tie_rows", C.c_int32)]


NAN_OUTSIDE_WINDOW = 0x7FF80000DEAD0000   # km_set_image_window: score of a chip outside the resident window
PATH_KEY_REGROW, PATH_STAGE_FALLBACK, PATH_SECOND_PASS, PATH_PREFIX_GROWN = 1, 2, 4, 8


class KmUnit(C.Structure):
    """km_unit of include/karios_hip.h: one work unit of km_klt_units_frame_submit."""
    _fields_ = [("d_ref", C.c_void_p), ("d_mon", C.c_void_p), ("sref", C.c_ssize_t), ("smon", C.c_ssize_t),
                ("d_ref_full", C.c_void_p), ("d_mon_full", C.c_void_p), ("sref_f", C.c_ssize_t), ("smon_f", C.c_ssize_t),
                ("H", C.c_int32), ("W", C.c_int32), ("Hf", C.c_int32), ("Wf", C.c_int32), ("x_off", C.c_float), ("y_off", C.c_float),
                ("win_ox", C.c_int32), ("win_oy", C.c_int32), ("win_H", C.c_int32), ("win_W", C.c_int32),
                ("d_mask", C.c_void_p), ("smask", C.c_ssize_t)]


UNITS_PER_SUBMISSION = 16      # KM_UNITS_PER_SUBMISSION
E_UNSUPPORTED = -4             # KM_E_UNSUPPORTED

_vp, _i, _d, _sz, _pd = C.c_void_p, C.c_int, C.c_double, C.c_ssize_t, C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int)

# name -> (restype, argtypes); mirrors include/karios_hip.h one to one
SIGNATURES = {
    "km_version": (_i, []),
    "km_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "km_ctx_destroy": (_i, [_vp]),
    "km_last_error": (C.c_char_p, [_vp]),
    "km_ctx_sync": (_i, [_vp]),
    "km_set_profiling": (_i, [_vp, _i]),
    "km_set_option": (_i, [_vp, C.c_char_p, _i]),
    "km_is_dev_build": (_i, []),
    "km_dev_counters": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "km_get_stage_ms": (_i, [_vp, C.POINTER(C.c_float), _i, _pi]),
    "km_stage_name": (C.c_char_p, [_i]),
    "km_get_klt_stats": (_i, [_vp, C.POINTER(KltStats)]),
    "km_dev_alloc": (_i, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "km_dev_free": (_i, [_vp, _vp]),
    "km_h2d": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "km_d2h": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "km_host_alloc": (_i, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "km_host_free": (_i, [_vp, _vp]),
    "km_upload_async": (_i, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_size_t, C.c_size_t]),
    "km_upload_wait": (_i, [_vp]),
    "km_upload_check_stats": (_i, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "km_upload_mark": (_i, [_vp, _pi]),
    "km_upload_join": (_i, [_vp, _i]),
    "km_set_frame_sink": (_i, [_vp, _vp, C.c_size_t]),
    "km_set_frame_sink_pitch": (_i, [_vp, _vp, C.c_size_t, C.c_size_t]),
    "km_klt_units_frame_submit": (_i, [_vp, C.POINTER(KmUnit), _i, _i, _pd, _pd, C.POINTER(KltParams), _d, _i, C.POINTER(C.c_int)]),
    "km_stream_wait_frame": (_i, [_vp, _i, _vp]),
    "km_phase_info": (_i, [_vp, _pi, _pd]),
    "km_phase_plan": (_i, [_i, _i, _vp, _i, _pi, _pi, _vp]),
    "km_set_image_window": (_i, [_vp, _i, _i, _i, _i]),
    "km_minmax_dev": (_i, [_vp, _vp, _i, _i, _i, _sz, _pd]),
    "km_band_prefilter_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _sz, _sz, _pd, _pd, _pd, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, C.POINTER(C.c_int64)]),
    "km_band_eigen_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _d, C.POINTER(C.c_uint)]),
    "km_band_keys_dev": (_i, [_vp, _vp, _i, _i, _d, C.c_uint, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "km_select_keys": (_i, [_vp, _vp, C.c_size_t, _i, _i, _i, _d, _vp, _i, _pi]),
    "km_sort_pairs_u64": (_i, [_vp, _vp, _vp, C.c_size_t, _i]),
    "km_exclusive_scan_u32": (_i, [_vp, _vp, _vp, C.c_size_t, _i]),
    "km_band_track_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _pi]),
    "km_to_uint8": (_i, [_vp, _vp, _i, _i, _i, _sz, _i, _vp, _pd]),
    "km_auto_mask": (_i, [_vp, _vp, _vp, _i, _i, _i, _sz, _sz, _pd, _pd, _vp, C.POINTER(C.c_int64)]),
    "km_lk_oscillation_probe": (_i, [_vp, _vp, _i, _vp]),
    "km_laplacian_u8": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "km_min_eigen": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "km_good_features": (_i, [_vp, _vp, _vp, _i, _i, _i, _d, _d, _i, _vp, _i, _pi]),
    "km_pyrdown_u8": (_i, [_vp, _vp, _i, _i, _vp]),
    "km_pyrlk": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _d, _vp]),
    "km_klt_track": (_i, [_vp, _vp, _vp, _vp, _i, _i, C.POINTER(KltParams), _vp, _i, _vp, _vp, _vp, _i, _pi]),
    "km_klt_tile": (_i, [_vp, _vp, _vp, _i, _i, _i, _sz, _sz, _vp, _pd, _pd, C.POINTER(KltParams), _vp, _vp,
                         _vp, _i, _pi]),
    "km_tile_prefilter": (_i, [_vp, _vp, _vp, _i, _i, _i, _sz, _sz, _pd, _pd, _i, _i, _i, _vp, _vp, _vp, C.POINTER(C.c_int64)]),
    "km_zncc_batch": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _sz, _sz, _vp, _vp, _vp, _vp, _i, _vp]),
    "km_mi_batch": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _sz, _sz, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "km_zncc_windows": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _sz, _sz, _vp, _i, _i, _vp, _vp]),
    "km_phase_shift": (_i, [_vp, _vp, _vp, _i, _i, _i, _sz, _sz, _pd]),
    "km_shift_image": (_i, [_vp, _vp, _i, _i, _i, _sz, _i, _i, _vp]),
    "km_klt_tile_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _sz, _sz, _vp, _sz, _pd, _pd, C.POINTER(KltParams), _vp,
                             _vp, _vp, _i, _vp]),
    "km_klt_tile_frame_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _sz, _sz, _vp, _sz, _pd, _pd, C.POINTER(KltParams), C.c_float,
                                   C.c_float, _vp, _i]),
    "km_klt_tile_frame_zncc_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _sz, _sz, _vp, _sz, _pd, _pd, C.POINTER(KltParams), C.c_float,
                                        C.c_float, _vp, _vp, _i, _i, _sz, _sz, _d, _vp, _i]),
    "km_klt_tile_frame_submit": (_i, [_vp, _vp, _vp, _i, _i, _i, _sz, _sz, _vp, _sz, _pd, _pd, C.POINTER(KltParams), C.c_float,
                                      C.c_float, _vp, _vp, _i, _i, _sz, _sz, _d, _i, C.POINTER(C.c_int)]),
    "km_frame_wait": (_i, [_vp, _i, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "km_frame_stage_ms": (_i, [_vp, _i, _vp, _i, C.POINTER(C.c_int)]),
    "km_frame_flush": (_i, [_vp, _i]),
    "km_zncc_batch_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _sz, _sz, _vp, _vp, _vp, _vp, _i, _vp]),
    "km_mi_batch_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _sz, _sz, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "km_klt_auto_ksize_frame_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _sz, _sz, _vp, _sz, _pd, _pd, C.POINTER(KltParams), _vp, _i, C.c_float,
                                         C.c_float, _vp, _i, _vp, _vp]),
    "km_dn_keep_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _sz, _sz, _vp, _vp, _i, _vp, _i, _pd, _pd, _vp]),
    "km_phase_shift_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _sz, _sz, _pd]),
    "km_shift_image_dev": (_i, [_vp, _vp, _i, _i, _i, _sz, _i, _i, _vp]),
}

_lib = None
_lib_lock = threading.Lock()


def _preload_shared_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (same SONAME as /opt/rocm's).  A process must use ONE
    HIP runtime: if the system copy is loaded first, torch later fails with "no ROCm-capable device".  When torch
    is installed (it is the multi-GPU plumbing of karios_amd.parallel), make its runtime the process-wide one
    before libkarios_hip.so is mapped; without torch the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # torch already brought its runtime
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libkarios_hip.so (built by `__graft_entry__.build()` / csrc/Makefile)."""
    global _lib
    with _lib_lock:
        if _lib is None:
            # KARIOS_HIP_LIB: another build of the same library, e.g. the development build karios_amd/libkarios_hip_dev.so
            # (make -C karios_amd/csrc DEV=1) for tools/ A-B measurements; the product loads the release library
            path = os.environ.get("KARIOS_HIP_LIB") or LIB_PATH
            if not os.path.exists(path):
                raise KariosHipError(
                    f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(karios_amd has no CPU fallback)")
            _preload_shared_hip_runtime()
            lib = C.CDLL(path)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            _lib = lib
    return _lib


_ANY_DTYPES = {np.dtype("float64"): 4, np.dtype("int32"): 5, np.dtype("uint32"): 6}


def any_dtype_code(arr: np.ndarray) -> int:
    """Pixel-type code for the entry points that read every numeric type (km_zncc_windows)."""
    code = _DTYPES.get(arr.dtype, _ANY_DTYPES.get(arr.dtype))
    if code is None:
        raise KariosHipError(f"unsupported pixel type {arr.dtype}")
    return code


def dtype_code(arr: np.ndarray) -> int:
    try:
        return _DTYPES[arr.dtype]
    except KeyError:
        raise KariosHipError(f"unsupported pixel type {arr.dtype} (uint8, uint16, int16, float32)") from None


def as_image(arr) -> np.ndarray:
    """2-D array whose rows are contiguous (row stride may exceed the width)."""
    a = np.asarray(arr)
    if a.ndim != 2:
        raise KariosHipError(f"expected a 2-D image, got shape {a.shape}")
    if a.shape[1] > 1 and a.strides[1] != a.itemsize or a.strides[0] % a.itemsize or a.strides[0] < 0:
        a = np.ascontiguousarray(a)
    if a.shape[0] > 1 and a.strides[0] < a.shape[1] * a.itemsize:
        a = np.ascontiguousarray(a)
    return a


def row_stride(a: np.ndarray) -> int:
    return a.strides[0] // a.itemsize if a.shape[0] > 1 else a.shape[1]


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One HIP stream + grow-only device workspace.  Not thread-safe: one per thread."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.km_ctx_create(int(device), C.byref(h))
        if rc != 0:
            msg = self.lib.km_last_error(None)
            raise KariosHipError(f"km_ctx_create(device={device}) failed ({rc}): {msg.decode() if msg else ''}")
        self.handle = h
        self.device = device
        # The context is not thread-safe and finalizers (`__del__` of buffers / pairs) run on whichever thread drops the last
        # reference or triggers the cyclic GC - e.g. FrameStream's worker.  Library calls of finalizers are therefore queued here
        # and run by the owning thread at its next call (`drain`) instead of racing with it.
        self._owner = threading.get_ident()
        self._abandoned = []
        self._abandoned_lock = threading.Lock()
        self._alive = [True]      # shared with the finalizers of this context's page-locked arrays (`pinned_empty`)
        # development / A-B measurements: KARIOS_HIP_OPTIONS="lk2=0,eig3=0" applies `set_option` to every new context
        for item in filter(None, os.environ.get("KARIOS_HIP_OPTIONS", "").split(",")):
            name, _, value = item.partition("=")
            self.set_option(name.strip(), int(value or 1))

    # ---- device buffers are recycled: a tile loop would otherwise pay a hipMalloc / hipFree pair (and their implicit device
    # synchronisations) per image and tile
    POOL_LIMIT_BYTES = 16 << 30

    def dev_alloc(self, nbytes: int) -> tuple[int, int]:
        """-> (device pointer, capacity).  Capacities are multiples of 2 MiB so that boxes of similar size share buffers."""
        self.drain()
        cap = max(1, (int(nbytes) + (2 << 20) - 1) >> 21) << 21
        pool = self.__dict__.setdefault("_pool", {})
        free = pool.get(cap)
        if free:
            self._pool_bytes -= cap
            return free.pop(), cap
        p = C.c_void_p()
        self.check(self.lib.km_dev_alloc(self.handle, cap, C.byref(p)), "km_dev_alloc")
        return p.value, cap

    def dev_release(self, ptr_: int, cap: int, upload_in_flight: bool = False):
        """Give a buffer back.  Kernels queued on the context that still use it are waited for first; the copy stream only
        when an upload INTO this buffer was never joined (waiting for it unconditionally would serialise the release of pair
        i-1 with the upload of pair i+1 that is travelling under pair i's compute)."""
        if not getattr(self, "handle", None):
            return
        pool = self.__dict__.setdefault("_pool", {})
        held = self.__dict__.setdefault("_pool_bytes", 0)
        self.lib.km_ctx_sync(self.handle)
        if upload_in_flight:
            self.lib.km_upload_wait(self.handle)
        if held + cap <= self.POOL_LIMIT_BYTES:
            pool.setdefault(cap, []).append(ptr_)
            self._pool_bytes = held + cap
        else:
            self.lib.km_dev_free(self.handle, C.c_void_p(ptr_))

    def trim_pool(self):
        for cap, ptrs in self.__dict__.get("_pool", {}).items():
            for p in ptrs:
                self.lib.km_dev_free(self.handle, C.c_void_p(p))
        self._pool, self._pool_bytes = {}, 0

    def run_or_defer(self, fn):
        """Run `fn()` (library calls on this context) now when called on the owning thread, otherwise leave it for that thread."""
        if threading.get_ident() == self.__dict__.get("_owner"):
            self.drain()
            fn()
        else:
            with self._abandoned_lock:
                self._abandoned.append(fn)

    def drain(self):
        """Owning thread: run what finalizers on other threads left behind (released buffers, unused upload tickets)."""
        if not self.__dict__.get("_abandoned"):
            return
        with self._abandoned_lock:
            todo, self._abandoned = self._abandoned, []
        for fn in todo:
            try:
                fn()
            except Exception:  # noqa: BLE001 - a finalizer's failure must not break the caller's call
                pass

    def adopt(self):
        """Declare the calling thread the context's owner (a context handed from the thread that created it to another one)."""
        self._owner = threading.get_ident()
        self.drain()

    def check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.km_last_error(self.handle)
            raise KariosHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "handle", None):
            self._owner = threading.get_ident()      # (whoever closes it has it to itself)
            self.drain()
            # page-locked arrays that outlive the context are released without it (km_host_free(NULL, p)); the staging area the
            # context itself owns goes first, while its streams still exist
            self.__dict__.pop("_kp_staging", None)
            self._alive[0] = False
            self.trim_pool()
            self.lib.km_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover - interpreter shutdown
            pass

    def sync(self):
        self.check(self.lib.km_ctx_sync(self.handle), "km_ctx_sync")

    def flush(self, ticket: int = -1):
        """km_frame_flush: with the "units_pipeline" option a batched submission leaves its tail (LK, frame stage, scores, copy-out) to
        the NEXT submission - call this on the submitting thread when none follows before frame `ticket` is waited for (a no-op when
        that frame's tail is not the deferred one; -1: whichever is)."""
        self.check(self.lib.km_frame_flush(self.handle, int(ticket)), "km_frame_flush")

    def stats(self) -> KltStats:
        s = KltStats()
        self.check(self.lib.km_get_klt_stats(self.handle, C.byref(s)), "km_get_klt_stats")
        return s

    def set_profiling(self, on: bool):
        self.check(self.lib.km_set_profiling(self.handle, int(bool(on))), "km_set_profiling")

    def upload_check_stats(self):
        """-> (uploads checked, uploads whose first consumer saw stale rows) with KARIOS_HIP_UPLOAD_CHECKSUM=1 (csrc/staging.hip)."""
        a, m = C.c_int64(), C.c_int64()
        self.check(self.lib.km_upload_check_stats(self.handle, C.byref(a), C.byref(m)), "km_upload_check_stats")
        return int(a.value), int(m.value)

    def set_option(self, name: str, value: int):
        """Knob of include/karios_hip.h km_set_option, e.g. set_option("fused_eig", 0) or the test knobs "key_cap",
        "stage_cap", "topk_factor", "select_first", "defer" (0 restores a default)."""
        self.check(self.lib.km_set_option(self.handle, name.encode(), int(value)), "km_set_option")
        self.__dict__.setdefault("_options", {})[name] = int(value)

    def set_frame_sink(self, ptr: int | None, nbytes: int = 0, pitch: int = 0) -> None:
        """km_set_frame_sink[_pitch]: every frame block the tile entry points produce from now on is also copied to device memory at
        `ptr` (capacity `nbytes`; a batched submission's unit k at `ptr + k * pitch`, pitch 0 = the block size); None switches it off.
        The current sink is remembered (`frame_sink`): the exact repeat of a flagged unit (`PendingFrame.redo`) runs with the sink
        OFF - by then it belongs to a newer unit."""
        self.check(self.lib.km_set_frame_sink_pitch(self.handle, C.c_void_p(ptr) if ptr else None, int(nbytes) if ptr else 0, int(pitch) if ptr else 0),
                   "km_set_frame_sink_pitch")
        self.__dict__["_frame_sink"] = (int(ptr), int(nbytes), int(pitch)) if ptr else (None, 0, 0)

    @property
    def frame_sink(self) -> tuple:
        return self.__dict__.get("_frame_sink", (None, 0, 0))

    def get_option(self, name: str, default: int = 0) -> int:
        """Last value given to `set_option` (the library's own defaults are not queried)."""
        return self.__dict__.get("_options", {}).get(name, default)

    def phase_info(self) -> tuple[int, float]:
        """(path, margin) of the last phase correlation: path 1 = float32 hand-written FFT, 2 = double precision (hand-written too, k_fft64.hip)."""
        path, margin = C.c_int(), C.c_double()
        self.check(self.lib.km_phase_info(self.handle, C.byref(path), C.byref(margin)), "km_phase_info")
        return path.value, margin.value

    def stage_ms(self) -> dict:
        buf = (C.c_float * 16)()
        n = C.c_int()
        self.check(self.lib.km_get_stage_ms(self.handle, buf, 16, C.byref(n)), "km_get_stage_ms")
        return {self.lib.km_stage_name(i).decode(): float(buf[i]) for i in range(n.value)}


def pinned_empty(shape, dtype, ctx: "Context | None" = None) -> np.ndarray:
    """numpy array over page-locked host memory (km_host_alloc): let GDAL / numpy write the image into it
    (`band.ReadAsArray(buf_obj=arr)`, `np.copyto(arr, img)`) and uploads from it run asynchronously at full PCIe rate.
    The memory is released when the array (and every view of it) is gone."""
    import weakref
    ctx = ctx if ctx is not None else default_context()
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    p = C.c_void_p()
    ctx.check(ctx.lib.km_host_alloc(ctx.handle, max(n, 1), C.byref(p)), "km_host_alloc")
    raw = (C.c_char * max(n, 1)).from_address(p.value)
    # the finalizer may run after the context is gone (interpreter shutdown, an explicit close()): then the block is released
    # without touching the destroyed context
    weakref.finalize(raw, lambda lib=ctx.lib, h=ctx.handle, alive=ctx._alive, q=p.value:
                     lib.km_host_free(h if alive[0] else None, C.c_void_p(q)))
    return np.frombuffer(raw, dtype=dt, count=int(np.prod(shape))).reshape(shape)


_tls = threading.local()


def default_context(device: int | None = None) -> Context:
    """Per-thread context (klt_tracker is called from a thread pool, reference klt.py:526)."""
    if device is None:
        device = int(os.environ.get("KARIOS_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    ctxs = getattr(_tls, "ctxs", None)
    if ctxs is None:
        ctxs = _tls.ctxs = {}
    if device not in ctxs:
        ctxs[device] = Context(device)
    return ctxs[device]
