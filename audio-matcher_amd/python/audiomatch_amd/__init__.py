"""ctypes host binding of libaudiomatch_amd.so (MI355X / gfx950).

Mirrors the reference's matcher interface (src/matcher/audio_matcher.rs):

    LibConvolve::new(sample)                 -> HipConvolve(sample)
    algo.inverse_sample_auto_correlation()   -> HipConvolve.inverse_sample_auto_correlation()
    algo.correlate_with_sample(w, mode, sc)  -> HipConvolve.correlate_with_sample(w, mode, scale)
    calc_chunks(sr, samples, &algo, scale, config)
                                             -> calc_chunks(sr, samples, algo, scale, config)

Every call goes through the C ABI of include/audiomatch.h; there is no Python
or CPU implementation behind it.  Importing this module fails loudly when the
HIP library is missing or cannot be loaded.
"""
from __future__ import annotations

import ctypes as C
import struct
import enum
import os
from dataclasses import dataclass

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.abspath(os.path.join(_PKG, "..", ".."))
LIB_PATH = os.path.join(_ROOT, "libaudiomatch_amd.so")

AM_OK, AM_ERR_INVALID_ARG, AM_ERR_CAPACITY, AM_ERR_HIP, AM_ERR_NO_DEVICE, \
    AM_ERR_PEAK_OVERFLOW, AM_ERR_OOM = range(7)
AM_MAX_PEAKS_PER_CHUNK = 1024


class Fmt(enum.IntEnum):           # sample format of a haystack buffer (AM_FMT_*)
    F32_MONO = 0
    S16_STEREO = 1                 # interleaved i16 stereo frames (mp3_reader.rs:26-37)


class Mode(enum.IntEnum):          # audio_matcher.rs:55-59
    Full = 0
    Same = 1
    Valid = 2


class Scale(enum.IntEnum):
    NONE = 0
    LIB = 1                        # LibConvolve, production (audio_matcher.rs:306-308)
    MY = 2                         # MyConvolve (audio_matcher.rs:442-448)


class AmPeak(C.Structure):
    _fields_ = [("start", C.c_uint64), ("end", C.c_uint64),
                ("height", C.c_float), ("prominence", C.c_float)]


class AmMatchParams(C.Structure):
    _fields_ = [("sr", C.c_uint32), ("chunk", C.c_uint64), ("overlap", C.c_uint64),
                ("min_prominence", C.c_float), ("min_distance", C.c_uint64),
                ("overshadow_distance_s", C.c_double), ("scale", C.c_int)]


class AudioMatchError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"audiomatch error {code}: {msg}")
        self.code = code


# every symbol include/audiomatch.h declares: (name, restype, argtypes)
_f32p = C.POINTER(C.c_float)
_SIGNATURES = {
    "am_abi_version": (C.c_int, []),
    "am_last_error_string": (C.c_char_p, []),
    "am_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "am_shutdown": (C.c_int, []),
    "am_needle_create": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "am_needle_create_device": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "am_needle_destroy": (None, [C.c_void_p]),
    "am_needle_len": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "am_needle_inv_autocorr": (C.c_int, [C.c_void_p, _f32p]),
    "am_correlate_len": (C.c_int, [C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_size_t)]),
    "am_correlate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                               C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "am_correlate_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                      C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "am_match": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(AmMatchParams),
                           C.POINTER(AmPeak), C.c_size_t, C.POINTER(C.c_size_t)]),
    "am_match_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(AmMatchParams),
                                  C.POINTER(AmPeak), C.c_size_t, C.POINTER(C.c_size_t)]),
    "am_match_batch_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                        C.c_size_t, C.POINTER(AmMatchParams), C.POINTER(AmPeak),
                                        C.c_size_t, C.POINTER(C.c_size_t)]),
    "am_match_multi_device": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.POINTER(AmMatchParams), C.POINTER(AmPeak), C.c_size_t,
                                        C.POINTER(C.c_size_t)]),
    "am_needle_create_pcm16": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "am_match_pcm16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(AmMatchParams),
                                 C.POINTER(AmPeak), C.c_size_t, C.POINTER(C.c_size_t)]),
    "am_match_pcm16_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(AmMatchParams),
                                        C.POINTER(AmPeak), C.c_size_t, C.POINTER(C.c_size_t)]),
    "am_match_pcm16_batch_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                              C.c_size_t, C.POINTER(AmMatchParams), C.POINTER(AmPeak),
                                              C.c_size_t, C.POINTER(C.c_size_t)]),
    "am_find_peaks": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_float, C.c_uint64,
                                C.POINTER(AmPeak), C.c_size_t, C.POINTER(C.c_size_t)]),
    "am_pcm_s16_stereo_to_mono": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "am_pcm_s16_stereo_to_mono_device": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "am_device_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "am_device_free": (C.c_int, [C.c_int, C.c_void_p]),
    "am_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "am_host_free": (C.c_int, [C.c_void_p]),
    "am_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "am_host_unregister": (C.c_int, [C.c_void_p]),
    "am_memcpy_h2d": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "am_memcpy_d2h": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "am_device_synchronize": (C.c_int, [C.c_int]),
    "am_synth_uniform_device": (C.c_int, [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64,
                                          C.c_size_t, C.c_float]),
    "am_axpy_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float]),
    "am_synth_pcm16_stereo_device": (C.c_int, [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64,
                                               C.c_size_t, C.c_float]),
    "am_add_pcm16_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "am_set_progress_callback": (C.c_int, [C.c_void_p, C.c_void_p]),
    "am_set_chunk_progress_callback": (C.c_int, [C.c_void_p, C.c_void_p]),
    "am_needle_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_longlong]),
    "am_needle_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong)]),
    "am_shard_plan": (C.c_int, [C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t),
                                C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "am_pool_create": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.c_size_t, C.POINTER(C.c_void_p)]),
    "am_pool_destroy": (None, [C.c_void_p]),
    "am_pool_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "am_pool_slot": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "am_pool_match_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t,
                                      C.POINTER(AmMatchParams), C.POINTER(AmPeak), C.c_size_t,
                                      C.POINTER(C.c_size_t)]),
    "am_pool_match_batch_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t,
                                             C.POINTER(AmMatchParams), C.POINTER(AmPeak), C.c_size_t,
                                             C.POINTER(C.c_size_t)]),
    "am_match_multi_batch_device": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                              C.c_size_t, C.c_int, C.POINTER(AmMatchParams), C.POINTER(AmPeak), C.c_size_t,
                                              C.POINTER(C.c_size_t)]),
    "am_pool_match_batch_pcm16": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t,
                                            C.POINTER(AmMatchParams), C.POINTER(AmPeak), C.c_size_t,
                                            C.POINTER(C.c_size_t)]),
    "am_pool_match_batch_pcm16_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t,
                                                   C.POINTER(AmMatchParams), C.POINTER(AmPeak), C.c_size_t,
                                                   C.POINTER(C.c_size_t)]),
    "am_pool_create_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.POINTER(C.c_int), C.c_size_t,
                                       C.POINTER(C.c_void_p)]),
    "am_pool_needle_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "am_pool_match_multi_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int,
                                            C.POINTER(AmMatchParams), C.POINTER(AmPeak), C.c_size_t,
                                            C.POINTER(C.c_size_t)]),
    "am_pool_match_multi_batch_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int,
                                                   C.POINTER(AmMatchParams), C.POINTER(AmPeak), C.c_size_t,
                                                   C.POINTER(C.c_size_t)]),
    "am_long_plan": (C.c_int, [C.c_size_t, C.c_size_t, C.POINTER(AmMatchParams), C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t),
                               C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "am_match_part_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(AmMatchParams), C.c_size_t,
                                       C.c_uint64, C.POINTER(AmPeak), C.c_size_t, C.POINTER(C.c_size_t)]),
    "am_merge_peaks": (C.c_int, [C.POINTER(AmMatchParams), C.POINTER(AmPeak), C.c_size_t, C.POINTER(AmPeak), C.c_size_t,
                                 C.POINTER(C.c_size_t)]),
    "am_pool_match_long": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(AmMatchParams), C.POINTER(AmPeak),
                                     C.c_size_t, C.POINTER(C.c_size_t)]),
    "am_pool_match_long_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_int, C.POINTER(AmMatchParams),
                                            C.POINTER(AmPeak), C.c_size_t, C.POINTER(C.c_size_t)]),
    "am_match_stream_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.POINTER(AmMatchParams), C.POINTER(C.c_void_p)]),
    "am_match_stream_push": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "am_match_stream_finish": (C.c_int, [C.c_void_p, C.POINTER(AmPeak), C.c_size_t, C.POINTER(C.c_size_t)]),
    "am_match_stream_destroy": (None, [C.c_void_p]),
    "am_debug_column_bench": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "am_profile_enable": (C.c_int, [C.c_int, C.c_int]),
    "am_profile_reset": (C.c_int, [C.c_int]),
    "am_profile_query": (C.c_int, [C.c_int, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "am_set_option": (C.c_int, [C.c_char_p, C.c_longlong]),
    "am_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_longlong)]),
}


def declared_symbols():
    return sorted(_SIGNATURES)


_lib = None


def lib():
    """Load the HIP library (no fallback: raises if it is missing)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python audio-matcher_amd/build.py` "
                "(__graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc: int):
    if rc != AM_OK:
        msg = lib().am_last_error_string()
        raise AudioMatchError(rc, msg.decode() if msg else "")


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().am_device_count(C.byref(n))
    return n.value if rc == AM_OK else 0


def set_option(key: str, value: int):
    _check(lib().am_set_option(key.encode(), int(value)))


def get_option(key: str) -> int:
    v = C.c_longlong(0)
    _check(lib().am_get_option(key.encode(), C.byref(v)))
    return v.value


# ---------------------------------------------------------------------------
class DeviceBuffer:
    """A raw HBM allocation owned through the C ABI (am_device_malloc)."""

    def __init__(self, device: int, nbytes: int):
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        _check(lib().am_device_malloc(device, max(self.nbytes, 4), C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, device: int, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        buf = cls(device, a.nbytes)
        if a.nbytes:
            _check(lib().am_memcpy_h2d(device, buf.ptr, a.ctypes.data, a.nbytes))
        return buf

    def to_numpy(self, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        if out.nbytes:
            _check(lib().am_memcpy_d2h(self.device, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().am_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray:
    """A numpy array in pinned host memory (am_host_alloc): what a decoder would write its output into."""

    def __init__(self, shape, dtype):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        _check(lib().am_host_alloc(max(self.nbytes, 4), C.byref(p)))
        self.ptr = p.value
        self.array = np.frombuffer((C.c_char * self.nbytes).from_address(self.ptr), dtype=dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            lib().am_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class registered:
    """with registered(array): the array's memory is pinned for the duration (am_host_register)."""

    def __init__(self, a: np.ndarray):
        self.a = a

    def __enter__(self):
        _check(lib().am_host_register(self.a.ctypes.data, self.a.nbytes))
        return self.a

    def __exit__(self, *exc):
        _check(lib().am_host_unregister(self.a.ctypes.data))


def synth_uniform_device(device: int, n: int, seed: int, stream: int, first: int = 0,
                         amp: float = 0.25) -> DeviceBuffer:
    buf = DeviceBuffer(device, n * 4)
    _check(lib().am_synth_uniform_device(device, buf.ptr, seed, stream, first, n, amp))
    return buf


def synth_pcm16_stereo_device(device: int, frames: int, seed: int, stream: int, first: int = 0,
                              amp: float = 0.25) -> DeviceBuffer:
    """Interleaved i16 stereo frames of the synthetic signal (left = stream, right = stream + 5000)."""
    buf = DeviceBuffer(device, frames * 4)
    _check(lib().am_synth_pcm16_stereo_device(device, buf.ptr, seed, stream, first, frames, amp))
    return buf


def add_pcm16_device(device: int, dst: DeviceBuffer, dst_frame: int, src_ptr: int, frames: int):
    _check(lib().am_add_pcm16_device(device, dst.ptr + 4 * dst_frame, src_ptr, frames))


def axpy_device(device: int, dst: DeviceBuffer, dst_offset: int, src_ptr: int, n: int, gain: float = 1.0):
    _check(lib().am_axpy_device(device, dst.ptr + 4 * dst_offset, src_ptr, n, gain))


def pcm_s16_stereo_to_mono(interleaved: np.ndarray, device: int = 0) -> np.ndarray:
    """mp3_reader.rs:28-37 down-mix on the GPU."""
    a = np.ascontiguousarray(interleaved, dtype=np.int16)
    frames = a.size // 2
    out = np.empty(frames, dtype=np.float32)
    _check(lib().am_pcm_s16_stereo_to_mono(device, a.ctypes.data, frames, out.ctypes.data))
    return out


def find_peaks(y_data, min_prominence: float, min_distance: int = 0, device: int = 0, cap: int = 65536):
    """audio_matcher.rs:221-230 on the GPU; peaks by descending height."""
    a = np.ascontiguousarray(y_data, dtype=np.float32)
    buf = (AmPeak * cap)()
    n = C.c_size_t(0)
    _check(lib().am_find_peaks(device, a.ctypes.data, a.size, float(min_prominence), int(min_distance),
                               buf, cap, C.byref(n)))
    return [Peak(int(b.start), int(b.end), float(b.height), float(b.prominence)) for b in buf[:n.value]]


# ---------------------------------------------------------------------------
@dataclass
class Peak:
    """find_peaks::Peak<f32> as used downstream (position, height, prominence)."""
    start: int
    end: int
    height: float
    prominence: float

    @property
    def position(self):
        return range(self.start, self.end)


@dataclass
class Config:
    """audio_matcher.rs:25-53 (Config + PeakConfig), durations in seconds."""
    chunk_size_s: float = 60.0          # matcher/args.rs:70-72
    overlap_length_s: float = 0.0       # Config::from_args sets it to the snippet duration (:41)
    distance_s: float = 8 * 60.0        # matcher/args.rs:73-76
    prominence: float = 13.0 / 100.0    # args.prominence / 100 (:44), default 13 (args.rs:19)

    def params(self, sr: int, scale: int) -> AmMatchParams:
        def rnd(x):                     # f64::round: half away from zero (audio_matcher.rs:99-100)
            return int(np.floor(x + 0.5))
        return AmMatchParams(
            sr=sr, chunk=rnd(self.chunk_size_s * sr), overlap=rnd(self.overlap_length_s * sr),
            min_prominence=self.prominence,
            min_distance=int(self.distance_s) * sr,          # distance.as_secs() as usize * sr (:228)
            overshadow_distance_s=self.distance_s, scale=int(scale))


class HipConvolve:
    """CorrelateAlgo<f32> (audio_matcher.rs:65-76) backed by the HIP library."""

    def __init__(self, sample_data, device: int = 0):
        a = np.ascontiguousarray(sample_data, dtype=np.float32)
        self.device = device
        self._h = C.c_void_p()
        _check(lib().am_needle_create(device, a.ctypes.data, a.size, C.byref(self._h)))
        self.sample_len = int(a.size)

    @classmethod
    def from_device(cls, device: int, ptr: int, n: int) -> "HipConvolve":
        self = cls.__new__(cls)
        self.device = device
        self._h = C.c_void_p()
        _check(lib().am_needle_create_device(device, ptr, n, C.byref(self._h)))
        self.sample_len = int(n)
        return self

    @classmethod
    def from_pcm16(cls, interleaved, device: int = 0) -> "HipConvolve":
        """Needle given as interleaved i16 stereo frames (down-mixed on the GPU)."""
        a = np.ascontiguousarray(interleaved, dtype=np.int16)
        self = cls.__new__(cls)
        self.device = device
        self._h = C.c_void_p()
        _check(lib().am_needle_create_pcm16(device, a.ctypes.data, a.size // 2, C.byref(self._h)))
        self.sample_len = int(a.size // 2)
        return self

    def match_pcm16(self, interleaved, params: AmMatchParams, cap: int = 4096):
        a = np.ascontiguousarray(interleaved, dtype=np.int16)
        buf = (AmPeak * cap)()
        n = C.c_size_t(0)
        _check(lib().am_match_pcm16(self._h, a.ctypes.data, a.size // 2, C.byref(params), buf, cap, C.byref(n)))
        return [Peak(int(b.start), int(b.end), float(b.height), float(b.prominence)) for b in buf[:n.value]]

    def set_option(self, key: str, value: int):
        """Per-handle "log_n" / "half_pipeline" (-1 = follow the process default)."""
        _check(lib().am_needle_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key: str) -> int:
        v = C.c_longlong(0)
        _check(lib().am_needle_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    def match_pcm16_batch_device(self, ptrs, frames, params: AmMatchParams, cap_per_hay: int = 256):
        k = len(ptrs)
        arr_p = (C.c_void_p * k)(*ptrs)
        arr_l = (C.c_size_t * k)(*frames)
        buf = (AmPeak * (cap_per_hay * k))()
        counts = (C.c_size_t * k)()
        _check(lib().am_match_pcm16_batch_device(self._h, arr_p, arr_l, k, C.byref(params), buf,
                                                 cap_per_hay, counts))
        return _split_batch(buf, counts, k, cap_per_hay)

    def match_pcm16_device(self, ptr: int, frames: int, params: AmMatchParams, cap: int = 4096):
        buf = (AmPeak * cap)()
        n = C.c_size_t(0)
        _check(lib().am_match_pcm16_device(self._h, ptr, frames, C.byref(params), buf, cap, C.byref(n)))
        return [Peak(int(b.start), int(b.end), float(b.height), float(b.prominence)) for b in buf[:n.value]]

    def close(self):
        if getattr(self, "_h", None):
            lib().am_needle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def inverse_sample_auto_correlation(self) -> float:
        v = C.c_float(0)
        _check(lib().am_needle_inv_autocorr(self._h, C.byref(v)))
        return v.value

    def correlate_with_sample(self, within, mode: Mode = Mode.Valid, scale=False) -> np.ndarray:
        """scale: bool as in the reference (True = production/LibConvolve) or a Scale value."""
        w = np.ascontiguousarray(within, dtype=np.float32)
        sc = int(Scale.LIB if scale is True else Scale.NONE if scale is False else scale)
        n = C.c_size_t(0)
        _check(lib().am_correlate_len(w.size, self.sample_len, int(mode), C.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        _check(lib().am_correlate(self._h, w.ctypes.data, w.size, int(mode), sc,
                                  out.ctypes.data, out.size, C.byref(n)))
        return out

    # -- level 2 --
    def match(self, haystack, params: AmMatchParams, cap: int = 4096):
        h = np.ascontiguousarray(haystack, dtype=np.float32)
        buf = (AmPeak * cap)()
        n = C.c_size_t(0)
        _check(lib().am_match(self._h, h.ctypes.data, h.size, C.byref(params), buf, cap, C.byref(n)))
        return [Peak(int(b.start), int(b.end), float(b.height), float(b.prominence)) for b in buf[:n.value]]

    def match_device(self, ptr: int, length: int, params: AmMatchParams, cap: int = 4096):
        buf = (AmPeak * cap)()
        n = C.c_size_t(0)
        _check(lib().am_match_device(self._h, ptr, length, C.byref(params), buf, cap, C.byref(n)))
        return [Peak(int(b.start), int(b.end), float(b.height), float(b.prominence)) for b in buf[:n.value]]

    def match_batch_device(self, ptrs, lengths, params: AmMatchParams, cap_per_hay: int = 256):
        k = len(ptrs)
        arr_p = (C.c_void_p * k)(*ptrs)
        arr_l = (C.c_size_t * k)(*lengths)
        buf = (AmPeak * (cap_per_hay * k))()
        counts = (C.c_size_t * k)()
        _check(lib().am_match_batch_device(self._h, arr_p, arr_l, k, C.byref(params), buf,
                                           cap_per_hay, counts))
        return _split_batch(buf, counts, k, cap_per_hay)


class MatchStream:
    """calc_chunks on a haystack that arrives in pieces (am_match_stream_*): the reference's lazy
    sample iterator (audio_matcher.rs:88-97, mp3_reader.rs:13-41)."""

    def __init__(self, algo: "HipConvolve", params: AmMatchParams, expected_len: int = 0, fmt: int = Fmt.F32_MONO):
        self._algo = algo                      # keeps the needle handle alive
        self.fmt = int(fmt)
        self._s = C.c_void_p()
        _check(lib().am_match_stream_begin(algo._h, self.fmt, int(expected_len), C.byref(params), C.byref(self._s)))

    def push(self, samples):
        a = np.ascontiguousarray(samples, dtype=np.float32 if self.fmt == Fmt.F32_MONO else np.int16)
        n = a.size if self.fmt == Fmt.F32_MONO else a.size // 2
        _check(lib().am_match_stream_push(self._s, a.ctypes.data, n))

    def finish(self, cap: int = 4096):
        buf = (AmPeak * cap)()
        n = C.c_size_t(0)
        _check(lib().am_match_stream_finish(self._s, buf, cap, C.byref(n)))
        return [Peak(int(b.start), int(b.end), float(b.height), float(b.prominence)) for b in buf[:n.value]]

    def close(self):
        if getattr(self, "_s", None):
            lib().am_match_stream_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# am_peak records straight out of the result buffer's bytes (struct.iter_unpack): a third of the time of reading the
# four fields of every ctypes structure one by one -- this runs once per call, in the caller's timed loop
_PEAK_REC = struct.Struct("<QQff")
assert _PEAK_REC.size == C.sizeof(AmPeak)


def _peaks_at(buf, first: int, n: int):
    mv = memoryview(buf).cast("B")
    return [Peak(*t) for t in _PEAK_REC.iter_unpack(mv[first * _PEAK_REC.size:(first + n) * _PEAK_REC.size])]


def _split_batch(buf, counts, k: int, cap: int):
    return [_peaks_at(buf, i * cap, counts[i]) for i in range(k)]


def _peaks(buf, n):
    return _peaks_at(buf, 0, n)


def long_plan(length: int, needle_len: int, params: AmMatchParams, n_parts: int, part: int):
    """am_long_plan: (first_window, n_windows, first_sample, n_samples) of part `part` of one long haystack."""
    v = [C.c_size_t(0) for _ in range(4)]
    _check(lib().am_long_plan(length, needle_len, C.byref(params), n_parts, part, *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def match_part_device(algo: "HipConvolve", ptr: int, n_samples: int, params: AmMatchParams, n_windows: int, first_sample: int,
                      fmt: int = Fmt.F32_MONO, cap: int = 4096):
    """am_match_part_device: the unmerged peaks of one part (window order, positions in the whole haystack)."""
    buf = (AmPeak * cap)()
    n = C.c_size_t(0)
    _check(lib().am_match_part_device(algo._h, ptr, n_samples, int(fmt), C.byref(params), n_windows, first_sample, buf, cap, C.byref(n)))
    return _peaks(buf, n.value)


def merge_peaks(params: AmMatchParams, peaks, cap: int = 4096):
    """am_merge_peaks: sort by start + the overshadow filter (audio_matcher.rs:132-160)."""
    k = len(peaks)
    src = (AmPeak * max(1, k))(*[AmPeak(q.start, q.end, q.height, q.prominence) for q in peaks])
    buf = (AmPeak * cap)()
    n = C.c_size_t(0)
    _check(lib().am_merge_peaks(C.byref(params), src, k, buf, cap, C.byref(n)))
    return _peaks(buf, n.value)


def shard_plan(n_items: int, n_shards: int, shard: int):
    """am_shard_plan: (first, stride, count) of the items shard `shard` owns (k mod n_shards)."""
    a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    _check(lib().am_shard_plan(n_items, n_shards, shard, C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


class Pool:
    """The haystack loop of matcher::run (matcher/mod.rs:42-87) over several GPUs: the needle
    replicated per device, haystack k matched on slot k mod n (am_pool_*)."""

    def __init__(self, sample_data, devices=None):
        a = np.ascontiguousarray(sample_data, dtype=np.float32)
        self._p = C.c_void_p()
        if devices is None:
            _check(lib().am_pool_create(a.ctypes.data, a.size, None, 0, C.byref(self._p)))
        else:
            arr = (C.c_int * len(devices))(*devices)
            _check(lib().am_pool_create(a.ctypes.data, a.size, arr, len(devices), C.byref(self._p)))
        n = C.c_size_t(0)
        _check(lib().am_pool_size(self._p, C.byref(n)))
        self.size = n.value

    def device_of(self, slot: int) -> int:
        d = C.c_int(0)
        _check(lib().am_pool_slot(self._p, slot, C.byref(d), None))
        return d.value

    def match_batch(self, haystacks, params: AmMatchParams, cap_per_hay: int = 256):
        """Host haystacks (numpy f32 arrays; None = skipped)."""
        hs = [None if h is None else np.ascontiguousarray(h, dtype=np.float32) for h in haystacks]
        k = len(hs)
        arr_p = (C.c_void_p * k)(*[None if h is None else h.ctypes.data for h in hs])
        arr_l = (C.c_size_t * k)(*[0 if h is None else h.size for h in hs])
        buf = (AmPeak * max(1, cap_per_hay * k))()
        counts = (C.c_size_t * max(1, k))()
        _check(lib().am_pool_match_batch(self._p, arr_p, arr_l, k, C.byref(params), buf, cap_per_hay, counts))
        return _split_batch(buf, counts, k, cap_per_hay)

    def match_batch_device(self, ptrs, lengths, params: AmMatchParams, cap_per_hay: int = 256):
        """Resident haystacks: ptrs[k] must live on the device of slot k mod size."""
        k = len(ptrs)
        arr_p = (C.c_void_p * k)(*ptrs)
        arr_l = (C.c_size_t * k)(*lengths)
        buf = (AmPeak * max(1, cap_per_hay * k))()
        counts = (C.c_size_t * max(1, k))()
        _check(lib().am_pool_match_batch_device(self._p, arr_p, arr_l, k, C.byref(params), buf, cap_per_hay, counts))
        return _split_batch(buf, counts, k, cap_per_hay)

    def match_long(self, haystack, params: AmMatchParams, fmt: int = Fmt.F32_MONO, cap: int = 4096):
        """ONE long host haystack split over the pool's slots by window ranges (am_pool_match_long)."""
        a = np.ascontiguousarray(haystack, dtype=np.float32 if int(fmt) == Fmt.F32_MONO else np.int16)
        n = a.size if int(fmt) == Fmt.F32_MONO else a.size // 2
        buf = (AmPeak * cap)()
        cnt = C.c_size_t(0)
        _check(lib().am_pool_match_long(self._p, a.ctypes.data, n, int(fmt), C.byref(params), buf, cap, C.byref(cnt)))
        return _peaks(buf, cnt.value)

    def match_long_device(self, part_ptrs, length: int, params: AmMatchParams, fmt: int = Fmt.F32_MONO, cap: int = 4096):
        """The same with resident parts: part_ptrs[i] = the samples of part i (long_plan) on slot i's device."""
        arr = (C.c_void_p * len(part_ptrs))(*part_ptrs)
        buf = (AmPeak * cap)()
        cnt = C.c_size_t(0)
        _check(lib().am_pool_match_long_device(self._p, arr, length, int(fmt), C.byref(params), buf, cap, C.byref(cnt)))
        return _peaks(buf, cnt.value)

    def match_batch_pcm16(self, haystacks, params: AmMatchParams, cap_per_hay: int = 256):
        """Host haystacks as interleaved i16 stereo arrays (2 * frames values; None = skipped)."""
        hs = [None if h is None else np.ascontiguousarray(h, dtype=np.int16) for h in haystacks]
        k = len(hs)
        arr_p = (C.c_void_p * k)(*[None if h is None else h.ctypes.data for h in hs])
        arr_l = (C.c_size_t * k)(*[0 if h is None else h.size // 2 for h in hs])
        buf = (AmPeak * max(1, cap_per_hay * k))()
        counts = (C.c_size_t * max(1, k))()
        _check(lib().am_pool_match_batch_pcm16(self._p, arr_p, arr_l, k, C.byref(params), buf, cap_per_hay, counts))
        return _split_batch(buf, counts, k, cap_per_hay)

    def match_batch_pcm16_device(self, ptrs, frames, params: AmMatchParams, cap_per_hay: int = 256):
        k = len(ptrs)
        arr_p = (C.c_void_p * k)(*ptrs)
        arr_l = (C.c_size_t * k)(*frames)
        buf = (AmPeak * max(1, cap_per_hay * k))()
        counts = (C.c_size_t * max(1, k))()
        _check(lib().am_pool_match_batch_pcm16_device(self._p, arr_p, arr_l, k, C.byref(params), buf, cap_per_hay, counts))
        return _split_batch(buf, counts, k, cap_per_hay)

    def close(self):
        if getattr(self, "_p", None):
            lib().am_pool_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _split_pairs(buf, counts, n_hay: int, nn: int, cap: int):
    """[k][j] = peaks of haystack k against needle j (slot k * nn + j)."""
    return [[_peaks_at(buf, (k * nn + j) * cap, counts[k * nn + j]) for j in range(nn)] for k in range(n_hay)]


def match_multi_batch_device(algos, ptrs, lengths, params: AmMatchParams, fmt: int = Fmt.F32_MONO, cap_per_pair: int = 64):
    """Several equal-length needles against a batch of resident haystacks (am_match_multi_batch_device):
    result [k][j] = haystack k against needle j."""
    nn, k = len(algos), len(ptrs)
    handles = (C.c_void_p * nn)(*[a._h for a in algos])
    arr_p = (C.c_void_p * k)(*ptrs)
    arr_l = (C.c_size_t * k)(*lengths)
    buf = (AmPeak * max(1, cap_per_pair * k * nn))()
    counts = (C.c_size_t * max(1, k * nn))()
    _check(lib().am_match_multi_batch_device(handles, nn, arr_p, arr_l, k, int(fmt), C.byref(params), buf, cap_per_pair, counts))
    return _split_pairs(buf, counts, k, nn, cap_per_pair)


class MultiPool:
    """Several equal-length needles replicated on every listed device (am_pool_create_multi): the file
    loop of matcher::run around N snippets, haystack k on slot k mod n."""

    def __init__(self, samples, devices=None):
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in samples]
        assert arrs and all(a.size == arrs[0].size for a in arrs)
        self._keep = arrs
        self.n_needles = len(arrs)
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        self._p = C.c_void_p()
        if devices is None:
            _check(lib().am_pool_create_multi(ptrs, len(arrs), arrs[0].size, None, 0, C.byref(self._p)))
        else:
            arr = (C.c_int * len(devices))(*devices)
            _check(lib().am_pool_create_multi(ptrs, len(arrs), arrs[0].size, arr, len(devices), C.byref(self._p)))
        n = C.c_size_t(0)
        _check(lib().am_pool_size(self._p, C.byref(n)))
        self.size = n.value

    def _run(self, fn, ptrs, lens, fmt, params, cap_per_pair):
        k, nn = len(ptrs), self.n_needles
        arr_p = (C.c_void_p * max(1, k))(*ptrs)
        arr_l = (C.c_size_t * max(1, k))(*lens)
        buf = (AmPeak * max(1, cap_per_pair * k * nn))()
        counts = (C.c_size_t * max(1, k * nn))()
        _check(fn(self._p, arr_p, arr_l, k, int(fmt), C.byref(params), buf, cap_per_pair, counts))
        return _split_pairs(buf, counts, k, nn, cap_per_pair)

    def match_batch(self, haystacks, params: AmMatchParams, fmt: int = Fmt.F32_MONO, cap_per_pair: int = 64):
        """Host haystacks: f32 arrays, or interleaved i16 stereo arrays with fmt = Fmt.S16_STEREO."""
        dt = np.float32 if int(fmt) == Fmt.F32_MONO else np.int16
        per = 1 if int(fmt) == Fmt.F32_MONO else 2
        hs = [None if h is None else np.ascontiguousarray(h, dtype=dt) for h in haystacks]
        return self._run(lib().am_pool_match_multi_batch, [None if h is None else h.ctypes.data for h in hs],
                         [0 if h is None else h.size // per for h in hs], fmt, params, cap_per_pair)

    def match_batch_device(self, ptrs, lengths, params: AmMatchParams, fmt: int = Fmt.F32_MONO, cap_per_pair: int = 64):
        return self._run(lib().am_pool_match_multi_batch_device, ptrs, lengths, fmt, params, cap_per_pair)

    def close(self):
        if getattr(self, "_p", None):
            lib().am_pool_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def match_multi_device(algos, ptr: int, length: int, params: AmMatchParams, cap_per_needle: int = 256):
    """Several equal-length needles against one resident haystack (shared forward pass)."""
    k = len(algos)
    handles = (C.c_void_p * k)(*[a._h for a in algos])
    buf = (AmPeak * (cap_per_needle * k))()
    counts = (C.c_size_t * k)()
    _check(lib().am_match_multi_device(handles, k, ptr, length, C.byref(params), buf, cap_per_needle, counts))
    return [[Peak(int(b.start), int(b.end), float(b.height), float(b.prominence))
             for b in buf[i * cap_per_needle: i * cap_per_needle + counts[i]]] for i in range(k)]


def calc_chunks(sr: int, m_samples, algo_with_sample: HipConvolve, scale: bool, config: Config):
    """audio_matcher.rs:88-141 on the GPU: returns peaks sorted by position.start."""
    params = config.params(sr, Scale.LIB if scale else Scale.NONE)
    return algo_with_sample.match(m_samples, params)


PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t)
_progress_keepalive = None


def set_progress_callback(fn):
    """fn(haystack_index, stage, n_chunks): stage 0 = queued, 1 = finished
    (the f1/f2 callbacks of audio_matcher.rs:102-117, 129); None clears it."""
    global _progress_keepalive
    if fn is None:
        _progress_keepalive = None
        _check(lib().am_set_progress_callback(None, None))
        return
    cb = PROGRESS_FN(lambda user, k, stage, n: fn(int(k), int(stage), int(n)))
    _progress_keepalive = cb
    _check(lib().am_set_progress_callback(C.cast(cb, C.c_void_p), None))


CHUNK_PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int)
_chunk_progress_keepalive = None


def set_chunk_progress_callback(fn):
    """fn(haystack_index, chunk_index, n_chunks, stage): the per-chunk f1/f2 callbacks of
    audio_matcher.rs:116-117, 129 (stage 0 = picked up, 1 = done); None clears it."""
    global _chunk_progress_keepalive
    if fn is None:
        _check(lib().am_set_chunk_progress_callback(None, None))
        _chunk_progress_keepalive = None
        return
    cb = CHUNK_PROGRESS_FN(lambda user, k, i, n, stage: fn(int(k), int(i), int(n), int(stage)))
    _check(lib().am_set_chunk_progress_callback(C.cast(cb, C.c_void_p), None))
    _chunk_progress_keepalive = cb


class Profile:
    """HIP-event timing of the pipeline kernels (am_profile_*)."""

    def __init__(self, device: int = 0):
        self.device = device

    def __enter__(self):
        _check(lib().am_profile_reset(self.device))
        _check(lib().am_profile_enable(self.device, 1))
        return self

    def __exit__(self, *exc):
        _check(lib().am_profile_enable(self.device, 0))

    def query(self, kernel: str):
        ms, n = C.c_double(0), C.c_uint64(0)
        _check(lib().am_profile_query(self.device, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value
