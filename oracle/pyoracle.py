"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/oracle.h).  The product library never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

MODE_FULL, MODE_SAME, MODE_VALID = 0, 1, 2
SCALE_NONE, SCALE_LIB, SCALE_MY = 0, 1, 2
FFT_REFERENCE, FFT_POW2, FFT_DIRECT, FFT_POW2_CACHED = 0, 1, 2, 3
PREC_F64, PREC_F32 = 0, 1


class OrcPeak(C.Structure):
    _fields_ = [("start", C.c_uint64), ("end", C.c_uint64),
                ("height", C.c_float), ("prominence", C.c_float)]

    def as_tuple(self):
        return (int(self.start), int(self.end), float(self.height), float(self.prominence))


class OrcPolicy(C.Structure):
    """oracle.h orc_policy: the unpinned rules as switches (all zero = the defaults)."""
    _fields_ = [("peak_filter_order", C.c_int), ("distance_rule", C.c_int), ("tail_window", C.c_int),
                ("surrounding_from", C.c_int)]


def policy(peak_filter_order=0, distance_rule=0, tail_window=0, surrounding_from=0) -> OrcPolicy:
    return OrcPolicy(int(peak_filter_order), int(distance_rule), int(tail_window), int(surrounding_from))


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc if missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle.h", "fft_impl.inc")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        f32p = C.POINTER(C.c_float)
        L.orc_pcm_s16_stereo_to_mono.argtypes = [C.POINTER(C.c_int16), C.c_size_t, f32p]
        L.orc_pcm_s16_stereo_to_mono.restype = None
        L.orc_mode_len.argtypes = [C.c_size_t, C.c_size_t, C.c_int]
        L.orc_mode_len.restype = C.c_size_t
        L.orc_mode_start.argtypes = [C.c_size_t, C.c_size_t, C.c_int]
        L.orc_mode_start.restype = C.c_size_t
        L.orc_inv_autocorr.argtypes = [f32p, C.c_size_t]
        L.orc_inv_autocorr.restype = C.c_float
        L.orc_correlate.argtypes = [f32p, C.c_size_t, f32p, C.c_size_t, C.c_int, C.c_int,
                                    C.c_int, C.c_int, f32p, C.c_size_t]
        L.orc_correlate.restype = C.c_size_t
        L.orc_find_peaks.argtypes = [f32p, C.c_size_t, C.c_float, C.c_size_t,
                                     C.POINTER(OrcPeak), C.c_size_t]
        L.orc_find_peaks.restype = C.c_size_t
        L.orc_find_peaks_policy.argtypes = [f32p, C.c_size_t, C.c_float, C.c_size_t, C.POINTER(OrcPolicy),
                                            C.POINTER(OrcPeak), C.c_size_t]
        L.orc_find_peaks_policy.restype = C.c_size_t
        L.orc_calc_chunks_policy.argtypes = [C.c_uint32, f32p, C.c_size_t, f32p, C.c_size_t,
                                             C.c_size_t, C.c_size_t, C.c_float, C.c_size_t, C.c_double,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(OrcPolicy),
                                             C.POINTER(OrcPeak), C.c_size_t]
        L.orc_calc_chunks_policy.restype = C.c_size_t
        L.orc_is_overshadowed.argtypes = [C.POINTER(OrcPeak), C.POINTER(OrcPeak), C.c_uint32, C.c_double]
        L.orc_is_overshadowed.restype = C.c_int
        L.orc_calc_chunks.argtypes = [C.c_uint32, f32p, C.c_size_t, f32p, C.c_size_t,
                                      C.c_size_t, C.c_size_t, C.c_float, C.c_size_t, C.c_double,
                                      C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(OrcPeak), C.c_size_t]
        L.orc_calc_chunks.restype = C.c_size_t
        L.orc_round_samples.argtypes = [C.c_double, C.c_uint32]
        L.orc_round_samples.restype = C.c_size_t
        L.orc_start_nanos.argtypes = [C.c_uint64, C.c_uint32]
        L.orc_start_nanos.restype = C.c_uint64
        L.orc_synth_uniform.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_size_t, C.c_float, f32p]
        L.orc_synth_uniform.restype = None
        _lib = L
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def pcm_s16_stereo_to_mono(interleaved: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(interleaved, dtype=np.int16)
    frames = a.size // 2
    out = np.empty(frames, dtype=np.float32)
    lib().orc_pcm_s16_stereo_to_mono(a.ctypes.data_as(C.POINTER(C.c_int16)), frames,
                                     out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def mode_len(w, s, mode):
    return int(lib().orc_mode_len(w, s, mode))


def inv_autocorr(needle) -> float:
    n, p = _f32(needle)
    return float(lib().orc_inv_autocorr(p, n.size))


def correlate(within, needle, mode=MODE_VALID, scale=SCALE_NONE, fft=FFT_POW2, prec=PREC_F64):
    w, wp = _f32(within)
    n, np_ = _f32(needle)
    ln = mode_len(w.size, n.size, mode)
    out = np.empty(ln, dtype=np.float32)
    got = lib().orc_correlate(wp, w.size, np_, n.size, mode, scale, fft, prec,
                              out.ctypes.data_as(C.POINTER(C.c_float)), ln)
    if got != ln:
        raise RuntimeError("orc_correlate failed")
    return out


def find_peaks(y, min_prominence=0.0, min_distance=0, cap=None, pol: OrcPolicy | None = None):
    a, ap = _f32(y)
    cap = cap or max(16, a.size)
    buf = (OrcPeak * cap)()
    n = lib().orc_find_peaks_policy(ap, a.size, float(min_prominence), int(min_distance),
                                    C.byref(pol) if pol is not None else None, buf, cap)
    return [buf[i].as_tuple() for i in range(min(n, cap))]


def is_overshadowed(element, other, sr, max_distance_s) -> bool:
    e = OrcPeak(*element)
    o = OrcPeak(*other) if other is not None else None
    return bool(lib().orc_is_overshadowed(C.byref(e), C.byref(o) if o is not None else None,
                                          sr, float(max_distance_s)))


def calc_chunks(sr, haystack, needle, chunk, overlap, min_prominence, min_distance,
                overshadow_distance_s, scale=SCALE_LIB, fft=FFT_POW2, prec=PREC_F64,
                threads=1, cap=4096, pol: OrcPolicy | None = None):
    h, hp = _f32(haystack)
    n, np_ = _f32(needle)
    buf = (OrcPeak * cap)()
    cnt = lib().orc_calc_chunks_policy(sr, hp, h.size, np_, n.size, chunk, overlap,
                                       float(min_prominence), int(min_distance),
                                       float(overshadow_distance_s), scale, fft, prec, threads,
                                       C.byref(pol) if pol is not None else None, buf, cap)
    if cnt == C.c_size_t(-1).value:
        raise RuntimeError("orc_calc_chunks failed")
    return [buf[i].as_tuple() for i in range(min(cnt, cap))]


def round_samples(seconds, sr):
    return int(lib().orc_round_samples(float(seconds), sr))


def start_nanos(start, sr):
    return int(lib().orc_start_nanos(start, sr))


def synth_uniform(seed, stream, first, n, amp=0.25) -> np.ndarray:
    out = np.empty(n, dtype=np.float32)
    lib().orc_synth_uniform(seed, stream, first, n, amp, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out
