"""TEST INFRASTRUCTURE ONLY: ctypes front end of the CPU oracle (liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this package, and only as the checker.  The product library never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

PIN_HASH_BASIS = 1469598103934665603


def build(force=False):
    src = os.path.join(_HERE, "huffman_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _SO


class _Coding(ctypes.Structure):
    _fields_ = [
        ("codes", ctypes.c_uint16 * 512),  # (bits, len) pairs
        ("sorted_syms", ctypes.c_uint8 * 256),
        ("num_syms", ctypes.c_int32),
        ("len_count", ctypes.c_uint16 * 33),
        ("len_mask", ctypes.c_uint32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L.ho_compress_bound.restype = sz
        L.ho_compress_bound.argtypes = [sz, ci]
        L.ho_compress.restype = sz
        L.ho_compress.argtypes = [ci, vp, sz, vp, sz]
        L.ho_compress_with_coding.restype = sz
        L.ho_compress_with_coding.argtypes = [ci, ctypes.POINTER(_Coding), vp, sz, vp, sz]
        L.ho_decompress.restype = ctypes.c_long
        L.ho_decompress.argtypes = [ci, vp, sz, vp, sz]
        L.ho_histogram.restype = None
        L.ho_histogram.argtypes = [vp, sz, vp]
        L.ho_make_coding.restype = None
        L.ho_make_coding.argtypes = [vp, ctypes.POINTER(_Coding)]
        L.ho_sort_syms_desc.restype = None
        L.ho_sort_syms_desc.argtypes = [vp, ci, vp]
        L.ho_slice_sizes.restype = None
        L.ho_slice_sizes.argtypes = [sz, ci, vp]
        L.ho_fnv1a64_seeded.restype = ctypes.c_uint64
        L.ho_fnv1a64_seeded.argtypes = [vp, sz, ctypes.c_uint64]
        L.ho_generate.restype = None
        L.ho_generate.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64, vp, sz]
        L.ho_time_roundtrip.restype = ci
        L.ho_time_roundtrip.argtypes = [ci, vp, sz, sz, ci, ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_double), ctypes.POINTER(sz)]
        L.ho_time_roundtrip_mt.restype = ci
        L.ho_time_roundtrip_mt.argtypes = [ci, vp, sz, sz, ci, ci, ctypes.POINTER(ctypes.c_double),
                                           ctypes.POINTER(ctypes.c_double), ctypes.POINTER(sz)]
        _lib = L
    return _lib


def _buf(b):
    """bytes / bytearray / uint8 ndarray -> (pointer-or-None, length, keepalive)."""
    if isinstance(b, np.ndarray):
        a = np.ascontiguousarray(b, dtype=np.uint8)
    else:
        a = np.frombuffer(bytes(b), dtype=np.uint8)
    if a.size == 0:
        a = np.zeros(1, dtype=np.uint8)
        return a.ctypes.data, 0, a
    return a.ctypes.data, a.size, a


def compress_bound(n, K):
    return lib().ho_compress_bound(n, K)


def compress(K, raw):
    p, n, keep = _buf(raw)
    cap = compress_bound(n, K)
    out = np.zeros(cap, dtype=np.uint8)
    got = lib().ho_compress(K, p, n, out.ctypes.data, cap)
    if got == 0:
        raise RuntimeError("ho_compress failed")
    return out[:got].tobytes()


def decompress(K, comp):
    p, n, keep = _buf(comp)
    if n < 8:
        raise ValueError("frame shorter than its fixed header")
    raw_size = int(np.frombuffer(bytes(comp[:4]), dtype="<u4")[0])
    out = np.zeros(max(raw_size, 1) + 8, dtype=np.uint8)
    got = lib().ho_decompress(K, p, n, out.ctypes.data, raw_size)
    if got < 0:
        raise RuntimeError("ho_decompress failed")
    return out[:got].tobytes()


def histogram(raw):
    p, n, keep = _buf(raw)
    h = np.zeros(256, dtype=np.uint32)
    lib().ho_histogram(p, n, h.ctypes.data)
    return h


class Coding:
    def __init__(self, c):
        codes = np.frombuffer(bytes(c.codes), dtype=np.uint16).reshape(256, 2)
        self.bits = codes[:, 0].copy()
        self.len = codes[:, 1].copy()
        self.num_syms = int(c.num_syms)
        self.sorted_syms = np.frombuffer(bytes(c.sorted_syms), dtype=np.uint8)[: self.num_syms].copy()
        self.len_count = np.frombuffer(bytes(c.len_count), dtype=np.uint16).copy()
        self.len_mask = int(c.len_mask)
        self._c = c


def make_coding(hist):
    h = np.ascontiguousarray(hist, dtype=np.uint32)
    assert h.size == 256
    c = _Coding()
    lib().ho_make_coding(h.ctypes.data, ctypes.byref(c))
    return Coding(c)


def compress_with_coding(K, coding, raw):
    p, n, keep = _buf(raw)
    cap = compress_bound(n, K)
    out = np.zeros(cap, dtype=np.uint8)
    got = lib().ho_compress_with_coding(K, ctypes.byref(coding._c), p, n, out.ctypes.data, cap)
    if got == 0:
        raise RuntimeError("ho_compress_with_coding failed")
    return out[:got].tobytes()


def sort_syms_desc(syms, count):
    s = np.ascontiguousarray(syms, dtype=np.uint8).copy()
    c = np.ascontiguousarray(count, dtype=np.uint32)
    lib().ho_sort_syms_desc(s.ctypes.data, int(s.size), c.ctypes.data)
    return s


def slice_sizes(n, K):
    out = np.zeros(K, dtype=np.uint64)
    lib().ho_slice_sizes(n, K, out.ctypes.data)
    return out


def pin_hash(b):
    p, n, keep = _buf(b)
    return "%016x" % lib().ho_fnv1a64_seeded(p, n, PIN_HASH_BASIS)


def generate(cdf, seed, n, first_index=0):
    c = np.ascontiguousarray(cdf, dtype=np.uint32)
    assert c.size == 256
    out = np.zeros(max(n, 1), dtype=np.uint8)
    lib().ho_generate(c.ctypes.data, seed, first_index, out.ctypes.data, n)
    return out[:n]


def time_roundtrip(K, raw, frame_bytes, reps=1, threads=1):
    """Seconds spent compressing / decompressing `raw` frame by frame, `reps` times;
    threads > 1 spreads the frames of each phase over that many POSIX threads."""
    p, n, keep = _buf(raw)
    tc, td, ct = ctypes.c_double(), ctypes.c_double(), ctypes.c_size_t()
    if threads > 1:
        rc = lib().ho_time_roundtrip_mt(K, p, n, frame_bytes, reps, threads, ctypes.byref(tc),
                                        ctypes.byref(td), ctypes.byref(ct))
    else:
        rc = lib().ho_time_roundtrip(K, p, n, frame_bytes, reps, ctypes.byref(tc),
                                     ctypes.byref(td), ctypes.byref(ct))
    if rc != 0:
        raise RuntimeError("oracle round trip failed: %d" % rc)
    return tc.value, td.value, ct.value


# ---- the reference's own histogram code, compiled from its sources -----------
# oracle/_ref/libref_hist.so is built by `make -C oracle ref` (needs
# /root/reference, so only where the reference is present); it exists to check
# ho_histogram against MakeHistogram itself.

_REF_SO = os.path.join(_HERE, "_ref", "libref_hist.so")
REF_HIST_VARIANTS = {"MakeHistogram": 0, "Simple": 1, "Multi": 2, "Vectorized": 3,
                     "GatherScatter": 4}
_ref = None


def ref_available():
    """True when the compiled reference histogram exists and this CPU can run it
    (the reference's histogram.cpp is built with the AVX-512 subsets it uses)."""
    if not os.path.exists(_REF_SO):
        return False
    try:
        with open("/proc/cpuinfo") as f:
            flags = f.read()
    except OSError:
        return False
    return all(x in flags for x in ("avx512f", "avx512bw", "avx512dq", "avx512vl", "bmi2"))


def ref_histogram(raw, variant="MakeHistogram"):
    """huffman::MakeHistogram* of the reference (codec/histogram.cpp) on `raw`."""
    global _ref
    if _ref is None:
        L = ctypes.CDLL(_REF_SO)
        L.ref_histogram.restype = ctypes.c_int
        L.ref_histogram.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        _ref = L
    p, n, keep = _buf(raw)
    h = np.zeros(256, dtype=np.uint32)
    rc = _ref.ref_histogram(REF_HIST_VARIANTS[variant], p, n, h.ctypes.data)
    if rc != 0:
        raise RuntimeError("ref_histogram failed")
    return h
