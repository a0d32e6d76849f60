"""MI355X-native K-stream Huffman codec: Python front end of libhfz.so.

The codec itself is hand-written HIP behind a C ABI (include/hfz.h); this module
only binds it with ctypes and mirrors the reference's compressor interface
(codec/huffman.h:42-52: static Compress / Decompress / name).  There is no CPU
fallback: if libhfz.so is missing or no GPU is usable, calls raise.

The directory name contains a hyphen, so import it with
``importlib.import_module("huffman-avx512_amd")``.
"""
import ctypes
import os

import numpy as np

from . import distributions  # noqa: F401  (integer-only synthetic sources)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhfz.so")

HFZ_OK = 0
HFZ_E_ARG = -1
HFZ_E_CAPACITY = -2
HFZ_E_CORRUPT = -3
HFZ_E_NO_DEVICE = -4
HFZ_E_HIP = -5
HFZ_E_TOO_LARGE = -6
HFZ_E_WORKSPACE = -7
HFZ_E_CODE_MISMATCH = -8
MAX_STREAMS = 64
DECODER_TABLE_FREE = -1

# Every symbol include/hfz.h declares.
EXPORTS = (
    "hfz_version", "hfz_strerror", "hfz_last_hip_error", "hfz_device_count",
    "hfz_compress_bound", "hfz_compress", "hfz_decompress", "hfz_decompressed_size",
    "hfz_histogram", "hfz_build_code_table", "hfz_batch_frames", "hfz_workspace_bytes",
    "hfz_compress_batch_dev", "hfz_decompress_batch_dev", "hfz_histogram_dev",
    "hfz_histogram_batch_dev", "hfz_generate_dev", "hfz_timing_enable", "hfz_timing_read",
    "hfz_frame_code_from_table", "hfz_build_codes_dev", "hfz_set_decoder_table_bits",
    "hfz_compress_batch_shared_dev", "hfz_decompress_batch_shared_dev", "hfz_set_frames_per_wave",
    "hfz_count_batch_dev", "hfz_compress_batch_counted_dev", "hfz_set_encoder_block",
    "hfz_static_workspace_init", "hfz_static_max_frames", "hfz_compress_batch_static_dev",
    "hfz_set_decoder_store", "hfz_set_encoder_parts", "hfz_set_decoder_parts",
    "hfz_set_decoder_input_block",
)


class HfzError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        detail = ""
        try:
            msg = lib().hfz_strerror(code).decode()
            hip = lib().hfz_last_hip_error().decode()
            detail = msg + (" [" + hip + "]" if hip and code in (HFZ_E_HIP, HFZ_E_NO_DEVICE) else "")
        except Exception:  # pragma: no cover
            pass
        super().__init__("%s failed: %d %s" % (where, code, detail))


class BatchTiming(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in
                ("hist_ms", "table_ms", "sizes_ms", "place_ms", "encode_ms", "decode_ms")] + [
                    ("compress_calls", ctypes.c_int32), ("decompress_calls", ctypes.c_int32),
                    ("count_ms", ctypes.c_float), ("count_calls", ctypes.c_int32)]


class FrameCode(ctypes.Structure):
    _fields_ = [("enc", ctypes.c_uint16 * 256), ("hdr", ctypes.c_uint8 * 276),
                ("hdr_len", ctypes.c_uint32)]


class CodeTable(ctypes.Structure):
    _fields_ = [
        ("bits", ctypes.c_uint16 * 256),
        ("len", ctypes.c_uint8 * 256),
        ("sorted_syms", ctypes.c_uint8 * 256),
        ("len_count", ctypes.c_uint16 * 13),
        ("num_syms", ctypes.c_uint16),
        ("len_mask", ctypes.c_uint32),
    ]


_lib = None


def lib():
    """Loads libhfz.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libhfz.so not built: run `make -C %s`" % os.path.join(_HERE, "csrc"))
        L = ctypes.CDLL(LIB_PATH)
        vp, sz, ci, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64
        psz = ctypes.POINTER(ctypes.c_size_t)
        L.hfz_version.restype = ctypes.c_char_p
        L.hfz_strerror.restype = ctypes.c_char_p
        L.hfz_strerror.argtypes = [ci]
        L.hfz_last_hip_error.restype = ctypes.c_char_p
        L.hfz_device_count.restype = ci
        L.hfz_compress_bound.restype = sz
        L.hfz_compress_bound.argtypes = [sz, ci]
        L.hfz_compress.restype = ci
        L.hfz_compress.argtypes = [ci, vp, sz, vp, sz, psz]
        L.hfz_decompress.restype = ci
        L.hfz_decompress.argtypes = [ci, vp, sz, vp, sz, psz]
        L.hfz_decompressed_size.restype = ci
        L.hfz_decompressed_size.argtypes = [vp, sz, psz]
        L.hfz_histogram.restype = ci
        L.hfz_histogram.argtypes = [vp, sz, vp]
        L.hfz_build_code_table.restype = ci
        L.hfz_build_code_table.argtypes = [vp, ctypes.POINTER(CodeTable)]
        L.hfz_batch_frames.restype = sz
        L.hfz_batch_frames.argtypes = [sz, sz]
        L.hfz_workspace_bytes.restype = sz
        L.hfz_workspace_bytes.argtypes = [ci, sz]
        L.hfz_compress_batch_dev.restype = ci
        L.hfz_compress_batch_dev.argtypes = [ci, vp, sz, sz, vp, sz, sz, vp, vp, vp, vp, sz, vp]
        L.hfz_decompress_batch_dev.restype = ci
        L.hfz_decompress_batch_dev.argtypes = [ci, vp, sz, vp, vp, sz, vp, sz, sz, vp, vp]
        L.hfz_compress_batch_shared_dev.restype = ci
        L.hfz_compress_batch_shared_dev.argtypes = [ci, vp, sz, sz, vp, vp, sz, sz, vp, vp, vp, vp, sz, vp]
        L.hfz_decompress_batch_shared_dev.restype = ci
        L.hfz_decompress_batch_shared_dev.argtypes = [ci, vp, sz, vp, vp, sz, vp, sz, sz, vp, vp]
        L.hfz_count_batch_dev.restype = ci
        L.hfz_count_batch_dev.argtypes = [ci, vp, sz, sz, vp, sz, vp, vp]
        L.hfz_compress_batch_counted_dev.restype = ci
        L.hfz_compress_batch_counted_dev.argtypes = [ci, vp, sz, sz, vp, vp, sz, sz, vp, vp, vp, vp, sz, vp]
        L.hfz_static_workspace_init.restype = ci
        L.hfz_static_workspace_init.argtypes = []
        L.hfz_static_max_frames.restype = sz
        L.hfz_static_max_frames.argtypes = []
        L.hfz_compress_batch_static_dev.restype = ci
        L.hfz_compress_batch_static_dev.argtypes = [ci, vp, sz, sz, vp, sz, sz, vp, vp, vp, vp]
        L.hfz_histogram_dev.restype = ci
        L.hfz_histogram_dev.argtypes = [vp, sz, vp, vp]
        L.hfz_histogram_batch_dev.restype = ci
        L.hfz_histogram_batch_dev.argtypes = [vp, sz, sz, vp, vp]
        L.hfz_generate_dev.restype = ci
        L.hfz_generate_dev.argtypes = [vp, u64, u64, vp, sz, vp]
        L.hfz_frame_code_from_table.restype = ci
        L.hfz_frame_code_from_table.argtypes = [ctypes.POINTER(CodeTable), ctypes.POINTER(FrameCode)]
        L.hfz_build_codes_dev.restype = ci
        L.hfz_build_codes_dev.argtypes = [vp, sz, vp, vp]
        L.hfz_timing_enable.restype = ci
        L.hfz_timing_enable.argtypes = [ci]
        L.hfz_set_encoder_block.restype = ci
        L.hfz_set_encoder_block.argtypes = [ci]
        L.hfz_set_encoder_parts.restype = ci
        L.hfz_set_encoder_parts.argtypes = [ci]
        L.hfz_set_decoder_parts.restype = ci
        L.hfz_set_decoder_parts.argtypes = [ci]
        L.hfz_set_decoder_input_block.restype = ci
        L.hfz_set_decoder_input_block.argtypes = [ci]
        L.hfz_set_frames_per_wave.restype = ci
        L.hfz_set_frames_per_wave.argtypes = [ci]
        L.hfz_set_decoder_table_bits.restype = ci
        L.hfz_set_decoder_table_bits.argtypes = [ci]
        L.hfz_set_decoder_store.restype = ci
        L.hfz_set_decoder_store.argtypes = [ci]
        L.hfz_timing_read.restype = ci
        L.hfz_timing_read.argtypes = [ctypes.POINTER(BatchTiming)]
        _lib = L
    return _lib


def _check(rc, where):
    if rc != HFZ_OK:
        raise HfzError(rc, where)


def _host(b):
    if isinstance(b, np.ndarray):
        a = np.ascontiguousarray(b, dtype=np.uint8)
    else:
        a = np.frombuffer(bytes(b), dtype=np.uint8)
    if a.size == 0:
        return np.zeros(1, dtype=np.uint8), 0
    return a, a.size


# ---- host-buffer API (mirrors CompressMulti<K> / DecompressMulti<K>) -------

def compress_bound(n, K):
    return lib().hfz_compress_bound(n, K)


def compress(K, raw):
    """CompressMulti<K>(raw) (codec/huffman.h:9-10) on the GPU; returns bytes."""
    a, n = _host(raw)
    cap = compress_bound(n, K)
    out = np.empty(cap, dtype=np.uint8)
    got = ctypes.c_size_t()
    _check(lib().hfz_compress(K, a.ctypes.data, n, out.ctypes.data, cap, ctypes.byref(got)),
           "hfz_compress")
    return out[: got.value].tobytes()


def decompress(K, comp):
    """DecompressMulti<K>(compressed) (codec/huffman.h:11-12) on the GPU."""
    a, n = _host(comp)
    raw = ctypes.c_size_t()
    _check(lib().hfz_decompressed_size(a.ctypes.data, n, ctypes.byref(raw)), "hfz_decompressed_size")
    out = np.empty(max(raw.value, 1), dtype=np.uint8)
    got = ctypes.c_size_t()
    _check(lib().hfz_decompress(K, a.ctypes.data, n, out.ctypes.data, raw.value, ctypes.byref(got)),
           "hfz_decompress")
    return out[: got.value].tobytes()


def histogram(raw):
    """MakeHistogram(str) (codec/histogram.h:12) on the GPU; 256 x uint32."""
    a, n = _host(raw)
    h = np.zeros(256, dtype=np.uint32)
    _check(lib().hfz_histogram(a.ctypes.data, n, h.ctypes.data), "hfz_histogram")
    return h


def histogram_dev(raw):
    """256 byte counts (numpy uint64) of a device tensor, by hfz_histogram_dev in
    pieces small enough for its 32-bit counters."""
    import torch
    assert raw.is_cuda and raw.dtype == torch.uint8
    total = np.zeros(256, dtype=np.uint64)
    d_h = torch.zeros(256, dtype=torch.int32, device=raw.device)
    piece = 1 << 31
    st = torch.cuda.current_stream(raw.device).cuda_stream
    for lo in range(0, raw.numel(), piece):
        part = raw[lo:lo + piece]
        _check(lib().hfz_histogram_dev(part.data_ptr(), part.numel(), d_h.data_ptr(), st),
               "hfz_histogram_dev")
        total += d_h.cpu().numpy().view(np.uint32).astype(np.uint64)
    return total


def scale_histogram(hist):
    """256 counts of any width -> uint32 counts for the code builder.  The builders
    (host, device and the reference's MakeCanonicalCoding, codec/huffman.cpp:364-416)
    add node counts in 32 bits, which is exact only while the SUM of the counts
    stays below 2^32 (the reference's frames are below 4 GiB, so it always does
    there).  All counts are therefore shifted right until the sum is below 2^31;
    a symbol that occurs at all keeps a count of at least 1, so it keeps a code
    (that adds at most 255 to the sum)."""
    h = np.asarray(hist, dtype=np.uint64).reshape(256)
    shift = max(0, int(h.sum(dtype=np.uint64)).bit_length() - 31)
    if shift:
        h = np.where(h > 0, np.maximum(h >> np.uint64(shift), np.uint64(1)), np.uint64(0))
    return h.astype(np.uint32)


def scale_histogram_dev(h):
    """scale_histogram for a device (or host) int64 tensor of 256 counts, without
    a host round trip; returns the int32 tensor holding the uint32 bit patterns,
    ready for hfz_build_codes_dev."""
    import torch
    assert h.dtype == torch.int64 and h.numel() == 256
    total = h.sum()
    # bit_length(total) - 31, at least 0: number of k in 31..62 with total >> k != 0
    ks = torch.arange(31, 63, dtype=torch.int64, device=h.device)
    shift = ((total >> ks) > 0).sum()
    scaled = torch.where(h > 0, torch.clamp(h >> shift, min=1), h)
    return scaled.to(torch.int32)


def build_code_table(hist):
    """MakeCanonicalCoding (codec/huffman.cpp:339-437); host only."""
    h = np.ascontiguousarray(hist, dtype=np.uint32)
    if h.size != 256:
        raise ValueError("histogram must have 256 bins")
    t = CodeTable()
    _check(lib().hfz_build_code_table(h.ctypes.data, ctypes.byref(t)), "hfz_build_code_table")
    return t


def frame_code(hist):
    """Host-built encoder record (enc table + header bytes) for one histogram."""
    t = build_code_table(hist)
    fc = FrameCode()
    _check(lib().hfz_frame_code_from_table(ctypes.byref(t), ctypes.byref(fc)),
           "hfz_frame_code_from_table")
    return fc


def build_codes_dev(hists):
    """Device-built encoder records for an (n, 256) uint32 array of histograms;
    returns an (n, 792) uint8 numpy array of hfz_frame_code records."""
    import torch
    h = np.ascontiguousarray(hists, dtype=np.uint32).reshape(-1, 256)
    n = h.shape[0]
    d_h = torch.as_tensor(h.view(np.int32)).cuda()
    d_c = torch.zeros(n * ctypes.sizeof(FrameCode), dtype=torch.uint8, device="cuda")
    _check(lib().hfz_build_codes_dev(d_h.data_ptr(), n, d_c.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream), "hfz_build_codes_dev")
    torch.cuda.synchronize()
    return d_c.cpu().numpy().reshape(n, ctypes.sizeof(FrameCode))


class HuffmanCompressorGpu:
    """The reference's compressor concept (codec/huffman.h:42-52) for a fixed K."""

    def __init__(self, K):
        if not 1 <= K <= MAX_STREAMS:
            raise ValueError("K must be in 1..%d" % MAX_STREAMS)
        self.K = K

    def Compress(self, raw):
        return compress(self.K, raw)

    def Decompress(self, compressed):
        return decompress(self.K, compressed)

    def name(self):
        return "HuffmanGpu<%d>" % self.K


# ---- device-resident batch API (torch tensors carry the device memory) -----

class Batch:
    """Compressed batch living in device memory."""

    def __init__(self, K, comp, comp_off, comp_size, status, total, frame_bytes):
        self.K, self.comp, self.comp_off, self.comp_size = K, comp, comp_off, comp_size
        self.status, self.total, self.frame_bytes = status, total, frame_bytes

    @property
    def n_frames(self):
        return int(self.comp_off.numel())

    def frame_bytes_host(self, i):
        off = int(self.comp_off[i].item())
        size = int(self.comp_size[i].item())
        return bytes(self.comp[off:off + size].cpu().numpy().tobytes())

    def compressed_bytes(self):
        return int(self.comp_size.to("cpu").to(dtype=__import__("torch").int64).sum().item())


class BatchCodec:
    """Pre-allocated buffers + the batch entry points for one (K, total, frame_bytes)."""

    def __init__(self, K, total, frame_bytes, device="cuda", strided=False):
        import torch
        self.torch = torch
        self.K, self.total, self.frame_bytes = K, total, frame_bytes
        self.device = torch.device(device)
        L = lib()
        self.n_frames = L.hfz_batch_frames(total, frame_bytes)
        bound = L.hfz_compress_bound(frame_bytes, K)
        self.stride = ((bound + 255) // 256) * 256 if strided else 0
        # packed capacity: the sum of per-frame bounds is a safe upper limit
        self.comp_cap = self.n_frames * (self.stride if strided else bound)
        self.comp = torch.empty(self.comp_cap, dtype=torch.uint8, device=self.device)
        self.comp_off = torch.zeros(self.n_frames, dtype=torch.int64, device=self.device)
        self.comp_size = torch.zeros(self.n_frames, dtype=torch.int32, device=self.device)
        self.status = torch.zeros(self.n_frames, dtype=torch.int32, device=self.device)
        self.dstatus = torch.zeros(self.n_frames, dtype=torch.int32, device=self.device)
        self.ws_bytes = L.hfz_workspace_bytes(K, self.n_frames)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def shared_code(self, raw):
        """One code for the whole batch: device histogram of `raw`, canonical code on
        the host (as hfz_build_code_table), record copied to the device.  Returns the
        device tensor to pass to compress(..., code=...).  A multi-GPU job sums the
        histograms of all ranks before building (see histogram_dev)."""
        return self.code_from_histogram(histogram_dev(raw))

    def code_from_histogram(self, hist):
        """hist: 256 counts (any integer width).  Counts beyond 32 bits are scaled
        down, keeping every symbol that occurs at all."""
        torch = self.torch
        fc = frame_code(scale_histogram(hist))
        host = torch.frombuffer(bytearray(bytes(fc)), dtype=torch.uint8)
        return host.to(self.device)

    def count(self, raw):
        """First half of the two-step shared-code encoder: one pass over `raw` that
        leaves per-stream counts in the workspace; returns the batch's 256 byte
        counts as a device int64 tensor (sum them over ranks, build the code, then
        compress(raw, code=..., counted=True))."""
        torch = self.torch
        assert raw.dtype == torch.uint8 and raw.numel() == self.total and raw.is_cuda
        if not hasattr(self, "hist_total"):
            self.hist_total = torch.zeros(256, dtype=torch.int64, device=self.device)
        _check(lib().hfz_count_batch_dev(self.K, raw.data_ptr(), self.total, self.frame_bytes,
                                         self.ws.data_ptr(), self.ws_bytes,
                                         self.hist_total.data_ptr(), self._stream()),
               "hfz_count_batch_dev")
        return self.hist_total

    def compress(self, raw, code=None, counted=False):
        assert raw.dtype == self.torch.uint8 and raw.numel() == self.total and raw.is_cuda
        if code is not None:
            assert code.is_cuda and code.numel() == ctypes.sizeof(FrameCode)
            fn = lib().hfz_compress_batch_counted_dev if counted else lib().hfz_compress_batch_shared_dev
            _check(fn(
                self.K, raw.data_ptr(), self.total, self.frame_bytes, code.data_ptr(),
                self.comp.data_ptr(), self.comp_cap, self.stride, self.comp_off.data_ptr(),
                self.comp_size.data_ptr(), self.status.data_ptr(), self.ws.data_ptr(),
                self.ws_bytes, self._stream()), "hfz_compress_batch_shared_dev")
            return Batch(self.K, self.comp, self.comp_off, self.comp_size, self.status,
                         self.total, self.frame_bytes)
        _check(lib().hfz_compress_batch_dev(
            self.K, raw.data_ptr(), self.total, self.frame_bytes, self.comp.data_ptr(),
            self.comp_cap, self.stride, self.comp_off.data_ptr(), self.comp_size.data_ptr(),
            self.status.data_ptr(), self.ws.data_ptr(), self.ws_bytes, self._stream()),
            "hfz_compress_batch_dev")
        return Batch(self.K, self.comp, self.comp_off, self.comp_size, self.status, self.total,
                     self.frame_bytes)

    def decompress(self, batch, out=None, shared=False):
        torch = self.torch
        if out is None:
            out = torch.empty(max(self.total, 1), dtype=torch.uint8, device=self.device)
        fn = lib().hfz_decompress_batch_shared_dev if shared else lib().hfz_decompress_batch_dev
        _check(fn(
            self.K, batch.comp.data_ptr(), batch.comp.numel(), batch.comp_off.data_ptr(),
            batch.comp_size.data_ptr(), batch.n_frames, out.data_ptr(), self.total,
            self.frame_bytes, self.dstatus.data_ptr(), self._stream()),
            "hfz_decompress_batch_shared_dev" if shared else "hfz_decompress_batch_dev")
        return out[: self.total]


def shard_frames(n_frames, rank, world):
    """Contiguous range [lo, hi) of frame indices that `rank` of `world` owns.
    Frames are independent units (own header, table and streams), so sharding a
    batch over GPUs needs no exchange: every rank encodes/decodes its range and
    results are concatenated by frame index."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    q, r = divmod(n_frames, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def set_decoder_table_bits(bits=0):
    """0 = automatic, 11 or 12 = force that build of the decoder's lookup table,
    DECODER_TABLE_FREE (-1) = the table-free compare-and-index build."""
    _check(lib().hfz_set_decoder_table_bits(int(bits)), "hfz_set_decoder_table_bits")


def set_decoder_store(mode=0):
    """0 = automatic, 1 = every lane stores its own blocks, 2 = lane pairs store together."""
    _check(lib().hfz_set_decoder_store(int(mode)), "hfz_set_decoder_store")


def set_frames_per_wave(n=0):
    """0 = automatic, 1 = one frame per wave, 2 = two (K <= 32)."""
    _check(lib().hfz_set_frames_per_wave(int(n)), "hfz_set_frames_per_wave")


def set_encoder_block(nbytes=0):
    """0 = automatic, 32 or 64 = bytes per block the encoder stores."""
    _check(lib().hfz_set_encoder_block(int(nbytes)), "hfz_set_encoder_block")


def set_encoder_parts(n=0):
    """0 = automatic, 1 = one thread per stream, n = up to n threads per stream."""
    _check(lib().hfz_set_encoder_parts(int(n)), "hfz_set_encoder_parts")


def set_decoder_parts(n=0):
    """0 = automatic, 1 = one thread per stream, n = up to n threads per stream."""
    _check(lib().hfz_set_decoder_parts(int(n)), "hfz_set_decoder_parts")


def set_decoder_input_block(nbytes=0):
    """0 or 64 = 64-byte input blocks, 32 = 32-byte ones (11-bit-table decoder)."""
    _check(lib().hfz_set_decoder_input_block(int(nbytes)), "hfz_set_decoder_input_block")


def timing_enable(every=1):
    """Record kernel times on every `every`-th batch call (True = 1, False/0 = off)."""
    _check(lib().hfz_timing_enable(int(every)), "hfz_timing_enable")


def timing_read():
    t = BatchTiming()
    _check(lib().hfz_timing_read(ctypes.byref(t)), "hfz_timing_read")
    return {n: getattr(t, n) for n, _ in BatchTiming._fields_}


def generate(cdf, seed, n, device="cuda", first_index=0):
    """Synthetic bytes on the device (include/hfz.h: hfz_generate_dev)."""
    import torch
    c = torch.as_tensor(np.ascontiguousarray(cdf, dtype=np.uint32).view(np.int32)).to(device)
    out = torch.empty(max(n, 1), dtype=torch.uint8, device=device)
    _check(lib().hfz_generate_dev(c.data_ptr(), seed, first_index, out.data_ptr(), n,
                                  torch.cuda.current_stream().cuda_stream), "hfz_generate_dev")
    torch.cuda.current_stream().synchronize()
    return out[:n]
