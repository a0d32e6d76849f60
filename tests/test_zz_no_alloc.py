"""The allocation-free batch encoder (hfz_compress_batch_static_dev) against
hfz_compress_batch_dev, through the C ABI: same frames, same offsets, same
sizes, on two calls in a row of which the second is smaller (the library's
device array keeps the first call's contents), plus the size limit."""
import importlib

import numpy as np
import pytest

from helpers import geometric_cdf

pytestmark = pytest.mark.gpu

hfz = importlib.import_module("huffman-avx512_amd")


def _both(torch, K, raw, total, frame_bytes, strided):
    L = hfz.lib()
    codec = hfz.BatchCodec(K, total, frame_bytes, strided=strided)   # the existing entry point
    ref = codec.compress(raw)
    torch.cuda.synchronize()
    assert int(codec.status.abs().sum().item()) == 0
    n = codec.n_frames
    comp = torch.full((codec.comp_cap,), 0xEE, dtype=torch.uint8, device="cuda")
    off = torch.zeros(n, dtype=torch.int64, device="cuda")
    size = torch.zeros(n, dtype=torch.int32, device="cuda")
    status = torch.full((n,), 77, dtype=torch.int32, device="cuda")
    rc = L.hfz_compress_batch_static_dev(K, raw.data_ptr() if total else None, total, frame_bytes,
                                         comp.data_ptr(), codec.comp_cap, codec.stride,
                                         off.data_ptr(), size.data_ptr(), status.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [0] * n
    assert torch.equal(off, ref.comp_off) and torch.equal(size, ref.comp_size)
    offs, sizes = off.cpu().numpy(), size.cpu().numpy()
    a, b = comp.cpu().numpy(), ref.comp.cpu().numpy()
    for i in range(n):
        assert a[offs[i]:offs[i] + sizes[i]].tobytes() == b[offs[i]:offs[i] + sizes[i]].tobytes(), i
    # and the frames decode back to the input
    out = torch.zeros(max(total, 1), dtype=torch.uint8, device="cuda")
    dstatus = torch.full((n,), 77, dtype=torch.int32, device="cuda")
    rc = L.hfz_decompress_batch_dev(K, comp.data_ptr(), comp.numel(), off.data_ptr(), size.data_ptr(),
                                    n, out.data_ptr(), total, frame_bytes, dstatus.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert dstatus.cpu().tolist() == [0] * n
    assert torch.equal(out[:total], raw[:total])


def test_static_workspace_encoder_matches_batch_encoder():
    import torch
    assert torch.cuda.is_available()
    L = hfz.lib()
    assert L.hfz_static_workspace_init() == 0
    max_frames = L.hfz_static_max_frames()
    assert max_frames == 256
    cdf = geometric_cdf()
    # first call: as many frames as the static workspace holds, counting-kernel path
    K, frame_bytes = 64, 64 << 10
    total = max_frames * frame_bytes - 1234
    raw = hfz.generate(cdf, 11, total)
    _both(torch, K, raw, total, frame_bytes, False)
    # second call, smaller and of another shape: 32 streams, ragged, strided layout
    K2, fb2 = 32, 50000
    total2 = 7 * fb2 + 99
    raw2 = hfz.generate(hfz.distributions.enwik_like_cdf(), 12, total2)
    _both(torch, K2, raw2, total2, fb2, True)
    # third: slices too long for the packed counters (separate histogram and sizes passes)
    K3, fb3 = 64, 5 << 20
    total3 = 2 * fb3 + 17
    raw3 = hfz.generate(cdf, 13, total3)
    _both(torch, K3, raw3, total3, fb3, False)
    # an empty batch is one empty frame
    _both(torch, 8, torch.empty(0, dtype=torch.uint8, device="cuda"), 0, 4096, False)


def test_static_workspace_encoder_refuses_too_many_frames():
    import torch
    L = hfz.lib()
    max_frames = L.hfz_static_max_frames()
    frame_bytes = 4096
    total = (max_frames + 1) * frame_bytes
    raw = hfz.generate(geometric_cdf(), 5, total)
    n = max_frames + 1
    comp = torch.zeros(n * L.hfz_compress_bound(frame_bytes, 16), dtype=torch.uint8, device="cuda")
    off = torch.zeros(n, dtype=torch.int64, device="cuda")
    size = torch.zeros(n, dtype=torch.int32, device="cuda")
    rc = L.hfz_compress_batch_static_dev(16, raw.data_ptr(), total, frame_bytes, comp.data_ptr(),
                                         comp.numel(), 0, off.data_ptr(), size.data_ptr(), None, None)
    assert rc == hfz.HFZ_E_TOO_LARGE
    # one frame fewer is accepted
    rc = L.hfz_compress_batch_static_dev(16, raw.data_ptr(), total - frame_bytes, frame_bytes,
                                         comp.data_ptr(), comp.numel(), 0, off.data_ptr(),
                                         size.data_ptr(), None, None)
    assert rc == 0
    torch.cuda.synchronize()


# ---- under HIP graph capture ---------------------------------------------------

def _hip_runtime():
    """The HIP runtime this process already uses (torch and libhfz.so share it),
    found in the process map so that no second copy gets loaded."""
    import ctypes
    path = None
    with open("/proc/self/maps") as maps:
        for line in maps:
            if "libamdhip64.so" in line:
                path = line.split()[-1]
                break
    assert path, "HIP runtime not loaded"
    hip = ctypes.CDLL(path)
    vp = ctypes.c_void_p
    hip.hipStreamBeginCapture.argtypes = [vp, ctypes.c_int]
    hip.hipStreamEndCapture.argtypes = [vp, ctypes.POINTER(vp)]
    hip.hipGraphInstantiate.argtypes = [ctypes.POINTER(vp), vp, vp, vp, ctypes.c_size_t]
    hip.hipGraphLaunch.argtypes = [vp, vp]
    hip.hipGraphExecDestroy.argtypes = [vp]
    hip.hipGraphDestroy.argtypes = [vp]
    hip.hipGraphGetNodes.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_size_t)]
    hip.hipMemGetInfo.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    return hip


@pytest.mark.parametrize("K,n_frames,frame_bytes", [
    (64, 200, 64 << 10),   # one wave per frame, counting kernel
    (64, 3, 256 << 10),    # few frames: several threads per stream, 64 KiB of dynamic LDS
    (32, 9, 50000),        # 32 streams, ragged sizes
])
def test_static_workspace_encoder_records_into_a_hip_graph(K, n_frames, frame_bytes):
    """What the entry point exists for: a call between hipStreamBeginCapture and
    hipStreamEndCapture (global mode: any allocation, free or synchronisation by
    any thread would fail the capture) records the whole batch encoder, and the
    graph, replayed on fresh inputs, writes the frames hfz_compress_batch_dev writes."""
    import ctypes
    import torch
    L = hfz.lib()
    hip = _hip_runtime()
    assert L.hfz_static_workspace_init() == 0
    total = n_frames * frame_bytes - 321
    cdf = geometric_cdf()
    inputs = [hfz.generate(cdf, 100 + i, total) for i in range(3)]
    ref = hfz.BatchCodec(K, total, frame_bytes)
    raw = torch.empty(total, dtype=torch.uint8, device="cuda")
    comp = torch.empty(ref.comp_cap, dtype=torch.uint8, device="cuda")
    off = torch.zeros(ref.n_frames, dtype=torch.int64, device="cuda")
    size = torch.zeros(ref.n_frames, dtype=torch.int32, device="cuda")
    status = torch.zeros(ref.n_frames, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()

    def call():
        return L.hfz_compress_batch_static_dev(K, raw.data_ptr(), total, frame_bytes, comp.data_ptr(),
                                               ref.comp_cap, 0, off.data_ptr(), size.data_ptr(),
                                               status.data_ptr(), st.cuda_stream)

    # first use outside a capture (the runtime loads the kernels on their first launch)
    raw.copy_(inputs[0])
    torch.cuda.synchronize()
    assert call() == 0
    torch.cuda.synchronize()
    free0, tot0 = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(free0), ctypes.byref(tot0)) == 0

    graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipStreamBeginCapture(st.cuda_stream, 0) == 0   # hipStreamCaptureModeGlobal
    rc = call()
    end = hip.hipStreamEndCapture(st.cuda_stream, ctypes.byref(graph))
    assert rc == 0 and end == 0 and graph.value, (rc, end)
    n_nodes = ctypes.c_size_t()
    assert hip.hipGraphGetNodes(graph, None, ctypes.byref(n_nodes)) == 0
    assert n_nodes.value >= 4  # counting, code + sizes, placement, encode (or their separate-pass twins)
    assert hip.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, 0) == 0
    free1, tot1 = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(free1), ctypes.byref(tot1)) == 0
    try:
        for data in inputs[1:]:
            raw.copy_(data)
            comp.fill_(0xEE)
            torch.cuda.synchronize()
            assert hip.hipGraphLaunch(gexec, st.cuda_stream) == 0
            st.synchronize()
            want = ref.compress(data)
            torch.cuda.synchronize()
            assert status.cpu().tolist() == [0] * ref.n_frames
            assert torch.equal(off, want.comp_off) and torch.equal(size, want.comp_size)
            end_of_frames = int(off[-1].item()) + int(size[-1].item())
            assert torch.equal(comp[:end_of_frames], want.comp[:end_of_frames])
    finally:
        hip.hipGraphExecDestroy(gexec)
        hip.hipGraphDestroy(graph)
    # the captured call took no device memory (instantiating the graph is the caller's business)
    assert free1.value >= free0.value - (8 << 20), (free0.value, free1.value)
