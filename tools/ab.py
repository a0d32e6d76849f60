"""Diagnostic: kernel timings of several builds of libhfz.so on the same batches.

usage: ab.py LIB[,LIB...] K:total_mib:frame_kib:source [...]
Each library is loaded in a child process of its own.  Not part of the product."""
import importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(lib, cfgs):
    import torch
    hfz = importlib.import_module("huffman-avx512_amd")
    hfz.LIB_PATH = os.path.abspath(lib)
    hfz.lib()
    if os.environ.get("HFZ_FPW"):
        hfz.set_frames_per_wave(int(os.environ["HFZ_FPW"]))
    if os.environ.get("HFZ_ENCBLK"):
        hfz.set_encoder_block(int(os.environ["HFZ_ENCBLK"]))
    if os.environ.get("HFZ_STORE"):
        hfz.set_decoder_store(int(os.environ["HFZ_STORE"]))
    if os.environ.get("HFZ_PARTS"):
        hfz.set_encoder_parts(int(os.environ["HFZ_PARTS"]))
    if os.environ.get("HFZ_BITS"):
        hfz.set_decoder_table_bits(int(os.environ["HFZ_BITS"]))
    for cfg in cfgs:
        K, mib, fk, src = cfg.split(":")
        K, total, fb = int(K), int(mib) << 20, int(fk) << 10
        raw = hfz.generate(hfz.distributions.SOURCES[src](), 1000, total)
        codec = hfz.BatchCodec(K, total, fb)
        out = torch.empty(total, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            b = codec.compress(raw); codec.decompress(b, out)
        torch.cuda.synchronize()
        hfz.timing_enable(True)
        for _ in range(10):
            b = codec.compress(raw); codec.decompress(b, out)
        torch.cuda.synchronize()
        kt = hfz.timing_read(); hfz.timing_enable(False)
        ok = bool(torch.equal(out, raw))
        print("%-28s %-18s hist %.4f code %.4f sizes %.4f enc %.4f dec %.4f  sum %.4f ms  ok=%s" % (
            os.path.basename(lib), cfg, kt["hist_ms"], kt["table_ms"], kt["sizes_ms"], kt["encode_ms"], kt["decode_ms"],
            sum(kt[k + "_ms"] for k in ("hist", "table", "sizes", "place", "encode", "decode")), ok),
            flush=True)
        del codec, raw, out, b


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3:])
    else:
        for lib in sys.argv[1].split(","):
            rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--child", lib] + sys.argv[2:])
            if rc != 0:
                sys.exit(rc)
