"""Frame size / frames per wave / table bits sweep of a bench workload: per-kernel ms.
usage: cfg_sweep.py K total_mib source frame_kib[,frame_kib...] [fpw list] [bits list]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
hfz = importlib.import_module("huffman-avx512_amd")
K, total_mib, source = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
frames = [int(a) for a in sys.argv[4].split(",")]
fpws = [int(a) for a in (sys.argv[5] if len(sys.argv) > 5 else "0").split(",")]
bitss = [int(a) for a in (sys.argv[6] if len(sys.argv) > 6 else "0").split(",")]
blks = [int(a) for a in (sys.argv[7] if len(sys.argv) > 7 else "0").split(",")]
dev = torch.device("cuda", 0)
for fk in frames:
    w = bench.Workload(hfz, dev, 0, 1, K, total_mib, fk, source, False)
    for fpw in fpws:
        for bits, blk in [(b, k) for b in bitss for k in blks]:
            hfz.set_frames_per_wave(fpw)
            hfz.set_decoder_table_bits(bits)
            hfz.set_decoder_input_block(blk)
            elapsed, batch, kt = w.timed(8, 2, 1)
            print("K=%d %d MiB %s frame %d KiB fpw=%d bits=%d blk=%d: step %.4f ms hist %.4f table %.4f encode %.4f decode %.4f (frac %.3f)" % (
                K, total_mib, source, fk, fpw, bits, blk, 1e3 * elapsed / 8, kt["hist_ms"], kt["table_ms"], kt["encode_ms"], kt["decode_ms"],
                (w.total + batch.compressed_bytes()) / (kt["decode_ms"] * 1e-3) / 8e12), flush=True)
    hfz.set_frames_per_wave(0)
    hfz.set_decoder_table_bits(0)
    hfz.set_decoder_input_block(0)
    del w
    torch.cuda.empty_cache()
