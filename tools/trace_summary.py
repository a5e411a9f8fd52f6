"""Summarise an ML_DENSE_TRACE file: per launch, mean durations of the tile segments (in cycles of s_memtime)."""
import sys, struct
import numpy as np
data = open(sys.argv[1], 'rb').read()
off = 0
li = 0
while off < len(data):
    hdr = struct.unpack('8q', data[off:off + 64]); off += 64
    n = hdr[6]
    a = np.frombuffer(data[off:off + 8 * n], dtype=np.uint64).reshape(-1, 8, 64).astype(np.int64); off += 8 * n
    grid = hdr[1]
    a = a[:grid]
    t0 = a[:, :, 0].min()
    tend = a.max()
    print("launch %d: grid %d M %d N %d K %d res %d  total %d ticks" % (li, grid, hdr[2], hdr[3], hdr[4], hdr[5], tend - t0))
    ntile = 0
    for tile in range(8):
        base = 1 + tile * 6
        if base + 5 >= 64 or not (a[:, :, base + 5] > 0).any():
            break
        seg = a[:, :, base:base + 6]
        ok = (seg > 0).all(axis=2)
        for g, name in ((slice(0, 4), 'G0'), (slice(4, 8), 'G1')):
            s = seg[:, g][ok[:, g]]
            if len(s) == 0: continue
            prev_end = a[:, g, base - 1][ok[:, g]]
            d = np.diff(s, axis=1)
            print("  tile %d %s: start@%7d  wait-stage0 %6d | barrier %6d | kstep0 %6d | kstep1 %6d | rest-of-mainloop %7d | epilogue %6d   (start spread %d)" % (
                tile, name, (s[:, 0] - t0).mean(), (s[:, 0] - prev_end).mean(), d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d[:, 3].mean(), d[:, 4].mean(), s[:, 0].max() - s[:, 0].min()))
        ntile += 1
    li += 1
