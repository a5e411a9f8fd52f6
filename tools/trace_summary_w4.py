"""Summarise an ML_DENSE_TRACE file written by the w4 kernel (-DML_BRINGUP -DML_DENSE_TRACE build): per launch and tile,
mean cycle counts of: the first 8 phases, the rest of the main loop, the stream drain, the epilogue, the gap to the next tile."""
import sys, struct
import numpy as np
data = open(sys.argv[1], 'rb').read()
off, li = 0, 0
while off < len(data):
    hdr = struct.unpack('8q', data[off:off + 64]); off += 64
    n = hdr[6]
    a = np.frombuffer(data[off:off + 8 * n], dtype=np.uint64).reshape(-1, 8, 64).astype(np.int64); off += 8 * n
    grid = hdr[1]
    a = a[:grid, :4]                               # 4 waves per workgroup
    print("launch %d: grid %d M %d N %d K %d res %d" % (li, grid, hdr[2], hdr[3], hdr[4], hdr[5]))
    for tile in range(4):
        s = a[:, :, tile * 16:tile * 16 + 12]
        ok = (s[:, :, [0, 9, 10, 11]] > 0).all(axis=2)
        if not ok.any():
            break
        s = s[ok]
        ph = np.diff(s[:, 0:9], axis=1)            # phases 1..8 (phase 0's end is stamp 1)
        nxt = a[:, :, (tile + 1) * 16][ok] if tile < 3 else None
        line = "  tile %d: phases " % tile + " ".join("%5d" % v for v in ph.mean(0))
        line += " | rest of loop %7d | drain %6d | epilogue %6d" % ((s[:, 9] - s[:, 8]).mean(), (s[:, 10] - s[:, 9]).mean(), (s[:, 11] - s[:, 10]).mean())
        if nxt is not None and (nxt > 0).any():
            line += " | to next tile start %5d" % (nxt[nxt > 0] - s[:, 11][nxt > 0]).mean()
        line += " | tile total %7d" % (s[:, 11] - s[:, 0]).mean()
        print(line)
    li += 1
    if li >= int(sys.argv[2]) if len(sys.argv) > 2 else 16:
        break
