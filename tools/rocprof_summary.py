"""Turn a rocprofv3 results .db (rocpd sqlite) into the per-kernel stats table we commit under profiles/."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
        "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["# rocprofv3 --kernel-trace --stats summary of %s" % db_path,
             "%-78s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct")]
    for name, n, tot, avg, mn, mx in rows:
        lines.append("%-78s %8d %14d %12.0f %12d %12d %6.2f%%" % (name[:78], n, tot, avg, mn, mx, 100.0 * tot / total))
    # per-grid breakdown of the dense kernel (the K=64 first layer vs the 1024x1024 layers)
    lines.append("")
    lines.append("# dense kernel by grid size / kernarg (first layer K=64 is the short one)")
    for name, gx, n, avg, mn, mx in cur.execute(
            "select name, grid_x, count(*), avg(duration), min(duration), max(duration) from kernels "
            "where name like '%dense_kernel%' group by name, grid_x").fetchall():
        lines.append("%-60s grid_x=%-8d calls=%-5d avg_ns=%-10.0f min=%-9d max=%d" % (name[:60], gx, n, avg, mn, mx))
    lines.append("")
    lines.append("# exact-fp32 GEMM (training) by grid")
    try:
        for name, gx, gy, gz, n, avg, mn, mx in cur.execute(
                "select name, grid_x, grid_y, grid_z, count(*), avg(duration), min(duration), max(duration) from kernels "
                "where name like '%sgemm_kernel%' group by grid_x, grid_y, grid_z").fetchall():
            lines.append("%-40s grid=(%d,%d,%d) calls=%-5d avg_ns=%-10.0f min=%-9d max=%d" % (name[:40], gx, gy, gz, n, avg, mn, mx))
    except sqlite3.Error as e:
        lines.append("(n/a: %s)" % e)
    lines.append("")
    lines.append("# mid-route training GEMM (xgemm) by layout and grid: (N/64, row tiles of 32)")
    try:
        for name, gx, gy, n, avg, mn, mx in cur.execute(
                "select name, grid_x, grid_y, count(*), avg(duration), min(duration), max(duration) from kernels "
                "where name like '%xgemm_kernel%' group by name, grid_x, grid_y").fetchall():
            lines.append("%-40s grid=(%d,%d) calls=%-5d avg_ns=%-10.0f min=%-9d max=%d" % (name[:40], gx, gy, n, avg, mn, mx))
    except sqlite3.Error as e:
        lines.append("(n/a: %s)" % e)
    # device busy fraction: summed kernel time / (last end - first start) over the second half of the trace (warm)
    try:
        ks = cur.execute("select start, end from kernels order by start").fetchall()
        if len(ks) > 20:
            half = ks[len(ks) // 2:]
            span = half[-1][1] - half[0][0]
            busy = sum(e - s0 for s0, e in half)
            gaps = [half[i + 1][0] - half[i][1] for i in range(len(half) - 1)]
            gaps.sort()
            lines.append("")
            lines.append("# second half of the trace: %d launches, span %.3f ms, kernel time %.3f ms (busy %.1f %%), median gap %.2f us, "
                         "p90 gap %.2f us" % (len(half), span / 1e6, busy / 1e6, 100.0 * busy / max(span, 1), gaps[len(gaps) // 2] / 1e3,
                                              gaps[int(len(gaps) * 0.9)] / 1e3))
    except sqlite3.Error as e:
        lines.append("(timeline n/a: %s)" % e)
    txt = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, 'w').write(txt)
    print(txt)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
