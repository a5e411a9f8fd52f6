"""profiles/rNN_traffic.json from a PMC summary (tools/pmc_summary.py output of `bench.py --steps 6 --warmup 2 --no-extra` under the
separate rocprofv3 --pmc passes) [+ the kernel-stats summary of the same command]: per dense launch of the headline step the HBM
bytes (FETCH_SIZE / WRITE_SIZE: both in KiB, FETCH_SIZE doubled -- the gfx950 correction of MI355X_MICROARCH.md), and, when the
SQ pass and the kernel stats are there, the MFMA-busy share and the HBM rate bench.py replays as roofline.mfma_busy / hbm_gbps.
  python tools/traffic_from_pmc.py profiles/r04_pmc.txt profiles/r04_traffic.json [profiles/r04_kernel_stats.txt]
bench.py imports `record()` / `counters_of_db()` / `durations_of_db()` for its live leg (the same arithmetic on the rocprofv3 result databases of
child runs it starts itself)."""
import json, re, sys

WANT = ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'TCC_HIT_sum', 'TCC_MISS_sum',
        'SQ_WAIT_INST_LDS', 'SQ_WAVE_CYCLES')
# the 8 dense launches of a step: which kernel instantiation (name prefix up to the HEAD parameter) runs how often
ROLES = [('pp_k64', r'void mlk::dense_kernel_pp<3, true, false, 0[,>]', 1), ('w4_plain', r'void mlk::dense_kernel_w4<3, true, false, 0[,>]', 3),
         ('w4_res', r'void mlk::dense_kernel_w4<3, true, true, 0[,>]', 2), ('w4_res_aux', r'void mlk::dense_kernel_w4<3, true, true, -1[,>]', 1),
         ('pp_head', r'void mlk::dense_kernel_pp<3, true, false, 8[,>]', 1)]
ALGORITHMIC_BYTES_PER_LAUNCH = 574619648


def counters_of_text(path):
    """{(kernel name (50 chars), counter): (dispatches, mean value)} from a tools/pmc_summary.py text."""
    vals = {}
    for line in open(path):
        m = re.match(r'(.{50}) (\S+)\s+n=(\d+)\s+avg=([\d.]+)', line)
        if m and m.group(2) in WANT:
            vals[(m.group(1).strip(), m.group(2))] = (int(m.group(3)), float(m.group(4)))
    return vals


def counters_of_db(db_path, vals=None):
    """The same dictionary straight from a rocprofv3 --pmc result database (rocpd sqlite, view counters_collection)."""
    import sqlite3
    vals = {} if vals is None else vals
    cur = sqlite3.connect(db_path).cursor()
    for name, counter, n, avg in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                             "group by kernel_name, counter_name"):
        if counter in WANT:
            vals[(name[:50].strip(), counter)] = (int(n), float(avg))
    return vals


def durations_of_text(path):
    dur = {}
    for line in open(path):
        m = re.match(r'(void mlk::dense_kernel_\S+<[^>]*>)\(mlk::DenseParams\)\s+(\d+)\s+(\d+)\s+(\d+)', line)
        if m:
            dur[m.group(1)] = float(m.group(4))   # avg_ns
    return dur


def durations_of_db(db_path):
    """{kernel name: (launches, mean ns)} of a rocprofv3 --kernel-trace result database."""
    import sqlite3
    cur = sqlite3.connect(db_path).cursor()
    return {name: (int(n), float(avg)) for name, n, avg in cur.execute("select name, count(*), avg(duration) from kernels group by name")}


def record(vals, dur, src, stats_src):
    """The traffic record (per dense launch of the headline step) from counter means `vals` and kernel durations `dur` (name -> mean ns)."""
    def get(pattern, counter):
        for (name, c), v in vals.items():
            if c == counter and re.match(pattern, name):
                return v
        return None

    rd = wr = busy_num = busy_den = t_ns = 0.0
    per_r, per_w, per_busy, per_ns, hits, miss = {}, {}, {}, {}, 0.0, 0.0
    for role, pat, count in ROLES:
        f, w = get(pat, 'FETCH_SIZE'), get(pat, 'WRITE_SIZE')
        if not f or not w:
            raise KeyError('missing counters for %s (%s)' % (role, pat))
        r_b, w_b = f[1] * 2 * 1024, w[1] * 1024
        per_r[role], per_w[role] = round(r_b / 1e6, 1), round(w_b / 1e6, 1)
        rd += r_b * count
        wr += w_b * count
        h, ms = get(pat, 'TCC_HIT_sum'), get(pat, 'TCC_MISS_sum')
        if h and ms:
            hits += h[1] * count
            miss += ms[1] * count
        mf, gr = get(pat, 'SQ_VALU_MFMA_BUSY_CYCLES'), get(pat, 'GRBM_GUI_ACTIVE')
        if mf and gr:
            # SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs; GRBM_GUI_ACTIVE the active cycles of the 8 XCDs
            per_busy[role] = round(mf[1] / (gr[1] / 8 * 1024), 4)
            busy_num += mf[1] * count
            busy_den += gr[1] / 8 * 1024 * count
        for name, ns in dur.items():
            if re.match(pat, name + '>' if not name.endswith('>') else name) or re.match(pat, name):
                per_ns[role] = ns
                t_ns += ns * count
    out = {"kernel": "the 8 dense launches of a step: dense_kernel_pp<3,true,false,0> (K=64 input layer), 3 x dense_kernel_w4<3,true,false,0>, 2 x "
                     "dense_kernel_w4<3,true,true,0> (residual), dense_kernel_w4<3,true,true,-1> (residual + fused w_aux head), "
                     "dense_kernel_pp<3,true,false,8> (fused output head)",
           "source": src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ passes, separate runs of `bench.py --steps 6 --warmup 2 --no-extra`, KiB units, "
                     "FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md); launch-weighted mean over the 8 dense launches of a step",
           "launches_per_step": 8, "hbm_read_bytes_per_launch": int(rd / 8), "hbm_write_bytes_per_launch": int(wr / 8),
           "hbm_bytes_per_launch": int((rd + wr) / 8), "algorithmic_bytes_per_launch": ALGORITHMIC_BYTES_PER_LAUNCH,
           "per_kernel_read_MB": per_r, "per_kernel_write_MB": per_w}
    if hits + miss > 0:
        out["l2_hit_rate"] = round(hits / (hits + miss), 3)
    if busy_den > 0:
        out["mfma_busy"] = round(busy_num / busy_den, 4)
        out["mfma_busy_per_kernel"] = per_busy
        out["mfma_busy_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), weighted over the 8 launches of a step"
    if t_ns > 0 and len(per_ns) == len(ROLES):
        out["avg_launch_us_under_rocprof"] = {k: round(v / 1e3, 1) for k, v in per_ns.items()}
        out["hbm_gbps"] = round((rd + wr) / t_ns, 1)   # bytes / ns = GB/s
        out["hbm_gbps_note"] = "HBM bytes of the 8 dense launches / their summed average durations (" + str(stats_src) + "); peak ~8000"
    return out


if __name__ == '__main__':
    src, dst = sys.argv[1], sys.argv[2]
    stats = sys.argv[3] if len(sys.argv) > 3 else None
    try:
        out = record(counters_of_text(src), durations_of_text(stats) if stats else {}, src, stats)
    except KeyError as exc:
        sys.exit(str(exc))
    json.dump(out, open(dst, 'w'), indent=1)
    print(json.dumps(out)[:900])
