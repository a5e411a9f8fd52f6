"""profiles/rNN_traffic.json from a pmc summary (tools/pmc_summary.py output of `bench.py --steps 6 --warmup 2 --no-extra` under
the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes): HBM bytes per dense launch of the headline step.  Units and the
gfx950 correction as MI355X_MICROARCH.md prescribes: both counters in KiB, FETCH_SIZE doubled.
  python tools/traffic_from_pmc.py profiles/r03_pmc.txt profiles/r03_traffic.json"""
import json, re, sys

src, dst = sys.argv[1], sys.argv[2]
vals = {}
for line in open(src):
    m = re.match(r'(.{50}) (\S+)\s+n=(\d+)\s+avg=([\d.]+)', line)
    if m and m.group(2) in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'TCC_HIT_sum', 'TCC_MISS_sum'):
        vals[(m.group(1).strip(), m.group(2))] = (int(m.group(3)), float(m.group(4)))
# the 8 dense launches of a step: which kernel instantiation runs how often
roles = [('pp_k64', 'void mlk::dense_kernel_pp<3, true, false, 0>(mlk::', 1), ('w4_plain', 'void mlk::dense_kernel_w4<3, true, false, 0>(mlk::', 3),
         ('w4_res', 'void mlk::dense_kernel_w4<3, true, true, 0>(mlk::D', 2), ('w4_res_aux', 'void mlk::dense_kernel_w4<3, true, true, -1>(mlk::', 1),
         ('pp_head', 'void mlk::dense_kernel_pp<3, true, false, 8>(mlk::', 1)]
rd = wr = 0.0
per_r, per_w, hits, miss = {}, {}, 0.0, 0.0
for role, name, count in roles:
    f = vals.get((name, 'FETCH_SIZE'))
    w = vals.get((name, 'WRITE_SIZE'))
    if not f or not w:
        sys.exit('missing counters for %s (%s)' % (role, name))
    r_b, w_b = f[1] * 2 * 1024, w[1] * 1024
    per_r[role], per_w[role] = round(r_b / 1e6, 1), round(w_b / 1e6, 1)
    rd += r_b * count
    wr += w_b * count
    h, ms = vals.get((name, 'TCC_HIT_sum')), vals.get((name, 'TCC_MISS_sum'))
    if h and ms:
        hits += h[1] * count
        miss += ms[1] * count
out = {"kernel": "the 8 dense launches of a step: dense_kernel_pp<3,true,false,0> (K=64 input layer), 3 x dense_kernel_w4<3,true,false,0>, 2 x "
                 "dense_kernel_w4<3,true,true,0> (residual), dense_kernel_w4<3,true,true,-1> (residual + fused w_aux head), "
                 "dense_kernel_pp<3,true,false,8> (fused output head)",
       "source": src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `bench.py --steps 6 --warmup 2 --no-extra`, KiB units, "
                 "FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md); launch-weighted mean over the 8 dense launches of a step",
       "launches_per_step": 8, "hbm_read_bytes_per_launch": int(rd / 8), "hbm_write_bytes_per_launch": int(wr / 8),
       "hbm_bytes_per_launch": int((rd + wr) / 8), "algorithmic_bytes_per_launch": 574619648,
       "per_kernel_read_MB": per_r, "per_kernel_write_MB": per_w}
if hits + miss > 0:
    out["l2_hit_rate"] = round(hits / (hits + miss), 3)
json.dump(out, open(dst, 'w'), indent=1)
print(json.dumps(out)[:600])
