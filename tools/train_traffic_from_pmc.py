"""HBM traffic of one training step from the PMC summary of tools/prof_train.py under rocprofv3 (tools/gpu_run.sh trainpmc:<rows>):
per kernel launches x (FETCH_SIZE x 2 + WRITE_SIZE) KiB (the gfx950 correction of MI355X_MICROARCH.md), the SQ pass' MFMA-busy share of
the GEMM kernels, and the step's algorithmic minimum for comparison.
  python tools/train_traffic_from_pmc.py profiles/r04_train_pmc_rows65536.txt 6 65536 profiles/r04_train_traffic_rows65536.json"""
import json, re, sys
src, steps, rows, dst = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
vals = {}
for line in open(src):
    m = re.match(r'(.{50}) (\S+)\s+n=(\d+)\s+avg=([\d.]+)', line)
    if m:
        vals.setdefault(m.group(1).strip(), {})[m.group(2)] = (int(m.group(3)), float(m.group(4)))
per = {}
tot_r = tot_w = 0.0
for k, c in vals.items():
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        n = c['FETCH_SIZE'][0]
        r, w = c['FETCH_SIZE'][1] * 2 * 1024 * n / steps, c['WRITE_SIZE'][1] * 1024 * n / steps
        tot_r += r
        tot_w += w
        if (r + w) > 20e6:
            rec = {"launches_per_step": round(n / steps, 2), "read_MB_per_step": round(r / 1e6, 1), "write_MB_per_step": round(w / 1e6, 1)}
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c and c['SQ_VALU_MFMA_BUSY_CYCLES'][1] > 0:
                rec["mfma_busy"] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'][1] / (c['GRBM_GUI_ACTIVE'][1] / 8 * 1024), 4)
            per[k] = rec
H = 1024
act = rows * H * 4
# algorithmic minimum of the step as it is organised (fp32 tensors that MUST cross HBM once: 8 pre-activations z written + read twice
# (forward normalise, backward), 8 activations written + read (next GEMM, weight gradient), 8 gradients written + read): ~ 56 x m x H x 4
out = {"rows": rows, "steps_profiled": steps, "hbm_read_GB_per_step": round(tot_r / 1e9, 3), "hbm_write_GB_per_step": round(tot_w / 1e9, 3),
       "hbm_GB_per_step": round((tot_r + tot_w) / 1e9, 3), "one_activation_matrix_MB": round(act / 1e6, 1),
       "activation_matrix_crossings_per_step": round((tot_r + tot_w) / act, 1),
       "source": src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ passes of tools/prof_train.py; KiB units, FETCH_SIZE doubled)",
       "per_kernel": dict(sorted(per.items(), key=lambda kv: -(kv[1]['read_MB_per_step'] + kv[1]['write_MB_per_step'])))}
json.dump(out, open(dst, 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != 'per_kernel'}))
