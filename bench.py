#!/usr/bin/env python3
"""bench.py -- persons/s of the monoloco keypoint->3D hot path on N MI355X.

A "step" is one pass of the whole device pipeline over one batch of synthetic keypoints that is
already resident in HBM:  pre-process (pixel_to_camera) -> MonoLoco++ residual MLP (34->1024->9)
-> post-process (extract_outputs + back-projection), and for N > 1 the single RCCL gather of the
(m,5) (x,y,z,d,sigma) block to rank 0.  Workload = BASELINE.json configs[1]: batch 65536 persons
per GPU ("weak" scaling: per-GPU work fixed, rows sharded, no data-path collective but the
final gather).  `--total-rows 1048576` is BASELINE configs[3] (a fixed total cut into per-rank shards:
"strong").  Prints ONE JSON line on rank 0.

  python bench.py                                   # 1 GPU
  python bench.py --gpus N                          # launches its own N ranks (torch.distributed.run, 127.0.0.1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W      # the driver's form: same ranks, same line

N > 1 lines carry `ranks_seen` (distinct (rank, device UUID) pairs), `ranks` (per-rank device, rows and its own ms per
step), `ms_per_step_{min,max}_rank`, `gather_ms` (the collective alone), `gather_check` (per-shard checksums of the gathered
block against each rank's own) and `config4_strong` (BASELINE configs[3] in the same launch).

The line proves itself: `parity` is the deviation of the very outputs the timed loop produced from the CPU oracle on a
strided sample; `e2e_ms` adds the pinned host<->device copies; `extra` (N = 1) times the other BASELINE configs
(stereo 32768 pair rows, training step at 331 and 65536 rows, a 16-person frame) and the pure-bf16 comparison mode with
its measured deviation; `cpu_baseline` is a thread sweep of the oracle on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# algorithmic work of the reference layer structure, 2 FLOP/MAC, biases/BN/activations excluded
# (SURVEY.md 8d): mono++ 34->1024->9 and MonStereo 68->1024->10
FLOP_PER_ROW = {'mono': 16865280, 'stereo': 16936960}
PEAK_TFLOPS_F16_DENSE = 2500.0  # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_TFLOPS_F32_MFMA = 157.3
METRIC = "persons/sec (17-kp MLP forward+postproc) at batch 65536, 1/2/4/8 MI355X"


def cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def host_threads():
    """Threads worth giving torch on this box: the cores this process may run on, capped by the container's cgroup CPU quota
    (a box can show 256 cores and grant 16: torch's default of one thread per visible core then runs several times slower)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(sd, kps_np, kk, budget_s):
    """The CPU path timed on this box's host cores (BASELINE.md section 3 protocol: thread sweep, >= 3 warm-ups,
    >= 5 repetitions, median): the oracle -- the reference itself does not exist on the GPU box -- a
    torch-CPU fp32 restatement of preprocess_monoloco -> LocoModel eval forward -> extract_outputs -> back-projection
    (oracle/monoloco_oracle.py).  Bounded sample of the same synthetic batch."""
    import statistics
    import torch
    from oracle import monoloco_oracle as O
    ncpu = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = ncpu
    quota = None
    try:  # a container may be capped well below the cores it can see (cgroup v2 cpu.max = "<quota> <period>")
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    sd_t = {k: torch.tensor(v) for k, v in sd.items()}
    results = {}
    t_start = time.perf_counter()
    prev = torch.get_num_threads()

    def measure(t):
        n = min(2048 if t <= 2 else (8192 if t <= 8 else 16384), len(kps_np))
        kps = torch.tensor(kps_np[:n])
        torch.set_num_threads(t)
        for _ in range(3):
            O.forward_mono(sd_t, kps, kk)
        times = []
        for _ in range(5):
            t0 = time.perf_counter()
            O.forward_mono(sd_t, kps, kk)
            times.append(time.perf_counter() - t0)
        results[t] = n / statistics.median(times)

    # ascending powers of two up to the usable cores: the sweep stops once it is past the knee (oversubscribed or
    # quota-capped boxes get SLOWER with more threads) or out of budget; the 1-thread point always exists
    t = 1
    while t <= usable:
        measure(t)
        if results[t] < 0.8 * max(results.values()) or time.perf_counter() - t_start > budget_s:
            break
        t *= 2
    torch.set_num_threads(prev)
    best_t = max(results, key=results.get)
    # the port against the REAL reference on the build container's CPU (tools/cpu_port_vs_reference.py: same threads, same batch,
    # outputs bit-identical): a committed calibration -- the reference itself does not exist on the GPU box
    pvr, pvr_src, pvr_host = None, None, None
    for name in ('r06_port_vs_reference.json', 'r05_port_vs_reference.json', 'r04_port_vs_reference.json'):
        path = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(path):
            rec = json.load(open(path))
            pvr, pvr_src = rec.get("port_vs_reference"), "replayed:profiles/%s (%s, %s host cores; per thread count: %s)" % (
                name, rec.get("cpu_model"), rec.get("host_cores"),
                {k: v.get("port_vs_reference") for k, v in rec.get("by_threads", {}).items()})
            pvr_host = "%s, %s host cores: the BUILD container, not the box this line was measured on" % (rec.get("cpu_model"), rec.get("host_cores"))
            break
    return {"value": round(results[best_t], 1), "unit": "persons/s", "cores": best_t, "kind": "port",
            "port_vs_reference": pvr, "port_vs_reference_measured_on": pvr_host, "port_vs_reference_source": pvr_src,
            "reference_estimate": round(results[best_t] / pvr, 1) if pvr else None,
            "one_thread": round(results.get(1, 0.0), 1), "host_cores": ncpu, "usable_cores": usable,
            "cgroup_cpu_quota": quota, "cpu_model": cpu_model(),
            "sweep": {str(t): round(v, 1) for t, v in sorted(results.items())},
            "sample": "oracle/monoloco_oracle.forward_mono (torch %s CPU fp32) on the first 2048 / 8192 / 16384 persons "
                      "(<= 2 / <= 8 / more threads) of the same synthetic batch; ascending thread sweep, per thread count "
                      "3 warm-ups + 5 repetitions, median; value = best of the sweep; %.1f s" % (torch.__version__, time.perf_counter() - t_start)}


def traffic_from_profiles(args):
    """HBM bytes per dense launch (and the MFMA-busy share / HBM rate of the same launches) from the PMC passes committed under
    profiles/ (tools/traffic_from_pmc.py: FETCH_SIZE x2 + WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES over GRBM_GUI_ACTIVE, collected
    with rocprofv3 in separate runs of this very command; PMC cannot be read live).  Returns (record, source) or None if the
    committed measurement does not cover this configuration."""
    if args.workload != 'mono' or args.precision != 'f16x2' or args.batch != 65536 or args.no_merge or args.total_rows:
        return None
    for name in ('r06_traffic.json', 'r05_traffic.json', 'r04_traffic.json', 'r03_traffic.json', 'r02_traffic.json', 'r01_traffic.json'):
        path = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(path):
            return json.load(open(path)), "replayed:profiles/" + name
    return None


def live_counters(args, budget_s=150.0):
    """The roofline counters as a product of THIS run: re-executes the headline command in child processes under rocprofv3 --
    one `--kernel-trace --stats` run (kernel durations) and three separate `--kernel-trace --pmc` passes (SQ_VALU_MFMA_BUSY_CYCLES
    GRBM_GUI_ACTIVE | FETCH_SIZE | WRITE_SIZE: FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC
    slots") -- and reduces the result databases with tools/traffic_from_pmc.py (KiB units, FETCH_SIZE doubled: the guide's gfx950
    correction).  Fails soft: returns {"source": "rocprofv3 unavailable: ..."} and the line keeps its replayed values."""
    import glob
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import traffic_from_pmc as T
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return {"source": "rocprofv3 unavailable: not on PATH, not under /opt/rocm/bin"}
    if os.environ.get('ROCP_TOOL_LIBRARIES') or 'rocprofiler' in os.environ.get('LD_PRELOAD', ''):
        # (this process is itself being profiled: a profiler inside a profiler would fight over the counters)
        return {"source": "rocprofv3 unavailable: this run is already under a profiler (ROCP_TOOL_LIBRARIES / LD_PRELOAD)"}
    t_start = time.perf_counter()
    work = tempfile.mkdtemp(prefix='ml_live_', dir='/tmp')
    child = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6', '--warmup', '2', '--no-extra', '--cpu-seconds', '0',
             '--no-live-counters', '--no-parity', '--batch', str(args.batch), '--precision', args.precision, '--weights', args.weights]
    if args.no_merge:
        child.append('--no-merge')
    env = dict(os.environ, TMPDIR='/tmp', PYTHONUNBUFFERED='1')
    passes = [('stats', ['--kernel-trace', '--stats'], []),
              ('sq', ['--kernel-trace', '--pmc', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'], ['--no-profile']),
              ('fetch', ['--kernel-trace', '--pmc', 'FETCH_SIZE'], ['--no-profile']),
              ('write', ['--kernel-trace', '--pmc', 'WRITE_SIZE'], ['--no-profile'])]
    dbs, took = {}, {}
    try:
        for name, flags, extra in passes:
            left = budget_s - (time.perf_counter() - t_start)
            if left < 15:
                return {"source": "rocprofv3 unavailable: pass '%s' not started, %.0f s budget spent (%s)" % (name, budget_s, took)}
            t0 = time.perf_counter()
            out_dir = os.path.join(work, name)
            try:
                cp = subprocess.run([exe] + flags + ['-d', out_dir, '-o', name, '--'] + child + extra, cwd='/tmp', env=env,
                                    stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=min(left, 90.0))
            except subprocess.TimeoutExpired:
                return {"source": "rocprofv3 unavailable: pass '%s' timed out" % name}
            took[name] = round(time.perf_counter() - t0, 1)
            found = glob.glob(os.path.join(out_dir, '**', '*.db'), recursive=True)
            if cp.returncode != 0 or not found:
                tail = (cp.stderr or b'').decode(errors='replace').strip().splitlines()[-2:]
                return {"source": "rocprofv3 unavailable: pass '%s' rc %d, %d result db (%s)" % (name, cp.returncode, len(found), ' | '.join(tail)[:300])}
            dbs[name] = found[0]
        vals = {}
        for name in ('sq', 'fetch', 'write'):
            T.counters_of_db(dbs[name], vals)
        dur_all = T.durations_of_db(dbs['stats'])
        dense = {k: v for k, v in dur_all.items() if 'dense_kernel' in k}
        rec = T.record(vals, {k.split('(mlk::DenseParams)')[0]: v[1] for k, v in dense.items()},
                       "live: rocprofv3 passes started by this bench.py run", "its own --kernel-trace --stats pass")
        dom = max(dense.items(), key=lambda kv: kv[1][0] * kv[1][1]) if dense else None
        rec["dominant_kernel"] = dom[0].split('(mlk::DenseParams)')[0] if dom else None
        rec["dominant_kernel_avg_us"] = round(dom[1][1] / 1e3, 2) if dom else None
        rec["dominant_kernel_launches"] = dom[1][0] if dom else None
        rec["pass_seconds"] = took
        rec["source"] = ("live: child runs of `bench.py --steps 6 --warmup 2 --no-extra` under rocprofv3 started by this very run -- "
                         "--kernel-trace --stats, then --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE | FETCH_SIZE | WRITE_SIZE in separate passes; "
                         "KiB units, FETCH_SIZE doubled (MI355X_MICROARCH.md gfx950 correction); launch-weighted over the 8 dense launches of a step")
        return rec
    except Exception as exc:   # the counters must never take the headline line down with them
        return {"source": "rocprofv3 unavailable: %s: %s" % (type(exc).__name__, exc)}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def timed_steps(step, steps, warmup, world, device, begin=None, end=None, local_out=None):
    """The timing contract: `warmup` untimed steps, then exactly `steps` steps bracketed by barrier + device
    synchronisation on both sides; returns (seconds = MAX over ranks, whatever `end()` returned on this rank).
    `device` is the rank's HIP device (a CPU device in the gloo tests of this very function).  `local_out` (a dict)
    receives this rank's own seconds under 'local_s' (the per-rank spread bench.py reports for N > 1)."""
    import torch
    import torch.distributed as dist
    on_gpu = device.type == 'cuda'

    def fence():
        if on_gpu:
            torch.cuda.synchronize(device)
        if world > 1 or (dist.is_available() and dist.is_initialized()):   # (a forced world-1 group takes the barrier too)
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize(device)

    for _ in range(warmup):
        step()
    fence()
    if begin is not None:
        begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    extra = end() if end is not None else None
    if local_out is not None:
        local_out['local_s'] = dt
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, extra


def device_identity(dev):
    """What tells two ranks' devices apart: the HIP device's UUID (or PCI bus id) -- 'cpu:<pid>' for the stub."""
    import torch
    if dev.type != 'cuda':
        return "cpu:%d" % os.getpid()
    try:
        p = torch.cuda.get_device_properties(dev)
        uid = getattr(p, 'uuid', None)
        if uid is not None:
            return "%s uuid=%s" % (p.name, uid)
        return "%s pci=%s:%s.%s" % (p.name, getattr(p, 'pci_domain_id', '?'), getattr(p, 'pci_bus_id', '?'),
                                    getattr(p, 'pci_device_id', '?'))
    except Exception as e:  # identity garnish
        return "cuda:%s (%r)" % (dev.index, e)


def power_probe(step, dev, seconds=1.5):
    """Board power and shader clock while the timed workload keeps running (after the timed region; librocm_smi64 through
    ctypes, one sample every ~20 ms from a side thread).  Returns a dict for roofline.power, or None when the SMI library is
    not usable.  Evidence for what bounds the dense kernels: see profiles/r02_ablation.md section 2."""
    import ctypes
    import threading
    import torch
    try:
        smi = ctypes.CDLL('librocm_smi64.so')
    except OSError:
        try:
            smi = ctypes.CDLL('/opt/rocm/lib/librocm_smi64.so')
        except OSError:
            return None

    class Freqs(ctypes.Structure):
        _fields_ = [('has_deep_sleep', ctypes.c_bool), ('num_supported', ctypes.c_uint32), ('current', ctypes.c_uint32),
                    ('frequency', ctypes.c_uint64 * 33)]
    try:
        if smi.rsmi_init(ctypes.c_uint64(0)) != 0:
            return None
        idx = ctypes.c_uint32(dev.index or 0)
        cap = ctypes.c_uint64(0)
        if smi.rsmi_dev_power_cap_get(idx, ctypes.c_uint32(0), ctypes.byref(cap)) != 0:
            cap = ctypes.c_uint64(0)
        watts, mhz = [], []
        stop = threading.Event()

        def sample():
            pw = ctypes.c_uint64(0)
            fr = Freqs()
            while not stop.is_set():
                if smi.rsmi_dev_current_socket_power_get(idx, ctypes.byref(pw)) == 0:
                    watts.append(pw.value / 1e6)
                if smi.rsmi_dev_gpu_clk_freq_get(idx, ctypes.c_int(0), ctypes.byref(fr)) == 0 and fr.current < 33:
                    f = fr.frequency[fr.current] / 1e6
                    if 50.0 < f < 4000.0:
                        mhz.append(f)
                time.sleep(0.02)
        th = threading.Thread(target=sample, daemon=True)
        t_end = time.perf_counter() + seconds
        for _ in range(20):     # get the chip to its steady operating point before sampling
            step()
        torch.cuda.synchronize(dev)
        th.start()
        n = 0
        while time.perf_counter() < t_end:
            for _ in range(10):
                step()
            torch.cuda.synchronize(dev)
            n += 10
        stop.set()
        th.join(timeout=2.0)
        smi.rsmi_shut_down()
        if not watts:
            return None
        watts.sort()
        out = {"socket_power_w_median": round(watts[len(watts) // 2], 1), "socket_power_w_max": round(watts[-1], 1),
               "power_cap_w": round(cap.value / 1e6, 1) if cap.value else None, "samples": len(watts), "steps_while_sampling": n,
               "note": "rocm_smi while the same steps keep running right after the timed region"}
        if mhz:
            mhz.sort()
            out["sclk_mhz_median"] = round(mhz[len(mhz) // 2], 0)
        return out
    except Exception as e:  # measurement garnish: never fail the bench line over it
        return {"error": repr(e)[:200]}


def _ms(fn, iters, warmup, dev, min_warm_s=0.05):
    """Wall-clock ms per call of an asynchronous device call: warm-up, sync, `iters` calls, sync.  The warm-up lasts at least
    `min_warm_s` of device work as well as `warmup` calls: the parity legs run the CPU oracle for seconds between two timed legs, the
    GPU drops to its idle clocks meanwhile, and ten 150-us calls do not bring them back (round 6: the mid-size rows read 8 % low)."""
    import torch
    t_w = time.perf_counter()
    n_w = 0
    while n_w < warmup or time.perf_counter() - t_w < min_warm_s:
        fn()
        n_w += 1
        if n_w % 8 == 0:
            torch.cuda.synchronize(dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / iters * 1e3


def parity_of_timed_run(sd, kps, conf, xyzds, raw, kk, packed=None, chunk=8192, fp64_every=4, every=1):
    """Deviation of the outputs the timed loop just produced from the CPU oracle, on EVERY row of the batch (the oracle in
    chunks of 8192 rows: a couple of seconds for 65536).  Beside the north-star tensor (x, y, z back-projected, d, sigma) and the raw
    network outputs: the spherical z of extract_outputs (process.py:265, z = sqrt(d^2 - x^2 - y^2): a cancellation, so its error
    is the raw error times d / z) against the fp32 oracle, and -- on every 4th row -- against the oracle run in fp64, next to the
    reference arithmetic's OWN fp32-vs-fp64 distance on the same rows (the conditioning, SURVEY 7 hard part 1)."""
    import torch
    from oracle import monoloco_oracle as O
    m = kps.shape[0]
    sd_t = {k: torch.tensor(v) for k, v in sd.items()}
    sd_64 = {k: v.double() for k, v in sd_t.items()}
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(host_threads())
    kps_h, conf_h, xyzds_h, raw_h = kps.cpu()[::every], conf.cpu()[::every], xyzds.cpu()[::every], raw.cpu()[::every]
    z_h = packed[:, 2].cpu()[::every] if packed is not None else None
    m = kps_h.shape[0]
    e_par = e_raw = e_z = e_z64 = ref_z64 = 0.0
    nan_mismatch = rows64 = 0
    t0 = time.perf_counter()
    for lo in range(0, m, chunk):
        hi = min(m, lo + chunk)
        ref = O.forward_mono(sd_t, kps_h[lo:hi], kk, box_conf=conf_h[lo:hi])
        e_par = max(e_par, (xyzds_h[lo:hi] - ref['xyzds']).abs().max().item())
        e_raw = max(e_raw, (raw_h[lo:hi] - ref['raw']).abs().max().item())
        if z_h is not None:
            z_ref = ref['xyzd'][:, 2]
            both = torch.isfinite(z_ref) & torch.isfinite(z_h[lo:hi])
            nan_mismatch += int((torch.isnan(z_ref) != torch.isnan(z_h[lo:hi])).sum())
            if both.any():
                e_z = max(e_z, (z_h[lo:hi] - z_ref)[both].abs().max().item())
            sub = slice(0, hi - lo, fp64_every)
            r64 = O.forward_mono(sd_64, kps_h[lo:hi][sub].double(), kk, dtype=torch.float64)['xyzd'][:, 2]
            ok = torch.isfinite(r64) & torch.isfinite(z_ref[sub]) & torch.isfinite(z_h[lo:hi][sub])
            rows64 += int(ok.sum())
            if ok.any():
                e_z64 = max(e_z64, (z_h[lo:hi][sub].double() - r64)[ok].abs().max().item())
                ref_z64 = max(ref_z64, (z_ref[sub].double() - r64)[ok].abs().max().item())
    torch.set_num_threads(prev_threads)
    res = {"max_abs_xyzds": float('%.3e' % e_par), "max_abs_raw": float('%.3e' % e_raw), "rows_checked": int(m),
           "tolerance": 1e-4, "against": "oracle/monoloco_oracle.forward_mono (torch CPU fp32) on %s of the final timed step's "
           "outputs (%.1f s)" % ("EVERY row" if every == 1 else "every %d-th row" % every, time.perf_counter() - t0)}
    if z_h is not None:
        res["max_abs_z_spherical"] = float('%.3e' % e_z)
        res["z_spherical"] = {
            "rows_vs_fp64": rows64, "ours_vs_fp64": float('%.3e' % e_z64), "reference_fp32_vs_fp64": float('%.3e' % ref_z64),
            "nan_disagreements": nan_mismatch,
            "note": "spherical z = sqrt(d^2 - x^2 - y^2) of extract_outputs (xyzd[:,2]; process.py:265) is NOT the north-star z (that is the "
                    "back-projected one inside max_abs_xyzds); it amplifies the raw deviation by ~d/z, and the reference's own fp32 arithmetic "
                    "sits `reference_fp32_vs_fp64` away from exact on the same rows -- the tests judge it with that conditioning "
                    "(tests/test_gpu_headline.py)"}
    return res


def train_parity(sd_t, x, y, dev):
    """Deviation of ONE training step of the route this batch size takes (dropout 0, no update: the same kernels the timed steps ran,
    minus the dropout masks, which come from a device RNG the CPU cannot replay) from the CPU oracle of the reference's loop body
    (oracle/train_oracle.py: trainer.py:150-161 + losses.py:59-131 under torch autograd) run in fp64, next to the SAME oracle in fp32
    (= the reference's own arithmetic) against that fp64 run: loss values, every train-mode output row, every gradient tensor (worst
    element over the tensor's largest entry; rms error over the tensor's rms).  Bars = tests/test_gpu_train.py::
    test_headline_batch_training_step_against_fp64_oracle: as close to exact as the reference's arithmetic, times a stated factor."""
    import torch
    from monoloco_amd.train import HipTrainer
    from oracle.train_oracle import OracleTrainer
    t0 = time.perf_counter()
    tr = HipTrainer(sd_t, p_dropout=0.0, lr=0.001, device=dev)
    res, out = tr.step(x, y, update=False, want_outputs=True)
    route = tr.last_route
    g = tr.grads()
    out = out.cpu()
    tr.close()
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(host_threads())
    xc, yc = x.cpu(), y.cpu()
    orc = OracleTrainer(sd_t, lr=0.001, dtype=torch.float64)
    r64, out64 = orc.step(xc.double(), yc.double(), update=False)
    g64 = orc.grads()
    del orc
    orc = OracleTrainer(sd_t, lr=0.001)
    r32, out32 = orc.step(xc, yc, update=False)
    g32 = orc.grads()
    del orc
    torch.set_num_threads(prev_threads)
    names = [n for n in ('loss', 'd', 'x', 'y', 'h', 'w', 'l', 'ori') if n in r64]
    e_loss = max(abs(res[n] - r64[n]) / max(1.0, abs(r64[n])) for n in names)
    n_loss = max(abs(r32[n] - r64[n]) / max(1.0, abs(r64[n])) for n in names)
    e_out = float((out.double() - out64).abs().max())
    n_out = float((out32.double() - out64).abs().max())

    def rel(a, ref):
        d = a.double() - ref
        return float(d.abs().max()) / (float(ref.abs().max()) + 1e-300), float(d.pow(2).mean().sqrt()) / (float(ref.pow(2).mean().sqrt()) + 1e-300)
    worst = {"element": (0.0, 0.0, None), "rms": (0.0, 0.0, None)}
    f = lambda v: float('%.3e' % v)
    ok, n_t, failing = True, 0, []
    rows_ = int(x.shape[0])
    # one flipped ReLU mask switches one row's contribution to one unit on or off: that unit's own entries (its BatchNorm bias gradient, its
    # row of the weight gradient) move by a few / rows of the tensor's maximum, everything upstream of it by ~0.5 / rows in rms (measured at
    # 512 rows: one flip in stage 2 -> 7.9e-3 on that entry, 7-10e-4 rms on every tensor below it): the floors scale with the batch
    # (a second draw of the same 512-row batch: one flip in stage 0 -> 1.1e-2 / 1.9e-3; the floors leave room for two flipped units --
    #  a wrong 32 x 64 output tile of any product moves its tensors by >= 3e-2 / 1e-2 at these sizes, tests/test_gpu_train_mid.py)
    el_floor, rms_floor = max(3e-3, 8.0 / rows_), max(1e-4, 1.2 / rows_)
    # a Linear bias in front of a BatchNorm has a mathematically zero gradient (the batch mean absorbs it): rounding noise on every side
    judged = [k for k in g if not (k.endswith('.bias') and 'batch_norm' not in k and not k.startswith(('w_aux', 'w_fin', 'w2.')))]
    ref_dist = {k: rel(g32[k], g64[k]) for k in judged}
    # which units flip differs between any two fp32 runs, so a tensor is also in order when it is no further from exact than the reference's
    # own arithmetic is on its WORST tensor of this very step
    mx32_any, rms32_any = max(v[0] for v in ref_dist.values()), max(v[1] for v in ref_dist.values())
    for k in judged:
        v = g[k]
        mx, rms = rel(v, g64[k])
        mx32, rms32 = ref_dist[k]
        n_t += 1
        if not (mx <= max(3.0 * mx32, el_floor, 1.5 * mx32_any) and rms <= max(4.0 * rms32, rms_floor, 1.5 * rms32_any)):
            ok = False
            failing.append({"tensor": k, "worst_element": f(mx), "fp32_oracle": f(mx32), "rms": f(rms), "fp32_oracle_rms": f(rms32)})
        if mx > worst["element"][0]:
            worst["element"] = (mx, mx32, k)
        if rms > worst["rms"][0]:
            worst["rms"] = (rms, rms32, k)
    ok = ok and e_loss <= 2e-5 and e_out <= 2.0 * n_out + 2e-5
    return {"route": route, "rows": int(x.shape[0]), "ok": bool(ok),
            "max_rel_loss_values_vs_fp64": f(e_loss), "fp32_oracle_loss_values_vs_fp64": f(n_loss),
            "max_abs_outputs_vs_fp64": f(e_out), "fp32_oracle_outputs_vs_fp64": f(n_out),
            "gradients_vs_fp64": {"tensors": n_t,
                                  "worst_element_over_tensor_max": f(worst["element"][0]), "fp32_oracle_same_tensor": f(worst["element"][1]),
                                  "worst_element_tensor": worst["element"][2],
                                  "worst_rms_over_tensor_rms": f(worst["rms"][0]), "fp32_oracle_same_tensor_rms": f(worst["rms"][1]),
                                  "worst_rms_tensor": worst["rms"][2], "tensors_beyond_the_bars": failing},
            "bars": "loss values 2e-5 relative; outputs 2 x the fp32 oracle's own distance from fp64 + 2e-5; per gradient tensor: worst element <= "
                    "max(3 x the fp32 oracle's, %.1e of the tensor's maximum) and rms <= max(4 x the fp32 oracle's, %.1e) -- a ReLU mask of a "
                    "pre-activation within rounding of zero flips between any two fp32 implementations and moves a batch-mean-type entry by 1 / rows "
                    "of its size, so the floors are max(3e-3, 8 / rows) and max(1e-4, 1.2 / rows) (room for two flipped units) -- or, since WHICH units flip differs between any "
                    "two fp32 runs, 1.5 x the fp32 oracle's distance on ITS worst tensor of this step (%.1e / %.1e here); Linear biases in front "
                    "of a BatchNorm (mathematically zero gradient) excluded" % (el_floor, rms_floor, mx32_any, rms32_any),
            "against": "oracle/train_oracle.OracleTrainer (torch CPU autograd) in fp64 and fp32 on the whole batch, dropout 0, one step without "
                       "update (%.1f s)" % (time.perf_counter() - t0)}


def extras(args, dev, sd, eng, kps, conf, kinv, kk, main_ms, main_out=None):
    """The other BASELINE configs and comparison modes, a few timed iterations each (N = 1 only)."""
    import numpy as np
    import torch
    import synth
    from monoloco_amd import engine
    out = {}

    def guarded(name, fn):
        try:
            out[name] = fn()
        except Exception as exc:  # an extra must never take the headline line down with it; the failure stays visible
            out[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        torch.cuda.synchronize(dev)

    m = kps.shape[0]
    sd_t = {k: torch.tensor(v) for k, v in sd.items()}
    same_as_main = ("bit-identical to the outputs of the timed loop's last step, whose deviation from the oracle on every row is the "
                    "line's `parity` block")

    def e2e():
        kps_pin = kps.cpu().pin_memory()
        conf_pin = conf.cpu().pin_memory()
        kps_d, conf_d = torch.empty_like(kps), torch.empty_like(conf)
        o_d = torch.empty((m, 16), dtype=torch.float32, device=dev)
        x_d = torch.empty((m, 5), dtype=torch.float32, device=dev)
        o_pin, x_pin = torch.empty((m, 16)).pin_memory(), torch.empty((m, 5)).pin_memory()

        def call():
            kps_d.copy_(kps_pin, non_blocking=True)
            conf_d.copy_(conf_pin, non_blocking=True)
            eng.forward_mono(kps_d, kinv, box_conf=conf_d, out=o_d, xyzds=x_d)
            o_pin.copy_(o_d, non_blocking=True)
            x_pin.copy_(x_d, non_blocking=True)
            torch.cuda.synchronize(dev)
        # every pinned buffer is touched by a full round trip before anything is timed (first use of freshly pinned pages
        # and the clock ramp made the 10-iteration mean of round 2 swing between 3 and 11 ms), then 40 individually timed
        # iterations: median and minimum
        for _ in range(8):
            call()
        times = []
        for _ in range(40):
            t0 = time.perf_counter()
            call()
            times.append((time.perf_counter() - t0) * 1e3)
        times.sort()
        ms, ms_min = times[len(times) // 2], times[0]
        h2d = (kps_pin.numel() + conf_pin.numel()) * 4
        d2h = (o_pin.numel() + x_pin.numel()) * 4
        expect = main_ms + (h2d + d2h) / 25e9 * 1e3     # the step + both directions at 25 GB/s, nothing overlapped
        res = {"e2e_ms": round(ms, 4), "e2e_ms_min": round(ms_min, 4), "iterations": len(times), "persons_per_s": round(m / ms * 1e3, 1),
               "h2d_bytes": h2d, "d2h_bytes": d2h, "expected_ms_serial_25GBps": round(expect, 4),
               "within_1p3x_of_expected": bool(ms <= 1.3 * expect),
               "note": "pinned H2D of kps+conf, the step, pinned D2H of the (m,16) packed result and the (m,5) parity block, "
                       "device sync per step, median of 40 after 8 warm-up round trips; never reported as `value`"}
        if main_out is not None:    # the round trip over the link returns what the HBM-resident step computed
            res["parity"] = {"same_bits_as_timed_step": bool(torch.equal(o_pin, main_out[0].cpu()) and torch.equal(x_pin, main_out[1].cpu())),
                             "note": same_as_main}
        if ms > 1.3 * expect:
            res["why_slower"] = ("median %.2f ms against %.2f expected: the copies of this box run below 25 GB/s "
                                 "(h2d+d2h alone: %.2f ms at the measured difference)" % (ms, expect, ms - main_ms))
        return res
    guarded("e2e", e2e)

    def e2e_pipelined():
        # the same transfers on their own HIP streams, double-buffered: H2D of batch i+1 and D2H of batch i-1 run while
        # batch i computes (events order the three streams per buffer)
        s_in, s_comp, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        kps_pin = kps.cpu().pin_memory()
        conf_pin = conf.cpu().pin_memory()
        kd = [torch.empty_like(kps) for _ in range(2)]
        cd = [torch.empty_like(conf) for _ in range(2)]
        od = [torch.empty((m, 16), dtype=torch.float32, device=dev) for _ in range(2)]
        xd = [torch.empty((m, 5), dtype=torch.float32, device=dev) for _ in range(2)]
        op = [torch.empty((m, 16)).pin_memory() for _ in range(2)]
        xp = [torch.empty((m, 5)).pin_memory() for _ in range(2)]
        ev_in = [torch.cuda.Event() for _ in range(2)]
        ev_comp = [torch.cuda.Event() for _ in range(2)]
        ev_out = [torch.cuda.Event() for _ in range(2)]

        def run(n):
            for i in range(n):
                b = i & 1
                with torch.cuda.stream(s_in):
                    if i >= 2:
                        s_in.wait_event(ev_comp[b])        # batch i-2 has consumed this input buffer
                    kd[b].copy_(kps_pin, non_blocking=True)
                    cd[b].copy_(conf_pin, non_blocking=True)
                    ev_in[b].record(s_in)
                with torch.cuda.stream(s_comp):
                    s_comp.wait_event(ev_in[b])
                    if i >= 2:
                        s_comp.wait_event(ev_out[b])       # batch i-2's results have left this output buffer
                    eng.forward_mono(kd[b], kinv, box_conf=cd[b], out=od[b], xyzds=xd[b])
                    ev_comp[b].record(s_comp)
                with torch.cuda.stream(s_out):
                    s_out.wait_event(ev_comp[b])
                    op[b].copy_(od[b], non_blocking=True)
                    xp[b].copy_(xd[b], non_blocking=True)
                    ev_out[b].record(s_out)
            torch.cuda.synchronize(dev)
        run(4)
        n = 20
        t0 = time.perf_counter()
        run(n)
        ms = (time.perf_counter() - t0) / n * 1e3
        same = bool(torch.equal(op[0], op[1]))   # both buffers carry the same batch: identical results expected
        par = None
        if main_out is not None:
            par = {"same_bits_as_timed_step": bool(torch.equal(op[0], main_out[0].cpu()) and torch.equal(xp[1], main_out[1].cpu())),
                   "note": same_as_main}
        return {"e2e_pipelined_ms": round(ms, 4), "persons_per_s": round(m / ms * 1e3, 1), "buffers_agree": same, "parity": par,
                "note": "H2D / compute / D2H on three HIP streams, two buffers each; per batch over 20 back-to-back batches"}
    guarded("e2e_pipelined", e2e_pipelined)

    def stereo():
        sd_s = synth.make_state_dict(3, 68, 10, 1024)
        eng_s = engine.LocoEngine({k: torch.tensor(v) for k, v in sd_s.items()}, device=dev, reserve_rows=32768)
        kl = torch.tensor(synth.make_keypoints(256, seed=300)).to(dev)
        kr = torch.tensor(synth.make_keypoints(128, seed=301)).to(dev)
        cf = torch.rand(256, device=dev)
        ms = _ms(lambda: eng_s.forward_stereo(kl, kr, kinv, box_conf=cf), 20, 5, dev)
        # the same call once more with every pair row's raw output kept: all 32768 rows against the oracle, the per-left winner and
        # the (x, y, z, d, sigma) of the winners
        from oracle import monoloco_oracle as O
        got = eng_s.forward_stereo(kl, kr, kinv, box_conf=cf, want_raw_all=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        prev_threads = torch.get_num_threads()
        torch.set_num_threads(host_threads())
        ref = O.forward_stereo({k: torch.tensor(v) for k, v in sd_s.items()}, kl.cpu(), kr.cpu(), kk, box_conf=cf.cpu())
        torch.set_num_threads(prev_threads)
        raw_all = got['raw_all'].cpu()
        err = (raw_all - ref['raw_all']).abs()
        e_cols = float(err[:, :9].max())
        # (the aux logit of a non-matching pair is -50 .. -90 on a trained net: fp32 holds it to 4e-6 relative; judged as its sigmoid too)
        e_aux = float((err[:, 9] - 4e-6 * ref['raw_all'][:, 9].abs()).max())
        e_sig = float((torch.sigmoid(raw_all[:, 9]) - torch.sigmoid(ref['raw_all'][:, 9])).abs().max())
        best_ref = ref['raw_all'].view(256, 128, 10)[:, :, -1].argmax(1)
        same_winner = got['best'].cpu().long() == best_ref
        par = {"rows_checked": 32768, "max_abs_raw_cols_0_8": float('%.3e' % e_cols), "max_abs_aux_logit_minus_4e-6_rel": float('%.3e' % e_aux),
               "max_abs_aux_sigmoid": float('%.3e' % e_sig), "winners_equal": int(same_winner.sum()), "winners": 256,
               "ties_reported": int(got['ties'].item()), "tolerance": 1e-4,
               "against": "oracle/monoloco_oracle.forward_stereo (torch CPU fp32) on EVERY pair row (%.1f s)" % (time.perf_counter() - t0)}
        if 'xyzds' in ref and bool(same_winner.all()):
            par["max_abs_xyzds_of_winners"] = float('%.3e' % float((got['xyzds'].cpu() - ref['xyzds']).abs().max()))
        elif 'xyzds' in ref:   # a near-tie resolved the other way: compare the rows whose winner agrees
            par["max_abs_xyzds_of_winners"] = float('%.3e' % float((got['xyzds'].cpu() - ref['xyzds'])[same_winner].abs().max()))
        eng_s.close()
        return {"config": "BASELINE configs[2]: MonStereo 68->1024->10, 256 x 128 all-vs-all = 32768 pair rows, 1 GPU",
                "ms_per_step": round(ms, 4), "pair_rows_per_s": round(32768 / ms * 1e3, 1),
                "algorithmic_tflops": round(FLOP_PER_ROW['stereo'] * 32768 / ms / 1e9, 2), "parity": par}
    guarded("stereo_32768", stereo)

    def train():
        from monoloco_amd.train import HipTrainer
        g = np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_train_inputs.npz'))
        res = {"config": "BASELINE configs[4]: LocoModel 34->1024->9 train-mode fwd + MultiTaskLoss + bwd + clip + Adam, "
                         "dropout 0.2; fp32 tensors.  Below 4096 rows the mid route (monoloco_amd/csrc/train_mid.h): every GEMM "
                         "on the exact fp32 MFMA reading the row-major tensors as they lie (launch list: profiles/r06_train_kernel_stats_rows331.txt) -> fraction of "
                         "the 157 TF fp32-MFMA peak.  From 4096 rows the hidden-layer GEMMs run on the 3-product fp16 MFMA "
                         "kernel (fp32-class accuracy), w2 -> w3 as ONE Linear (round 6: 21 batch-sized GEMMs instead of 24; the algorithmic "
                         "FLOP count below stays the reference's 24): fraction of the 833 TF (2500 / 3) a 3-product scheme can reach"}
        sd_tr = {k: torch.tensor(v) for k, v in synth.make_state_dict(31, 34, 9, 1024).items()}
        for tag, rows in (("fixture_331", 331), ("batch_512", 512), ("batch_65536", 65536)):
            tr = HipTrainer(sd_tr, p_dropout=0.2, lr=0.001, device=dev)
            if rows == 331:
                x, y = torch.tensor(g['mono_x']).to(dev), torch.tensor(g['mono_y']).to(dev)
            else:
                rep = (rows + 330) // 331
                x = torch.tensor(g['mono_x']).repeat(rep, 1)[:rows].contiguous().to(dev)
                y = torch.tensor(g['mono_y']).repeat(rep, 1)[:rows].contiguous().to(dev)
                x = x + 0.01 * torch.randn(x.shape, generator=torch.Generator(device='cpu').manual_seed(1000 + rows)).to(dev)   # (seeded: the parity block below is reproducible)
            ms = _ms(lambda: tr.step(x, y), 40 if rows < 4096 else 4, 10 if rows < 4096 else 2, dev)
            # forward 2 FLOP/MAC, backward twice that (dX and dW): 3 x the forward's algorithmic work
            tf = 3 * FLOP_PER_ROW['mono'] * rows / ms / 1e9
            res[tag] = {"rows": rows, "route": tr.last_route, "ms_per_step": round(ms, 4), "rows_per_s": round(rows / ms * 1e3, 1),
                        "algorithmic_tflops": round(tf, 2)}
            if tr.last_route == 'fast':
                res[tag]["frac_of_3product_f16_ceiling"] = round(tf / (PEAK_TFLOPS_F16_DENSE / 3), 4)
            else:
                res[tag]["frac_of_f32_mfma_peak"] = round(tf / PEAK_TFLOPS_F32_MFMA, 4)
            tr.close()
            if not args.no_parity:
                res[tag]["parity"] = train_parity(sd_tr, x, y, dev)
        return res
    guarded("train", train)

    def train_module_loop():
        from monoloco_amd.train import HipTrainer  # noqa: F401  (same library entry points underneath)
        g = np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_train_inputs.npz'))
        sd_tr = {k: torch.tensor(v) for k, v in synth.make_state_dict(31, 34, 9, 1024).items()}
        # the same 512-row iteration as a CALLER-OWNED loop over the module (reference trainer.py:150-161 verbatim: model(inputs), the
        # caller's criterion, loss.backward(), clip_grad_norm_, torch's Adam): LocoModel's train-mode forward on the HIP training kernels
        from monoloco_amd.network.architectures import LocoModel
        import torch.nn.functional as F_

        def multitask_loss(out_, lab_):
            # the CALLER's criterion, plain torch on the device (MultiTaskLoss of the reference, train/losses.py:59-131: Laplace on d,
            # L1 on x, y, h, w, l, ori; all lambdas 1) -- the caller's code, not the library's and not the oracle's
            mu, si, xx = out_[:, 2:3], out_[:, 3:4], lab_[:, 3:4]
            total = torch.mean(torch.abs(1 - mu / xx) * torch.exp(-si) + 0.01 + si + 2)
            for c_ in (0, 1, 4, 5, 6):
                total = total + F_.l1_loss(out_[:, c_:c_ + 1], lab_[:, c_:c_ + 1])
            return total + F_.l1_loss(out_[:, 7:9], lab_[:, 7:9]), None
        rep = (512 + 330) // 331
        xm = torch.tensor(g['mono_x']).repeat(rep, 1)[:512].contiguous().to(dev)
        ym = torch.tensor(g['mono_y']).repeat(rep, 1)[:512].contiguous().to(dev)
        model = LocoModel(34, 9, 1024, p_dropout=0.2)
        model.load_state_dict(sd_tr)
        model = model.to(dev).train()
        opt = torch.optim.Adam(model.parameters(), lr=0.001)

        def iteration():
            opt.zero_grad()
            loss_, _ = multitask_loss(model(xm), ym)
            loss_.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 3)
            opt.step()
        ms = _ms(iteration, 20, 5, dev)
        entry = {"rows": 512, "ms_per_iteration": round(ms, 4), "route": "exact (ml_trainer_forward_train / ml_trainer_backward behind a "
                 "torch.autograd.Function; loss, clip and Adam are torch's own kernels on the module's nn.Parameters)"}
        if not args.no_parity:   # one dropout-0 iteration on fresh weights against the oracle's autograd (fp64, with its fp32 run beside it)
            from oracle.train_oracle import OracleTrainer
            m0 = LocoModel(34, 9, 1024, p_dropout=0.0)
            m0.load_state_dict(sd_tr)
            m0 = m0.to(dev).train()
            l0, _ = multitask_loss(m0(xm), ym)
            l0.backward()
            o64 = OracleTrainer(sd_tr, lr=0.001, dtype=torch.float64)
            r64, _ = o64.step(xm.cpu().double(), ym.cpu().double(), update=False)
            o32 = OracleTrainer(sd_tr, lr=0.001)
            o32.step(xm.cpu(), ym.cpu(), update=False)
            # (the oracle clips to norm 3 inside step(): undo nothing -- compare directions and sizes through the clip factor of each side)
            gn = float(torch.sqrt(sum((p_.grad.double() ** 2).sum() for p_ in m0.parameters())))
            clip = min(1.0, 3.0 / (gn + 1e-6))
            worst, worst32, wk = 0.0, 0.0, None
            g64, g32 = o64.grads(), o32.grads()
            for k, p_ in m0.named_parameters():
                if k.endswith('.bias') and 'batch_norm' not in k and not k.startswith(('w_aux', 'w_fin', 'w2.')):
                    continue
                ref = g64[k]
                sc = float(ref.abs().max()) + 1e-300
                e = float((p_.grad.cpu().double() * clip - ref).abs().max()) / sc
                e32 = float((g32[k].double() - ref).abs().max()) / sc
                if e > worst:
                    worst, worst32, wk = e, e32, k
            entry["parity"] = {"rel_loss_vs_fp64": float('%.3e' % (abs(float(l0.detach()) - r64['loss']) / max(1.0, abs(r64['loss'])))),
                               "worst_gradient_element_over_tensor_max_vs_fp64": float('%.3e' % worst), "fp32_oracle_same_tensor": float('%.3e' % worst32),
                               "worst_tensor": wk, "against": "oracle/train_oracle.OracleTrainer (fp64 / fp32), dropout 0, one iteration; "
                               "tests/test_gpu_autograd.py holds the per-tensor bars"}
        return entry
    guarded("train_module_loop", train_module_loop)

    def train_epoch():
        # the reference's only training figure is the wall time of its fixture run (tests/test_train_mono.py:42-50:
        # `run train --joints sample_joints-kitti-mono.json --lr 0.001 -e 10`, 8.9 s on 8 vCPU): the same
        # loop -- 10 epochs of (one 331-row batch at --bs 512, Adam, per-batch StepLR, clip; validation on the 169-row split;
        # best-epoch weights kept) through monoloco_amd.train.Trainer
        import argparse as ap_
        import tempfile
        from monoloco_amd.train import Trainer
        g = np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_train_inputs.npz'))
        gp = np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_path.npz'))
        joints = {'version': 'bench'}
        for ph, tag in (('train', ''), ('val', 'val'), ('test', 'val')):
            n = len(g['mono_x' + tag])
            joints[ph] = {'X': g['mono_x' + tag].tolist(), 'Y': g['mono_y' + tag].tolist(), 'names': ['x.png'] * n,
                          'kps': gp['mono_kps'][:n, None].tolist(), 'K': [], 'clst': {}}
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, 'joints.json')
            json.dump(joints, open(path, 'w'))
            a = ap_.Namespace(mode='mono', joints=path, epochs=10, no_save=True, lr=0.001, sched_step=30, sched_gamma=0.98,
                              hidden_size=1024, n_stage=3, r_seed=1, out=os.path.join(tmp, 'm.pkl'), bs=512, dropout=0.2,
                              auto_tune_mtl=False)
            t_setup = time.perf_counter()
            tr = Trainer(a)
            t_setup = time.perf_counter() - t_setup
            tr.train()          # warm (first steps allocate the workspace)
            tr2 = Trainer(a)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            tr2.train()
            torch.cuda.synchronize(dev)
            wall = time.perf_counter() - t0
            res = {"config": "Trainer.train(): 10 epochs x (1 training batch of 331 rows at --bs 512 + validation of 169 rows), "
                             "hidden 1024, dropout 0.2, best-epoch weights kept on the device; the reference's fixture run "
                             "(tests/test_train_mono.py:42-50) takes 8.9 s on 8 vCPU incl. start-up",
                   "epochs": 10, "wall_s": round(wall, 5), "ms_per_epoch": round(wall / 10 * 1e3, 4),
                   "device_call_s": round(tr2.step_seconds, 5), "device_call_fraction": round(tr2.step_seconds / wall, 4),
                   "trainer_setup_s": round(t_setup, 4), "route": tr2.hip.last_route,
                   "val_d_first_last": [round(tr2.epoch_losses['val']['d'][0], 4), round(tr2.epoch_losses['val']['d'][-1], 4)],
                   "note": "device_call = time inside ml_trainer_step / ml_trainer_eval (each synchronises the stream); the rest "
                           "is the epoch's row permutation (the draws of the reference's DataLoader, without the loader), one index upload and two "
                           "row-gather launches per batch, and the epoch bookkeeping"}
            tr.hip.close()
            tr2.hip.close()
            if not args.no_parity:
                # the run above draws dropout masks from a device RNG the CPU cannot replay: the SAME ten optimisation steps (one
                # 331-row fixture batch per epoch, Adam, per-batch StepLR, clip 3) with dropout 0 against the oracle's loop
                from monoloco_amd.train import HipTrainer
                from oracle.train_oracle import OracleTrainer
                sd0 = {k: torch.tensor(v) for k, v in synth.make_state_dict(33, 34, 9, 1024).items()}
                xf, yf = torch.tensor(g['mono_x']), torch.tensor(g['mono_y'])
                ht = HipTrainer(sd0, p_dropout=0.0, lr=0.001, sched_step=30, sched_gamma=0.98, device=dev)
                ot = OracleTrainer(sd0, lr=0.001, sched_step=30, sched_gamma=0.98)
                o64 = OracleTrainer(sd0, lr=0.001, sched_step=30, sched_gamma=0.98, dtype=torch.float64)
                names_ = ('loss', 'd', 'x', 'y', 'h', 'w', 'l', 'ori')
                ours, ref32 = [], []
                for _ in range(10):
                    a_, b_, c_ = ht.step(xf, yf), ot.step(xf, yf)[0], o64.step(xf.double(), yf.double())[0]
                    ours.append(max(abs(a_[n] - c_[n]) / max(1.0, abs(c_[n])) for n in names_))
                    ref32.append(max(abs(b_[n] - c_[n]) / max(1.0, abs(c_[n])) for n in names_))
                ht.close()
                # step 1 is a pure function of the inputs (bar 2e-5).  From step 2 on every parameter has moved by +-lr * sign(g) (Adam's
                # first update): gradients within rounding of zero take either sign in any two implementations, and on seeded synthetic
                # weights the loss falls 40x in three steps -- the trajectory itself is ill-conditioned, which the fp32 oracle's own distance
                # from the fp64 run on the same steps shows (tests/test_gpu_train_mid.py::test_mid_route_trajectory_tracks_exact_route)
                res["parity"] = {"steps": 10, "first_step_max_rel_loss_values_vs_fp64": float('%.3e' % ours[0]), "first_step_bar": 2e-5,
                                 "per_step_max_rel_vs_fp64": [float('%.2e' % v) for v in ours],
                                 "fp32_oracle_per_step_max_rel_vs_fp64": [float('%.2e' % v) for v in ref32],
                                 "ok": bool(ours[0] <= 2e-5 and all(o <= max(4.0 * r, 2e-3) for o, r in zip(ours[1:3], ref32[1:3]))),
                                 "bars": "judged on steps 1-3 -- step 1: 2e-5; steps 2, 3: max(4 x the fp32 oracle's own deviation from the fp64 run at "
                                         "that step, 2e-3).  Later steps are reported, not judged: the two CPU runs of the SAME oracle (fp32 vs fp64) "
                                         "drift apart at the percent level by step 4 -- the trajectory amplifies rounding, whoever computes it",
                                 "against": "oracle/train_oracle.OracleTrainer (trainer.py:150-161 under torch CPU autograd) in fp64 and fp32, the "
                                            "same ten steps (one 331-row fixture batch, Adam, per-batch StepLR, clip 3) at dropout 0"}
            return res
    guarded("train_epoch", train_epoch)

    def config4():
        # BASELINE configs[3] on ONE GPU: 1,048,576 persons in one step (the 8-GPU run of the driver cuts the same rows into 8
        # shards: `--total-rows 1048576`, "scaling": "strong")
        rows = 1048576
        k4 = torch.tensor(synth.make_keypoints(rows, seed=400)).to(dev)
        c4 = torch.rand(rows, device=dev)
        o4 = torch.empty((rows, 16), dtype=torch.float32, device=dev)
        x4 = torch.empty((rows, 5), dtype=torch.float32, device=dev)
        eng.reserve(rows)
        r4 = torch.empty((rows, eng.out_features), dtype=torch.float32, device=dev)
        ms = _ms(lambda: eng.forward_mono(k4, kinv, box_conf=c4, out=o4, xyzds=x4, raw=r4), 4, 1, dev)
        res = {"config": "BASELINE configs[3] on one GPU: 1,048,576 persons per step", "ms_per_step": round(ms, 4),
               "persons_per_s": round(rows / ms * 1e3, 1)}
        if not args.no_parity:   # every 61st row (prime: the sample walks through every position of a 256-row tile) = 17190 rows
            res["parity"] = parity_of_timed_run(sd, k4, c4, x4, r4, kk, o4, every=61)
        return res
    guarded("config4_1M_rows_one_gpu", config4)

    def other_batches():
        # row counts between one image and the headline batch (a video batch, a handful of camera streams): the same pipeline, the
        # dense layers on the kernel the engine picks for that row count (small-row <= 512 < dense_mid_kernel <= 8192 < 256x256 tiles)
        res = {"config": "the same mono pipeline at other batch sizes, one GPU; route = the dense kernel family the engine chooses"}
        kept = {}
        for rows in (1024, 2048, 4096, 6144, 8192, 16384):
            k = kps[:rows].contiguous()
            c = conf[:rows].contiguous()
            o = torch.empty((rows, 16), dtype=torch.float32, device=dev)
            x = torch.empty((rows, 5), dtype=torch.float32, device=dev)
            r = torch.empty((rows, eng.out_features), dtype=torch.float32, device=dev)
            ms = _ms(lambda: eng.forward_mono(k, kinv, box_conf=c, out=o, xyzds=x, raw=r), 60, 10, dev)
            res["rows_%d" % rows] = {"us_per_step": round(ms * 1e3, 1), "persons_per_s": round(rows / ms * 1e3, 1),
                                     "route": eng.route_for_rows(rows)}
            kept[rows] = (k, c, o, x, r)
        if not args.no_parity:   # every row of the timed calls' outputs (behind all the timings: the oracle idles the GPU for seconds)
            for rows, (k, c, o, x, r) in kept.items():
                par = parity_of_timed_run(sd, k, c, x, r, kk, o)
                res["rows_%d" % rows]["parity"] = {kk_: par[kk_] for kk_ in ("max_abs_xyzds", "max_abs_raw", "rows_checked", "tolerance",
                                                                            "max_abs_z_spherical")}
        return res
    guarded("other_batches", other_batches)

    def latency():
        k16 = kps[:16].contiguous()
        c16 = conf[:16].contiguous()
        o16 = torch.empty((16, 16), dtype=torch.float32, device=dev)
        x16 = torch.empty((16, 5), dtype=torch.float32, device=dev)
        back_to_back = _ms(lambda: eng.forward_mono(k16, kinv, box_conf=c16, out=o16, xyzds=x16), 300, 300, dev)

        def sync_call():
            eng.forward_mono(k16, kinv, box_conf=c16, out=o16, xyzds=x16)
            torch.cuda.synchronize(dev)
        synced = _ms(sync_call, 200, 20, dev)
        res = {"config": "one image: 16 persons, prep -> MLP -> heads -> post on the device",
               "us_per_call_back_to_back": round(back_to_back * 1e3, 2), "us_per_call_synchronous": round(synced * 1e3, 2)}
        # the drop-in surface, host side included: Loco.forward (lists in, CPU tensors out) + post_process
        import copy
        from monoloco_amd.network import Loco, load_calibration, preprocess_pifpaf
        from monoloco_amd.network.architectures import LocoModel
        model = LocoModel(34, 9, 1024)
        model.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
        net = Loco(model=model, mode='mono', device=dev)
        ann = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'pifpaf_002282.json')))
        boxes, kpl = preprocess_pifpaf(copy.deepcopy(ann), im_size=(1238, 374), enlarge_boxes=False)
        kk1 = load_calibration('kitti', (1238, 374))
        for _ in range(30):
            dic = net.forward(kpl, kk1)
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            dic = net.forward(kpl, kk1)
        t1 = time.perf_counter()
        for _ in range(n):
            net.post_process(dic, boxes, kpl, kk1)
        t2 = time.perf_counter()
        res["loco_forward_us"] = round((t1 - t0) / n * 1e6, 1)
        res["post_process_us"] = round((t2 - t1) / n * 1e6, 1)
        # the same frames with the host sleeping in hipStreamSynchronize instead of polling the frame's completion word (round 5)
        from monoloco_amd import _lib as _l
        _l.load().ml_debug_frame_spin(0)
        for _ in range(30):
            net.forward(kpl, kk1)
        ts = time.perf_counter()
        for _ in range(n):
            net.forward(kpl, kk1)
        res["loco_forward_us_stream_sync"] = round((time.perf_counter() - ts) / n * 1e6, 1)
        res["frame_flag_timeouts"] = int(_l.load().ml_debug_frame_spin(1))
        # MonStereo's per-pair forward (16 left x 5 right persons = 80 pair rows), host side included (ml_loco_frame_stereo)
        sd_s = synth.make_state_dict(3, 68, 10, 1024)
        model_s = LocoModel(68, 10, 1024)
        model_s.load_state_dict({k: torch.tensor(v) for k, v in sd_s.items()})
        net_s = Loco(model=model_s, mode='stereo', device=dev)
        kpr = [[[u - 12.0 for u in k[0]], k[1], k[2]] for k in kpl[:5]]
        for _ in range(30):
            net_s.forward(kpl, kk1, keypoints_r=kpr)
        ts = time.perf_counter()
        for _ in range(n):
            dic_s = net_s.forward(kpl, kk1, keypoints_r=kpr)
        res["stereo_loco_forward_us"] = round((time.perf_counter() - ts) / n * 1e6, 1)
        net_s.engine.close()
        # with ground truth, as GenerateKitti calls it on every image (reference eval/generate_kitti.py:114-132): IoU of all
        # detection x ground-truth pairs, greedy matching, left-to-right order and the matched xyz_real in batched calls
        dic_gt = {'boxes': [[b[0] + 3., b[1] - 2., b[2] + 1., b[3] + 4.] for b in boxes[::-1]],
                  'ys': [[0., 0., 0., 5.0 + i] for i in range(len(boxes))]}
        for _ in range(20):
            out = net.post_process(dic, boxes, kpl, kk1, dic_gt=dic_gt)
        t3 = time.perf_counter()
        for _ in range(n):
            net.post_process(dic, boxes, kpl, kk1, dic_gt=dic_gt)
        res["post_process_with_gt_us"] = round((time.perf_counter() - t3) / n * 1e6, 1)
        res["post_process_with_gt_matches"] = int(sum(out['gt']))
        from monoloco_amd.utils import iou as _iou0
        out_matches = _iou0.get_iou_matches_ordered(boxes, dic_gt['boxes'])
        # an image crowd of 2048 detections against 2048 ground-truth boxes: matching alone (the IoUs on the device from 32768
        # pairs on); the reference's scalar Python loop needs seconds here
        import synth as _synth
        from monoloco_amd.utils import iou as _iou
        for mm in (256, 2048):
            bx, gx = _synth.make_boxes(mm, mm, 5)
            _iou.get_iou_matches_ordered(bx, gx)
            t4 = time.perf_counter()
            for _ in range(5):
                found = _iou.get_iou_matches_ordered(bx, gx)
            res["gt_matching_%d_boxes_ms" % mm] = round((time.perf_counter() - t4) / 5 * 1e3, 3)
            res["gt_matching_%d_boxes_matches" % mm] = len(found)
            if mm == 256:
                found_256, boxes_256 = found, (bx, gx)
        if not args.no_parity:
            # every figure above against the oracle on the very frames that were timed
            from oracle import monoloco_oracle as O
            ref16 = O.forward_mono(sd_t, k16.cpu(), kk, box_conf=c16.cpu())
            ref_f = O.forward_mono(sd_t, torch.tensor(kpl), kk1, box_conf=torch.tensor([b[4] for b in boxes]))
            pp = net.post_process(dic, boxes, kpl, kk1)
            ref_s = O.forward_stereo({k: torch.tensor(v) for k, v in sd_s.items()}, torch.tensor(kpl), torch.tensor(kpr), kk1)
            m_ref = O.reorder_matches(O.get_iou_matches(boxes_256[0], boxes_256[1]), boxes_256[0])
            m_gt = O.associate(boxes, dic_gt['boxes'])[0]
            res["parity"] = {
                "device_pipeline_16_rows_max_abs_xyzds": float('%.3e' % float((x16.cpu() - ref16['xyzds']).abs().max())),
                "loco_forward_max_abs_d_bi": float('%.3e' % max(float((dic['d'] - ref_f['d']).abs().max()), float((dic['bi'] - ref_f['bi']).abs().max()))),
                "post_process_max_abs_xyz_pred": float('%.3e' % float((torch.tensor(pp['xyz_pred']) - ref_f['xyz_pred']).abs().max())),
                "stereo_loco_forward_max_abs_d_bi": (float('%.3e' % max(float((dic_s['d'] - ref_s['d']).abs().max()), float((dic_s['bi'] - ref_s['bi']).abs().max())))
                                                     if ref_s['d'].shape == dic_s['d'].shape else "tie: row counts differ"),
                "post_process_with_gt_same_matches": bool([tuple(int(v) for v in mt) for mt in out_matches] == [tuple(mt) for mt in m_gt]),
                "gt_matching_256_boxes_same_matches": bool([tuple(int(v) for v in mt) for mt in found_256] == [tuple(mt) for mt in m_ref]),
                "tolerance": 1e-4,
                "against": "oracle/monoloco_oracle.{forward_mono, forward_stereo, associate, get_iou_matches + reorder_matches} on the timed frames; "
                           "the 2048-box matching is checked at 16 / 256 / 2048 boxes against the real reference in tests/test_gpu_matching.py "
                           "(the scalar Python loop needs seconds there)"}
        return res
    guarded("latency_16_persons", latency)

    def reference_weights():
        # the same workload on the checkpoint the REFERENCE's own fixture training produced (width 1024) and SURVEY 8d's parity set P
        # (the fixture's real poses tiled to the batch, jittered): its own timing and its own parity block against the CPU oracle
        from oracle import monoloco_oracle as O
        gdir = os.path.join(ROOT, 'tests', 'golden')
        sd_r = dict(np.load(os.path.join(gdir, 'ckpt_mono_h1024.npz')))
        base = np.load(os.path.join(gdir, 'golden_path.npz'))['mono_kps']
        rng = np.random.default_rng(100)
        k = base[rng.integers(0, len(base), m)].astype(np.float32)
        k[:, 0:2, :] += rng.normal(0, 0.5, (m, 2, 17)).astype(np.float32)
        kr = torch.tensor(k).to(dev)
        eng_r = engine.LocoEngine({kk_: torch.tensor(v) for kk_, v in sd_r.items()}, device=dev, reserve_rows=m)
        o = torch.empty((m, 16), dtype=torch.float32, device=dev)
        x = torch.empty((m, 5), dtype=torch.float32, device=dev)
        r = torch.empty((m, eng_r.out_features), dtype=torch.float32, device=dev)
        ms = _ms(lambda: eng_r.forward_mono(kr, kinv, box_conf=conf, out=o, xyzds=x, raw=r), 20, 5, dev)
        par = parity_of_timed_run(sd_r, kr, conf, x, r, kk, o, every=8)
        d_col = r[:, 2]
        eng_r.close()
        return {"config": "the headline workload on the reference-TRAINED 1024-wide checkpoint (tests/golden/ckpt_mono_h1024.npz: the "
                          "reference's own fixture training, oracle/make_golden.py wb1024) and real fixture poses tiled to the batch "
                          "(+ N(0, 0.5 px)); same kernels and work as `value`",
                "ms_per_step": round(ms, 4), "persons_per_s": round(m / ms * 1e3, 1), "vs_synthetic_values": round(main_ms / ms, 4),
                "parity": par, "d_metres_min_max": [round(float(d_col.min()), 2), round(float(d_col.max()), 2)]}
    guarded("reference_trained_weights", reference_weights)

    def bf16():
        from oracle import monoloco_oracle as O
        eng_b = engine.LocoEngine({k: torch.tensor(v) for k, v in sd.items()}, device=dev, precision='bf16', reserve_rows=m)
        o = torch.empty((m, 16), dtype=torch.float32, device=dev)
        x = torch.empty((m, 5), dtype=torch.float32, device=dev)
        r = torch.empty((m, eng_b.out_features), dtype=torch.float32, device=dev)
        ms = _ms(lambda: eng_b.forward_mono(kps, kinv, box_conf=conf, out=o, xyzds=x, raw=r), 10, 3, dev)
        idx = torch.arange(0, m, max(1, m // 768))[:768].to(dev)
        ref = O.forward_mono({k: torch.tensor(v) for k, v in sd.items()}, kps[idx].cpu(), kk, box_conf=conf[idx].cpu())
        dev_abs = (x[idx].cpu() - ref['xyzds']).abs().max(0).values.tolist()
        eng_b.close()
        return {"config": "the same workload with plain bf16 operands (v_mfma_f32_32x32x16_bf16, one product per term, fp32 "
                          "accumulate) -- BASELINE configs[1] says 'bf16'; comparison only, misses the 1e-4 bar",
                "ms_per_step": round(ms, 4), "persons_per_s": round(m / ms * 1e3, 1),
                "speedup_vs_f16x2": round(main_ms / ms, 3),
                "max_abs_dev_x_y_z_d_sigma": [float('%.3e' % v) for v in dev_abs],
                "meets_1e-4": bool(max(dev_abs) <= 1e-4)}
    guarded("bf16_mode", bf16)
    return out


def ensure_library(path, local_rank, build, timeout_s=900.0, settle_s=2.0):
    """The library normally travels with the tree; on a box without it LOCAL rank 0 builds it and the other local ranks
    wait for the file (then `settle_s` for the linker to finish writing).  Raises if it never appears."""
    if os.path.exists(path):
        return False
    if local_rank == 0:
        build()
    else:
        t_wait = time.time()
        while not os.path.exists(path) and time.time() - t_wait < timeout_s:
            time.sleep(0.2)
        time.sleep(settle_s)
    if not os.path.exists(path):
        raise SystemExit("bench.py: %s was not built (local rank %d)" % (path, local_rank))
    return True


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n, argv):
    """`python bench.py --gpus N` without torchrun's environment: re-execute this file as N ranks under
    `python -m torch.distributed.run` (one process per GPU, rendezvous on 127.0.0.1, a free port), pass the ranks'
    stdout / stderr through and return the launcher's exit code.  Rank 0 prints the JSON line."""
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL needs it on this pool's host driver
    env.setdefault('OMP_NUM_THREADS', '1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=65536, help='persons per GPU per step (weak scaling)')
    ap.add_argument('--total-rows', type=int, default=0,
                    help='strong scaling: this many persons in total, cut into per-rank shards (BASELINE configs[3]: 1048576)')
    ap.add_argument('--precision', default='f16x2', choices=['f16x2', 'f16', 'bf16'])
    ap.add_argument('--no-merge', action='store_true', help='keep w2 and w3 as two dense layers')
    ap.add_argument('--workload', default='mono', choices=['mono', 'stereo'])
    ap.add_argument('--cpu-seconds', type=float, default=20.0, help='budget of the cpu_baseline leg (0 = skip)')
    ap.add_argument('--no-profile', action='store_true', help='do not bracket dense launches with HIP events')
    ap.add_argument('--no-extra', action='store_true', help='skip the `extra` legs (other configs, bf16 mode, e2e)')
    ap.add_argument('--no-live-counters', action='store_true',
                    help='do not re-run the headline command under rocprofv3 for roofline.traffic / mfma_busy (N = 1 only; the child runs pass this)')
    ap.add_argument('--no-parity', action='store_true', help='skip the oracle check of the timed outputs')
    ap.add_argument('--tile-kernel', type=int, default=0, choices=[0, 2, 4, 260],
                    help='A/B: 2 = dense_kernel_pp everywhere, 260 = dense_kernel_w4 everywhere, 0 / 4 = library default (w4 '
                         'for the long-K layers, pp for the input and fused-head layers)')
    ap.add_argument('--chunk-rows', type=int, default=-1,
                    help='walk the batch in row chunks of this size through all layers (-1 = library default)')
    ap.add_argument('--gather', default='gather', choices=['gather', 'all_gather'], help='N > 1: the final collective')
    ap.add_argument('--weights', default='synthetic', choices=['reference', 'synthetic'],
                    help="'reference': the 1024-wide checkpoint the reference's own fixture training produced "
                         "(tests/golden/ckpt_*_h1024.npz, oracle/make_golden.py wb1024) on SURVEY 8d's parity set P (the fixture's real "
                         "poses tiled to the batch, jittered by N(0, 0.5 px)); 'synthetic' (default, as in every round): seeded weights on uniform "
                         "random keypoints (set T).  Same kernels, same work; the board's power -- hence the clock at its cap -- depends on the "
                         "values by about 1 %% (measured: 2.647 vs 2.622 ms, tools/ab_weights.sh); `extra.reference_trained_weights` times and "
                         "checks the other set in every default run")
    ap.add_argument('--force-distributed', action='store_true',
                    help='take the N > 1 code path (process group on the real backend, sharded rows, the gather, gather_check, '
                         'config4_strong) at ANY world size, also 1: the pre-flight of the multi-GPU run on a one-GPU box')
    ap.add_argument('--stub-engine', action='store_true',
                    help='launch-contract test on a box without a GPU: tests/stub_engine.py on CPU over gloo; the line is '
                         'marked "data": "stub" and measures nothing')
    args = ap.parse_args(argv)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (one process per GPU) and relay their line
        if not args.stub_engine:
            import torch
            have = torch.cuda.device_count()
            if have < args.gpus:      # fail before N processes do, with one line
                raise SystemExit("bench.py: --gpus %d but this node shows %d HIP device(s) (torch.cuda.device_count()); nothing launched"
                                 % (args.gpus, have))
        return self_launch(args.gpus, list(sys.argv[1:] if argv is None else argv))

    import numpy as np
    import torch
    import torch.distributed as dist
    import synth
    from monoloco_amd import _lib
    stub = args.stub_engine
    if not stub:
        import __graft_entry__
        ensure_library(_lib.LIB_PATH, int(os.environ.get('LOCAL_RANK', '0')), __graft_entry__.build)
    from monoloco_amd import engine, parallel

    multi = int(os.environ.get('WORLD_SIZE', '1')) > 1 or args.force_distributed
    if not stub and multi:
        # one-line, non-zero failure BEFORE the rendezvous when this rank has no device of its own (torchrun with more ranks than GPUs)
        have, want_local = torch.cuda.device_count(), int(os.environ.get('LOCAL_RANK', '0'))
        if have <= want_local or have < min(args.gpus, int(os.environ.get('LOCAL_WORLD_SIZE', args.gpus))):
            raise SystemExit("bench.py: rank %s (local rank %d) has no HIP device of its own: --gpus %d, torch.cuda.device_count() = %d"
                             % (os.environ.get('RANK', '0'), want_local, args.gpus, have))
    rank, world, local = parallel.init_from_env(('gloo' if stub else 'nccl') if multi else None, force=args.force_distributed)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (the launcher's --nproc-per-node and --gpus must agree)"
                         % (args.gpus, world))
    if stub:
        import stub_engine
        dev = torch.device('cpu')
        make_engine = stub_engine.StubEngine
    else:
        dev = torch.device('cuda', parallel.local_device_index(local))
        torch.cuda.set_device(dev)
        make_engine = engine.LocoEngine

    strong = args.total_rows > 0
    if strong:
        lo, hi = parallel.shard_bounds(args.total_rows, world, rank)
        m = hi - lo
    else:
        m = args.batch
    kk = synth.KITTI_K
    kinv = engine.inverse_intrinsics(kk)
    gdir = os.path.join(ROOT, 'tests', 'golden')
    ckpt = os.path.join(gdir, 'ckpt_%s_h1024.npz' % args.workload)
    use_ref = args.weights == 'reference' and os.path.exists(ckpt) and os.path.exists(os.path.join(gdir, 'golden_path.npz'))
    weights_txt = ("reference-trained: tests/golden/ckpt_%s_h1024.npz (the reference's own fixture training at width 1024, "
                   "oracle/make_golden.py wb1024); poses = the fixture's real poses tiled to the batch, jittered by N(0, 0.5 px)"
                   % args.workload) if use_ref else "seeded synthetic (tests/synth.py) on uniform random keypoints"

    def poses(n, seed, key='mono_kps'):
        if not use_ref:
            return synth.make_keypoints(n, seed=seed)
        base = np.load(os.path.join(gdir, 'golden_path.npz'))[key]
        rng = np.random.default_rng(seed)
        k = base[rng.integers(0, len(base), n)].astype(np.float32)
        k[:, 0:2, :] += rng.normal(0, 0.5, (n, 2, 17)).astype(np.float32)
        return k
    if args.workload == 'mono':
        sd = dict(np.load(ckpt)) if use_ref else synth.make_state_dict(1, 34, 9, 1024)
        kps_np = poses(m, 100 + rank)
        rows = m
    else:  # MonStereo: ml x mr all-vs-all pairs = `batch` network rows
        sd = dict(np.load(ckpt)) if use_ref else synth.make_state_dict(3, 68, 10, 1024)
        mr = 128
        ml = m // mr
        kps_np = poses(ml, 100 + rank, 'stereo_kps_l')
        kps_r = torch.tensor(poses(mr, 200 + rank, 'stereo_kps_r')).to(dev)
        rows = ml * mr
    eng = make_engine({k: torch.tensor(v) for k, v in sd.items()}, device=dev, precision=args.precision,
                      merge_w2w3=not args.no_merge, reserve_rows=rows)
    if args.tile_kernel:
        eng.set_tuning(tile_kernel=args.tile_kernel & 255, everywhere=bool(args.tile_kernel & 256))
    if args.chunk_rows >= 0:
        eng.set_tuning(chunk_rows=args.chunk_rows)
    kps = torch.tensor(kps_np).to(dev)
    n_out = kps.shape[0]
    conf = torch.rand(n_out, device=dev)
    out = torch.empty((n_out, 16), dtype=torch.float32, device=dev)
    xyzds = torch.empty((n_out, 5), dtype=torch.float32, device=dev)
    raw = torch.empty((n_out, eng.out_features), dtype=torch.float32, device=dev)
    total_out = args.total_rows if strong else n_out * world
    sharded = parallel.ShardedRows(total_out, 5, dev, mode=args.gather) if multi else None

    def local_block(lo=0, hi=0):
        if args.workload == 'mono':
            eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds, raw=raw)
            return xyzds
        return eng.forward_stereo(kps, kps_r, kinv, box_conf=conf)['xyzds']

    def step():
        if sharded is not None:
            sharded.run(local_block)   # local compute, then the single gather to rank 0
        else:
            local_block()

    if not args.no_profile:
        hooks = (lambda: eng.profile_begin(args.steps * eng.num_layers * 16 + 8), eng.profile_end)
    else:
        hooks = (None, None)
    mine = {}
    dt, prof = timed_steps(step, args.steps, args.warmup, world, dev, begin=hooks[0], end=hooks[1], local_out=mine)

    # who took part: one record per rank (its device's identity and its own clock around the same K steps)
    me = {"rank": rank, "local_rank": local, "device": device_identity(dev), "rows": rows,
          "ms_per_step": round(mine['local_s'] / args.steps * 1e3, 4)}
    if multi:
        seen = [None] * world
        dist.all_gather_object(seen, me)
    else:
        seen = [me]
    if multi and not stub and len({r["device"] for r in seen}) < world:
        # (cannot happen under torchrun's device-per-local-rank convention; kept as the line's own guarantee)
        raise SystemExit("bench.py: %d ranks share %d HIP device(s) (%s): one process per GPU is the contract"
                         % (world, len({r["device"] for r in seen}), sorted({r["device"] for r in seen})))

    gather_check = None
    if sharded is not None:
        # the gathered block really holds every rank's rows, in rank order: per-shard checksums on rank 0 against each
        # rank's own checksum of what it computed -- the int32 bit patterns summed in int64 (exact, independent of the order a
        # reduction walks an offset slice in; a gather moves bits, so NaN rows compare like any other)
        def bits_sum(t):
            return int(t.contiguous().view(torch.int32).to(torch.int64).sum().item())
        full = sharded.run(local_block)
        own = bits_sum(xyzds)
        sums = [None] * world
        dist.all_gather_object(sums, own)
        if rank == 0:
            got = [bits_sum(full[lo_:hi_]) for lo_, hi_ in sharded.gather.bounds]
            gather_check = {"ok": bool(all(a == b for a, b in zip(got, sums))), "shards": len(got),
                            "rows": int(full.shape[0]), "distinct_shards": len(set(got)), "checksum": "int64 sum of the fp32 bit patterns"}

    gather_ms = None
    if sharded is not None:   # the collective alone, same buffers, same barriers (not part of `value`'s clock)
        g_dt, _ = timed_steps(lambda: sharded.gather(xyzds), args.steps, 2, world, dev)
        gather_ms = g_dt / args.steps * 1e3

    # N > 1: the same launch also times BASELINE configs[3] (1,048,576 persons cut into N shards, one gather: "strong"), so the
    # driver's `bench.py --gpus N` needs no further flags to cover it; reported as `config4_strong` beside the weak line
    strong4 = None
    if multi and not strong and args.workload == 'mono':
        total4 = 1048576
        lo4, hi4 = parallel.shard_bounds(total4, world, rank)
        m4 = hi4 - lo4
        kps4 = torch.tensor(synth.make_keypoints(m4, seed=500 + rank)).to(dev)
        conf4 = torch.rand(m4, device=dev)
        out4 = torch.empty((m4, 16), dtype=torch.float32, device=dev)
        xyz4 = torch.empty((m4, 5), dtype=torch.float32, device=dev)
        eng.reserve(m4)
        sh4 = parallel.ShardedRows(total4, 5, dev, mode=args.gather)

        def block4(lo=0, hi=0):
            eng.forward_mono(kps4, kinv, box_conf=conf4, out=out4, xyzds=xyz4)
            return xyz4
        n4 = max(3, args.steps // 4)
        dt4, _ = timed_steps(lambda: sh4.run(block4), n4, 2, world, dev)
        g4, _ = timed_steps(lambda: sh4.gather(xyz4), n4, 2, world, dev)
        strong4 = {"scaling": "strong", "total_rows": total4, "rows_per_gpu": m4, "steps": n4,
                   "ms_per_step": round(dt4 / n4 * 1e3, 4), "value": round(total4 * n4 / dt4, 1), "unit": "persons/s",
                   "gather_ms": round(g4 / n4 * 1e3, 4),
                   "config": "BASELINE configs[3]: MonoLoco++ mono, 1,048,576 persons sharded over %d GPUs, one %s" % (world, args.gather)}

    if rank == 0:
        total_rows = (args.total_rows if strong else rows * world) * args.steps
        value = total_rows / dt
        ms_per_step = dt / args.steps * 1e3
        prec_txt = {'f16x2': "f16x2 (fp16 hi+lo split operands, 3 MFMA/term, fp32 accumulate)", 'f16': "f16 (fp32 accumulate)",
                    'bf16': "bf16 (fp32 accumulate)"}[args.precision]
        line = {
            "metric": METRIC,
            "value": round(value, 1), "unit": "persons/s" if args.workload == 'mono' else "pair-rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": prec_txt,
            "data": "synthetic",
            "config": {"workload": ("MonoLoco++ mono MLP 34->1024->9 + pre/post-process, synthetic 17x2 keypoints, "
                                    + ("%d persons in total over %d GPUs" % (args.total_rows, world) if strong
                                       else "batch %d per GPU" % m)) if args.workload == 'mono' else
                                   ("MonStereo 68->1024->10, %d x %d all-vs-all pairs per GPU" % (ml, mr)),
                       "rows_per_gpu": rows, "precision": args.precision, "merge_w2w3": not args.no_merge,
                       "weights": weights_txt,
                       "parallelism": "rows sharded x%d, 1 %s" % (world, args.gather)},
        }
        if multi:
            # which collective library carried the gather: RCCL's version as torch reports it (the `nccl` backend IS RCCL on ROCm)
            try:
                line["collectives"] = {"backend": dist.get_backend(), "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version())
                                       if not stub else None, "forced_at_world_1": bool(args.force_distributed and world == 1)}
            except Exception as e:   # identity garnish
                line["collectives"] = {"backend": dist.get_backend(), "rccl_version": repr(e)[:80]}
        line["ranks_seen"] = len({(r["rank"], r["device"]) for r in seen})
        line["ranks"] = seen
        per = [r["ms_per_step"] for r in seen]
        line["ms_per_step_min_rank"], line["ms_per_step_max_rank"] = min(per), max(per)
        if stub:
            line["data"] = "stub"
            line["config"]["stub"] = ("tests/stub_engine.py on CPU over gloo: the launch contract only, no device work -- "
                                      "`value` measures nothing")
        if gather_ms is not None:
            line["gather_ms"] = round(gather_ms, 4)
        if gather_check is not None:
            line["gather_check"] = gather_check
        if strong4 is not None:
            line["config4_strong"] = strong4
        if prof and prof['launches']:
            dense_s = prof['total_ms'] * 1e-3
            replay = traffic_from_profiles(args)
            alg_flop = FLOP_PER_ROW[args.workload] * rows * args.steps
            achieved = alg_flop / dense_s / 1e12
            line["roofline"] = {
                "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_TFLOPS_F16_DENSE, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_TFLOPS_F16_DENSE, 4),
                # the same numerator over the WHOLE step's wall time (prep, tail and launch gaps included; what `value` implies)
                "achieved_of_step": round(alg_flop / dt / 1e12, 2),
                "frac_of_step": round(alg_flop / dt / 1e12 / PEAK_TFLOPS_F16_DENSE, 4),
                "traffic": replay[0]["hbm_bytes_per_launch"] if replay else None,
                "traffic_source": replay[1] if replay else "none: PMC cannot be read live; no committed pass covers this configuration",
                "algorithmic_bytes_per_launch": replay[0].get("algorithmic_bytes_per_launch") if replay else None,
                # the two counters the north_star names, replayed from the same committed passes (their own *_source fields)
                "mfma_busy": replay[0].get("mfma_busy") if replay else None,
                "mfma_busy_source": (replay[1] + " (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), launch-weighted)") if replay and replay[0].get("mfma_busy") is not None else None,
                "hbm_gbps": replay[0].get("hbm_gbps") if replay else None,
                "hbm_gbps_source": (replay[1] + " (HBM bytes / kernel duration under rocprofv3; HBM peak ~8000 GB/s)") if replay and replay[0].get("hbm_gbps") is not None else None,
                # (replayed byte count over THIS run's event-timed kernel seconds: half live -- roofline.hbm_gbps_live below is the fully live one)
                "hbm_gbps_replayed_bytes_live_time": round((replay[0]["hbm_bytes_per_launch"] * prof['launches']) / (prof['total_ms'] * 1e-3) / 1e9, 1) if replay else None,
                "kernel": "mlk::dense_kernel_%s<%d,*,*,*>" % ({2: "pp", 260: "w4"}.get(args.tile_kernel, "w4 (6 long-K layers) + dense_kernel_pp (input and fused-head layers)"), {'f16x2': 3, 'f16': 1, 'bf16': 0}[args.precision]),
                "launches": prof['launches'], "avg_launch_ms": round(prof['total_ms'] / prof['launches'], 5),
                "per_layer_avg_ms": [round(a / max(n, 1), 5) for a, n in zip(prof['per_layer_ms'], prof['per_layer_n'])],
                "note": "achieved = algorithmic FLOP of the reference layer structure (%d/row) / summed dense-kernel "
                        "time on rank 0 (HIP events on the launch stream); executed MFMA FLOP are %sx higher"
                        % (FLOP_PER_ROW[args.workload], "~2.6" if args.precision == 'f16x2' else "~0.88"),
            }
        if world == 1 and not multi and "roofline" in line and not args.no_live_counters and args.workload == 'mono' and not stub \
                and not args.total_rows and not args.tile_kernel and args.chunk_rows < 0 \
                and args.batch == 65536 and args.precision == 'f16x2' and not args.no_merge:   # (the launch list tools/traffic_from_pmc.py knows)
            # the counters of THIS run: child processes of the same command under rocprofv3 (the parent only waits meanwhile)
            rl = line["roofline"]
            live = live_counters(args)
            rl["counters_source"] = live.get("source")
            if "hbm_bytes_per_launch" in live:
                rl["traffic_replayed"], rl["mfma_busy_replayed"], rl["hbm_gbps_replayed"] = rl["traffic"], rl["mfma_busy"], rl["hbm_gbps"]
                rl["traffic"] = rl["traffic_live"] = live["hbm_bytes_per_launch"]
                rl["traffic_source"] = "live (roofline.counters_source); the committed passes are kept beside it as *_replayed"
                rl["algorithmic_bytes_per_launch"] = live.get("algorithmic_bytes_per_launch")
                rl["mfma_busy"] = rl["mfma_busy_live"] = live.get("mfma_busy")
                rl["mfma_busy_source"] = "live: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), launch-weighted over the 8 dense launches"
                rl["hbm_gbps"] = rl["hbm_gbps_live"] = live.get("hbm_gbps")
                rl["hbm_gbps_source"] = "live: counter bytes / kernel durations of the --kernel-trace --stats pass (HBM peak ~8000 GB/s)"
                rl["dominant_kernel_live"] = live.get("dominant_kernel")
                rl["dominant_kernel_avg_us_live"] = live.get("dominant_kernel_avg_us")
                rl["per_kernel_live"] = {"read_MB": live.get("per_kernel_read_MB"), "write_MB": live.get("per_kernel_write_MB"),
                                         "mfma_busy": live.get("mfma_busy_per_kernel"), "avg_launch_us": live.get("avg_launch_us_under_rocprof")}
                rl["live_pass_seconds"] = live.get("pass_seconds")
        if world == 1 and "roofline" in line:
            pw = power_probe(step, dev)
            if pw is not None:
                line["roofline"]["power"] = pw
        if args.workload == 'mono' and not stub and not args.no_parity:
            line["parity"] = parity_of_timed_run(sd, kps, conf, xyzds, raw, kk, out)
        if world == 1 and not multi and args.workload == 'mono' and not args.no_extra and not stub:
            line["extra"] = extras(args, dev, sd, eng, kps, conf, kinv, kk, ms_per_step, main_out=(out.clone(), xyzds.clone()))
            if "e2e" in line["extra"] and "e2e_ms" in line["extra"]["e2e"]:
                line["e2e_ms"] = line["extra"]["e2e"]["e2e_ms"]
        if world == 1 and args.cpu_seconds > 0 and args.workload == 'mono' and not stub:
            line["cpu_baseline"] = cpu_baseline(sd, kps_np, kk, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == '__main__':
    sys.exit(main() or 0)
