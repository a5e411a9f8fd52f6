#!/usr/bin/env python3
"""bench.py -- persons/s of the monoloco keypoint->3D hot path on N MI355X.

A "step" is one pass of the whole device pipeline over one batch of synthetic keypoints that is
already resident in HBM:  pre-process (pixel_to_camera) -> MonoLoco++ residual MLP (34->1024->9)
-> post-process (extract_outputs + back-projection), and for N > 1 the single RCCL gather of the
(m,5) (x,y,z,d,sigma) block to rank 0.  Workload = BASELINE.json configs[1]: batch 65536 persons
per GPU ("weak" scaling: per-GPU work fixed, rows sharded, no data-path collective but the
final gather).  Prints ONE JSON line on rank 0.

  python bench.py                                   # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
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


def cpu_baseline(sd, kps_np, kk, budget_s):
    """The CPU oracle (a torch-CPU fp32 restatement of the reference path: preprocess_monoloco ->
    LocoModel eval forward -> extract_outputs -> back-projection) timed on this box's host cores on a
    bounded sample of the same workload."""
    import torch
    from oracle import monoloco_oracle as O
    threads = torch.get_num_threads()
    sd_t = {k: torch.tensor(v) for k, v in sd.items()}
    n = min(len(kps_np), 65536)
    kps = torch.tensor(kps_np[:n])
    O.forward_mono(sd_t, kps[:4096], kk)  # warm-up (thread pool, allocator)
    reps, t_tot = 0, 0.0
    while t_tot < budget_s and reps < 20:
        t0 = time.perf_counter()
        O.forward_mono(sd_t, kps, kk)
        t_tot += time.perf_counter() - t0
        reps += 1
    return {"value": round(n * reps / t_tot, 1), "unit": "persons/s", "cores": threads, "kind": "port",
            "sample": "%d x %d persons of the same synthetic batch, oracle/monoloco_oracle.forward_mono "
                      "(torch CPU fp32, %d threads), %.1f s" % (reps, n, threads, t_tot)}


def traffic_from_profiles(args):
    """HBM bytes per dense launch from the PMC passes committed under profiles/ (FETCH_SIZE x2 + WRITE_SIZE,
    collected with rocprofv3 in separate runs of this very command; PMC cannot be read live).  None if the
    committed measurement does not cover this configuration."""
    path = os.path.join(ROOT, 'profiles', 'r01_traffic.json')
    if args.workload != 'mono' or args.precision != 'f16x2' or args.batch != 65536 or args.no_merge \
            or not os.path.exists(path):
        return None
    return json.load(open(path))['hbm_bytes_per_launch']


def timed_steps(step, steps, warmup, world, device, begin=None, end=None):
    """The timing contract: `warmup` untimed steps, then exactly `steps` steps bracketed by barrier + device
    synchronisation on both sides; returns (seconds = MAX over ranks, whatever `end()` returned on this rank).
    `device` is the rank's HIP device (a CPU device in the gloo tests of this very function)."""
    import torch
    import torch.distributed as dist
    on_gpu = device.type == 'cuda'

    def fence():
        if on_gpu:
            torch.cuda.synchronize(device)
        if world > 1:
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
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=65536, help='persons per GPU per step')
    ap.add_argument('--precision', default='f16x2', choices=['f16x2', 'f16'])
    ap.add_argument('--no-merge', action='store_true', help='keep w2 and w3 as two dense layers')
    ap.add_argument('--workload', default='mono', choices=['mono', 'stereo'])
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='budget of the cpu_baseline leg (0 = skip)')
    ap.add_argument('--no-profile', action='store_true', help='do not bracket dense launches with HIP events')
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import synth
    from monoloco_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        # the library normally travels with the tree; on a box without it local rank 0 builds it, the others wait
        if int(os.environ.get('LOCAL_RANK', '0')) == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            t_wait = time.time()
            while not os.path.exists(_lib.LIB_PATH) and time.time() - t_wait < 900:
                time.sleep(1.0)
            time.sleep(2.0)  # let the linker finish writing
    from monoloco_amd import engine, parallel

    rank, world, local = parallel.init_from_env('nccl' if int(os.environ.get('WORLD_SIZE', '1')) > 1 else None)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    m = args.batch
    kk = synth.KITTI_K
    kinv = engine.inverse_intrinsics(kk)
    if args.workload == 'mono':
        sd = synth.make_state_dict(1, 34, 9, 1024)
        kps_np = synth.make_keypoints(m, seed=100 + rank)
        rows = m
    else:  # MonStereo: ml x mr all-vs-all pairs = `batch` network rows
        sd = synth.make_state_dict(3, 68, 10, 1024)
        mr = 128
        ml = m // mr
        kps_np = synth.make_keypoints(ml, seed=100 + rank)
        kps_r = torch.tensor(synth.make_keypoints(mr, seed=200 + rank)).to(dev)
        rows = ml * mr
    eng = engine.LocoEngine({k: torch.tensor(v) for k, v in sd.items()}, device=dev, precision=args.precision,
                            merge_w2w3=not args.no_merge, reserve_rows=rows)
    kps = torch.tensor(kps_np).to(dev)
    n_out = kps.shape[0]
    conf = torch.rand(n_out, device=dev)
    out = torch.empty((n_out, 16), dtype=torch.float32, device=dev)
    xyzds = torch.empty((n_out, 5), dtype=torch.float32, device=dev)
    sharded = parallel.ShardedRows(n_out * world, 5, dev) if world > 1 else None  # this rank's shard = its own batch

    def local_block(lo=0, hi=0):
        if args.workload == 'mono':
            eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
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
    dt, prof = timed_steps(step, args.steps, args.warmup, world, dev, begin=hooks[0], end=hooks[1])

    if rank == 0:
        total_rows = rows * world * args.steps
        value = total_rows / dt
        line = {
            "metric": "persons/sec (17-kp MLP forward+postproc) at batch 65536, 1/2/4/8 MI355X",
            "value": round(value, 1), "unit": "persons/s" if args.workload == 'mono' else "pair-rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16x2 (fp16 hi+lo split operands, 3 MFMA/term, fp32 accumulate)" if args.precision == 'f16x2'
                     else "f16 (fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": ("MonoLoco++ mono MLP 34->1024->9 + pre/post-process, synthetic 17x2 keypoints, "
                                    "batch %d per GPU" % m) if args.workload == 'mono' else
                                   ("MonStereo 68->1024->10, %d x %d all-vs-all pairs per GPU" % (ml, mr)),
                       "rows_per_gpu": rows, "precision": args.precision, "merge_w2w3": not args.no_merge,
                       "weights": "seeded synthetic (tests/synth.py)", "parallelism": "rows sharded x%d, 1 gather" % world},
        }
        if prof and prof['launches']:
            dense_s = prof['total_ms'] * 1e-3
            alg_flop = FLOP_PER_ROW[args.workload] * rows * args.steps
            achieved = alg_flop / dense_s / 1e12
            line["roofline"] = {
                "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_TFLOPS_F16_DENSE, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_TFLOPS_F16_DENSE, 4), "traffic": traffic_from_profiles(args),
                "kernel": "mlk::dense_kernel_pp<%d,*,*,*>" % (3 if args.precision == 'f16x2' else 1),
                "launches": prof['launches'], "avg_launch_ms": round(prof['total_ms'] / prof['launches'], 5),
                "per_layer_avg_ms": [round(a / max(n, 1), 5) for a, n in zip(prof['per_layer_ms'], prof['per_layer_n'])],
                "note": "achieved = algorithmic FLOP of the reference layer structure (%d/row) / summed dense-kernel "
                        "time on rank 0 (HIP events on the launch stream); executed MFMA FLOP are %sx higher"
                        % (FLOP_PER_ROW[args.workload], "~2.6" if args.precision == 'f16x2' else "~0.88"),
            }
        if world == 1 and args.cpu_seconds > 0 and args.workload == 'mono':
            line["cpu_baseline"] = cpu_baseline(sd, kps_np, kk, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == '__main__':
    main()
