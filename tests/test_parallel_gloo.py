"""The N > 1 layout on CPU: contiguous row shards, one gather to rank 0 (gloo, world_size 2 and 3)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from monoloco_amd import parallel


def test_shard_bounds_cover_and_balance():
    for total in (0, 1, 7, 65536, 1048576, 1000003):
        for world in (1, 2, 3, 8):
            b = [parallel.shard_bounds(total, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_from_env('gloo')
    assert (r, w) == (rank, world)
    lo, hi = parallel.shard_bounds(total, world, rank)
    # stand-in for the per-rank device pipeline: a deterministic function of the global row id
    rows = torch.arange(lo, hi, dtype=torch.float32).unsqueeze(1) * torch.tensor([[1., 2., 3., 4., 5.]])
    sharded = parallel.ShardedRows(total, 5, torch.device('cpu'))
    assert (sharded.lo, sharded.hi) == (lo, hi)
    for _ in range(2):  # the pre-allocated gather is reusable step after step (this is bench.py's N > 1 step)
        full = sharded.run(lambda a, b: rows)
    if rank == 0:
        ref = torch.arange(total, dtype=torch.float32).unsqueeze(1) * torch.tensor([[1., 2., 3., 4., 5.]])
        q.put(bool(torch.equal(full, ref)))
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 4096), (2, 4097), (3, 1000)])
def test_row_gather_gloo(world, total):
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() is True


def _bench_worker(rank, world, port, q):
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_from_env('gloo')
    calls = []
    sharded = parallel.ShardedRows(8 * world, 5, torch.device('cpu'))

    def step():  # bench.py's N > 1 step with a stand-in for the device pipeline; rank r is (r + 1) x slower
        calls.append(1)
        time.sleep(0.01 * (rank + 1))
        sharded.run(lambda lo, hi: torch.full((hi - lo, 5), float(rank)))

    marks = []
    dt, extra = bench.timed_steps(step, 5, 2, world, torch.device('cpu'), begin=lambda: marks.append(len(calls)),
                                  end=lambda: 'done')
    q.put((rank, dt, len(calls), marks, extra))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_timing_contract_gloo():
    """bench.timed_steps: W untimed warm-up steps, exactly K timed ones between barriers, MAX over ranks."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get() for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (r0, dt0, n0, marks0, e0), (r1, dt1, n1, marks1, e1) = res
    assert n0 == n1 == 7 and marks0 == marks1 == [2] and e0 == e1 == 'done'   # 2 warm-up + 5 timed, begin() after warm-up
    assert dt0 == dt1                                                           # every rank reports the MAX
    assert 5 * 0.02 <= dt0 < 5 * 0.02 + 0.5                                     # the slow rank's 5 x 20 ms


def _stereo_worker(rank, world, port, q):
    """Stereo shards by LEFT person: each rank runs the all-vs-all pairing of ITS left persons against
    ALL right persons, the per-left arg-max over the aux logit stays local, one gather of the (ml, 5) block."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    import synth
    from oracle import monoloco_oracle as O
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_from_env('gloo')
    torch.set_num_threads(1)
    import numpy as np
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    ml = 23                                                           # ragged over 2 and 3 ranks
    # the reference-trained fixture net and fixture poses: the aux head really discriminates the right candidates
    sd = {k: torch.tensor(v) for k, v in np.load(os.path.join(gdir, 'ckpt_stereo_h256.npz')).items()}
    g = np.load(os.path.join(gdir, 'golden_path.npz'))
    kl = torch.tensor(g['stereo_kps_l'][:ml])
    kr = torch.tensor(g['stereo_kps_r'][[2, 5, 8, 11, 14, 20]])    # six distinct right poses (no exact ties)
    sharded = parallel.ShardedRows(ml, 5, torch.device('cpu'))

    def local(lo, hi):  # stand-in for LocoEngine.forward_stereo on this rank's left persons
        return O.forward_stereo(sd, kl[lo:hi], kr, synth.KITTI_K)['xyzds'] if hi > lo else torch.empty((0, 5))

    full = sharded.run(local)
    if rank == 0:
        ref = O.forward_stereo(sd, kl, kr, synth.KITTI_K)
        # non-trivial selection: the winners are not all the same right person
        best = ref['mask'].float().argmax(1)
        q.put((bool(torch.equal(full, ref['xyzds'])), int(best.unique().numel())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_stereo_shard_by_left_gloo(world):
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_stereo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    same, distinct = q.get()
    assert same and distinct >= 3


def _allgather_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_from_env('gloo')
    total = 4096
    lo, hi = parallel.shard_bounds(total, world, rank)
    rows = torch.arange(lo, hi, dtype=torch.float32).unsqueeze(1) * torch.tensor([[1., 2., 3., 4., 5.]])
    sharded = parallel.ShardedRows(total, 5, torch.device('cpu'), mode='all_gather')
    full = sharded.run(lambda a, b: rows)
    ref = torch.arange(total, dtype=torch.float32).unsqueeze(1) * torch.tensor([[1., 2., 3., 4., 5.]])
    q.put(bool(torch.equal(full, ref)))          # every rank holds the full block in this mode
    dist.barrier()
    dist.destroy_process_group()


def test_row_all_gather_mode_gloo():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_allgather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() is True and q.get() is True


# ---------------------------------------------------------------- bench.main() end to end at world size 2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(cmd, extra_env=None):
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, "exactly ONE JSON line on stdout, got %d" % len(lines)
    return json.loads(lines[0])


def _check_world2_line(d, rows_per_rank):
    assert d['n_gpus'] == 2 and d['ranks_seen'] == 2 and d['data'] == 'stub'
    assert [r['rank'] for r in d['ranks']] == [0, 1] and d['ranks'][0]['device'] != d['ranks'][1]['device']
    assert d['ms_per_step_min_rank'] <= d['ms_per_step_max_rank'] <= d['ms_per_step'] * 1.0001 + 1e-3
    gc = d['gather_check']
    assert d['gather_ms'] > 0 and (gc['ok'], gc['shards'], gc['rows'], gc['distinct_shards']) == (True, 2, 2 * rows_per_rank, 2)
    assert d['scaling'] == 'weak' and d['config']['rows_per_gpu'] == rows_per_rank
    assert abs(d['value'] - 2 * rows_per_rank / d['ms_per_step'] * 1e3) <= 2e-3 * d['value']   # whole-job aggregate
    c4 = d['config4_strong']                                   # BASELINE configs[3] rides in the same launch
    assert c4['scaling'] == 'strong' and c4['total_rows'] == 1048576 and c4['rows_per_gpu'] == 524288 and c4['value'] > 0


def test_bench_main_self_launches_world2():
    """`python bench.py --gpus 2` with no torchrun environment launches its own two ranks; the stub
    engine stands in for the device so that argument handling, sharding, the gather, config4_strong and the JSON line of
    main() are executed on CPU."""
    d = _run_bench(['bench.py', '--gpus', '2', '--stub-engine', '--steps', '3', '--warmup', '1', '--batch', '384'])
    _check_world2_line(d, 384)
    assert d['steps'] == 3 and d['warmup'] == 1


def test_bench_main_under_torchrun_world2():
    """The driver's own form: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2."""
    d = _run_bench(['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                    '--master-port', str(_free_port()), 'bench.py', '--gpus', '2', '--stub-engine', '--steps', '2',
                    '--warmup', '1', '--batch', '256', '--gather', 'all_gather'])
    _check_world2_line(d, 256)
    assert 'all_gather' in d['config']['parallelism']


def test_bench_main_forced_through_the_n_gt_1_branch_at_world_1():
    """--force-distributed: a world-size-1 process group, and every statement of the N > 1 branch (sharded rows, gather,
    gather_check, the all_gather_object of device identities, config4_strong) -- what tests/test_gpu_parallel.py runs on RCCL."""
    d = _run_bench(['bench.py', '--gpus', '1', '--stub-engine', '--steps', '2', '--warmup', '1', '--batch', '256', '--force-distributed'],
                   extra_env={'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(_free_port())})
    assert d['n_gpus'] == 1 and d['ranks_seen'] == 1 and d['collectives']['backend'] == 'gloo' and d['collectives']['forced_at_world_1']
    assert d['gather_check']['ok'] and d['gather_check']['shards'] == 1 and d['gather_ms'] > 0
    assert d['config4_strong']['rows_per_gpu'] == 1048576


def test_bench_main_strong_world2():
    d = _run_bench(['bench.py', '--gpus', '2', '--stub-engine', '--steps', '2', '--warmup', '1', '--total-rows', '1001'])
    assert d['scaling'] == 'strong' and d['ranks_seen'] == 2
    assert [r['rows'] for r in d['ranks']] == [501, 500]                       # ragged shards: point-to-point gather
    assert d['gather_check']['ok'] and d['gather_check']['rows'] == 1001
    assert 'config4_strong' not in d


@pytest.mark.parametrize("launcher", ['self', 'torchrun'])
def test_a_rank_that_dies_mid_run_ends_the_launch(launcher):
    """Pre-flight for the driver's 8-GPU lease: when one rank crashes between two steps (the survivors are then parked in the next
    barrier / gather), the launcher must come back NON-ZERO within a bounded time and without a JSON line -- not hang."""
    import subprocess
    import sys
    import time
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(ML_STUB_DIE_RANK='1', ML_STUB_DIE_AT_CALL='3')
    tail = ['bench.py', '--gpus', '2', '--stub-engine', '--steps', '4', '--warmup', '1', '--batch', '256']
    cmd = tail if launcher == 'self' else ['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                                           '127.0.0.1', '--master-port', str(_free_port())] + tail
    t0 = time.time()
    r = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    assert time.time() - t0 < 120
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')], "a failed launch must not print a result line"


def test_bench_gpus_mismatch_is_an_error():
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--stub-engine'], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and '--gpus 2 but WORLD_SIZE=1' in r.stderr


def _build_worker(local_rank, path, q):
    import sys
    import time
    sys.path.insert(0, ROOT)
    import bench
    built = []

    def build():                 # local rank 0's job: takes a while, writes the file at the end
        time.sleep(1.0)
        with open(path, 'w') as f:
            f.write('built')
        built.append(1)
    t0 = time.time()
    did = bench.ensure_library(path, local_rank, build, timeout_s=30.0, settle_s=0.1)
    q.put((local_rank, did, len(built), os.path.exists(path), time.time() - t0))


def test_on_demand_build_handoff(tmp_path):
    """bench.py on a box without the library: local rank 0 builds, the other local ranks wait for the file."""
    path = str(tmp_path / 'libfake.so')
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_build_worker, args=(r, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get() for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, did0, n0, ok0, _), (r1, did1, n1, ok1, t1) = res
    assert did0 and n0 == 1 and ok0                 # rank 0 built it
    assert did1 and n1 == 0 and ok1 and t1 >= 0.9   # rank 1 did not build, and did wait for it
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.ensure_library(path, 1, lambda: 1 / 0) is False     # present: nobody builds
