"""Deterministic synthetic LocoModel checkpoints and inputs shared by tests, smoke() and bench.py.

No trained monoloco checkpoint can be shipped (they live on Google Drive, reference
monoloco/predict.py:36-39), so the 1024-wide parity/bench weights are generated from a seed
with numpy's PCG64 stream: Linear weights like nn.Linear's default init (U(+-1/sqrt(fan_in))),
BatchNorm statistics shaped like the ones the reference's own fixture training produces
(running_var 0.5..12, running_mean ~N(0, 0.15..0.5), gamma ~1, beta ~0), and the heads scaled so
that the outputs span realistic values (d of metres to tens of metres).  The same function runs
in the build container (to make goldens with the real reference) and on the GPU box.
"""
import math

import numpy as np

KITTI_K = [[718.3351, 0., 600.3891], [0., 718.3351, 181.5122], [0., 0., 1.]]


def make_state_dict(seed, in_features=34, out_features=9, hidden=1024, num_stage=3):
    """Reference-keyed state_dict (numpy fp32 arrays) of a LocoModel(in, out, hidden)."""
    rng = np.random.default_rng(seed)
    sd = {}

    def linear(name, n_out, n_in, scale=1.0):
        bound = scale / math.sqrt(n_in)
        sd[name + '.weight'] = rng.uniform(-bound, bound, (n_out, n_in)).astype(np.float32)
        sd[name + '.bias'] = rng.uniform(-bound, bound, (n_out,)).astype(np.float32)

    def bnorm(name, n, mean_std, var_hi):
        sd[name + '.weight'] = (1.0 + 0.005 * rng.standard_normal(n)).astype(np.float32)
        sd[name + '.bias'] = (0.005 * rng.standard_normal(n)).astype(np.float32)
        sd[name + '.running_mean'] = (mean_std * rng.standard_normal(n)).astype(np.float32)
        sd[name + '.running_var'] = np.exp(rng.uniform(math.log(0.5), math.log(var_hi), n)).astype(np.float32)

    linear('w1', hidden, in_features)
    bnorm('batch_norm1', hidden, 0.15, 12.0)
    for s in range(num_stage):
        p = 'linear_stages.%d.' % s
        linear(p + 'w1', hidden, hidden)
        bnorm(p + 'batch_norm1', hidden, 0.3, 3.0)
        linear(p + 'w2', hidden, hidden)
        bnorm(p + 'batch_norm2', hidden, 0.15, 1.5)
    linear('w2', hidden, hidden)
    linear('w3', hidden, hidden)
    bnorm('batch_norm3', hidden, 0.5, 5.0)
    linear('w_aux', 1, hidden)
    linear('w_fin', out_features - 1, hidden)
    # give the heads realistic output statistics: theta, psi around pi/2; d 2..40 m; s ~ -2.5;
    # h, w, l small; (sin, cos) O(1)
    gain = np.array([2.0, 1.0, 60.0, 4.0, 1.0, 1.0, 1.0, 4.0, 4.0])[:out_features - 1]
    offs = np.array([1.57, 1.45, 16.0, -2.5, 0.0, 0.0, 0.0, 0.1, -0.2])[:out_features - 1]
    sd['w_fin.weight'] = (sd['w_fin.weight'] * gain[:, None].astype(np.float32)).astype(np.float32)
    sd['w_fin.bias'] = offs.astype(np.float32)
    return sd


def checksum(sd):
    """Order-independent fingerprint of a state_dict, to detect RNG drift between machines."""
    tot = 0.0
    for k in sorted(sd):
        a = np.asarray(sd[k], dtype=np.float64)
        tot += float(np.sum(a * np.cos(np.arange(a.size).reshape(a.shape) * 0.37 + 0.1)))
    return tot


def make_keypoints(m, seed=0, width=1238, height=374):
    """Throughput set T: u~U(0,W), v~U(0,H), c~U(0,1); (m,3,17) fp32."""
    rng = np.random.default_rng(seed)
    kps = np.empty((m, 3, 17), dtype=np.float32)
    kps[:, 0] = rng.uniform(0, width, (m, 17))
    kps[:, 1] = rng.uniform(0, height, (m, 17))
    kps[:, 2] = rng.uniform(0, 1, (m, 17))
    return kps


def make_poses(m, seed=0, width=1238, height=374):
    """Person-like poses: a random box (height 40..300 px) filled with 17 jittered joints -- closer to
    real pifpaf output than set T, so that network outputs stay in a realistic range."""
    rng = np.random.default_rng(seed)
    hgt = rng.uniform(40, 300, (m, 1))
    wid = hgt * rng.uniform(0.25, 0.5, (m, 1))
    cx = rng.uniform(0.05 * width, 0.95 * width, (m, 1))
    cy = rng.uniform(0.35 * height, 0.75 * height, (m, 1))
    # canonical skeleton in box coordinates (x in [-.5,.5], y in [-.5,.5]), COCO order
    sk_x = np.array([0, .05, -.05, .12, -.12, .3, -.3, .4, -.4, .42, -.42, .18, -.18, .2, -.2, .2, -.2])
    sk_y = np.array([-.45, -.47, -.47, -.45, -.45, -.3, -.3, -.1, -.1, .05, .05, .05, .05, .28, .28, .48, .48])
    kps = np.empty((m, 3, 17), dtype=np.float32)
    kps[:, 0] = cx + wid * (sk_x[None] + 0.04 * rng.standard_normal((m, 17)))
    kps[:, 1] = cy + hgt * (sk_y[None] + 0.02 * rng.standard_normal((m, 17)))
    kps[:, 2] = rng.uniform(0.2, 1, (m, 17))
    return kps


def big_train_batch(x, y, m, seed):
    """A training batch of m rows drawn (with replacement) from the fixture batch (x, y numpy fp32), keypoints jittered by
    N(0, 0.01) -- the same rows for oracle/make_golden.py (reference run) and the GPU tests."""
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, x.shape[0], size=m)
    xb = x[idx] + rng.normal(0, 0.01, size=(m, x.shape[1])).astype(np.float32)
    return xb.astype(np.float32), y[idx].astype(np.float32)


def make_boxes(m, g, seed, ties=True, width=1238, height=374):
    """Detection boxes [x1, y1, x2, y2, conf] (m) and ground-truth boxes [x1, y1, x2, y2] (g) as nested Python lists, from
    Python's own Mersenne Twister (random.Random(seed): the same doubles on every machine).  About 70 % of the ground-truth
    boxes are jittered copies of a detection; with `ties` there are repeated confidences (argsort ties), exact copies of
    detection boxes (IoU 1.0 several times over), duplicated ground-truth boxes (np.argmax ties) and duplicated left edges
    (reorder ties).  The same function feeds oracle/make_golden.py (reference run) and the tests."""
    import random
    rnd = random.Random(seed)
    boxes = []
    for _ in range(m):
        x, y = rnd.uniform(0, width - 120), rnd.uniform(0, height - 220)
        conf = rnd.choice([0.25, 0.5, 0.75]) if ties and rnd.random() < 0.4 else rnd.random()
        boxes.append([x, y, x + rnd.uniform(20, 100), y + rnd.uniform(40, 200), conf])
    if ties and m > 8:
        boxes[5][0] = boxes[2][0]      # equal left edges
        boxes[7] = list(boxes[1])      # a detection listed twice (same box, same confidence)
    gt = []
    for i in range(g):
        if m and rnd.random() < 0.7:
            b = boxes[i % m]
            if ties and rnd.random() < 0.3:
                gt.append(list(b[:4]))
            else:
                gt.append([b[0] + rnd.uniform(-10, 10), b[1] + rnd.uniform(-10, 10), b[2] + rnd.uniform(-10, 10),
                           b[3] + rnd.uniform(-10, 10)])
        else:
            x, y = rnd.uniform(0, width - 120), rnd.uniform(0, height - 220)
            gt.append([x, y, x + rnd.uniform(20, 100), y + rnd.uniform(40, 200)])
    rnd.shuffle(gt)
    if ties and g > 4:
        gt[3] = list(gt[1])
        gt[-1] = list(gt[0])
    return boxes, gt
