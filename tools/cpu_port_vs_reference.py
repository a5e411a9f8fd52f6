"""cpu_baseline calibration (build container only: needs /root/reference): the oracle ("port", what bench.py times on the
GPU box, where the reference does not exist) against the REAL reference on the same CPU, same threads, same batch.

Reference leg (SURVEY 8d "CPU baseline"): preprocess_monoloco -> LocoModel (eval, no_grad) -> extract_outputs on tensors
(list marshalling skipped) -> the post_process geometry as tensors (get_keypoints, pixel_to_camera, xyz_from_distance).
Oracle leg: oracle/monoloco_oracle.forward_mono.  3 warm-ups + 7 repetitions, median, per thread count.
Writes profiles/r05_port_vs_reference.json; bench.py replays its ratio as cpu_baseline.port_vs_reference."""
import json
import os
import statistics
import sys
import time
import types

os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in ('torchvision', 'torchvision.transforms'):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
sys.path[:0] = ['/root/reference', ROOT, os.path.join(ROOT, 'tests')]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import synth  # noqa: E402
from monoloco.network.architectures import LocoModel  # noqa: E402
from monoloco.network.process import extract_outputs, preprocess_monoloco  # noqa: E402
from monoloco.utils import get_keypoints, pixel_to_camera, xyz_from_distance  # noqa: E402
from oracle import monoloco_oracle as O  # noqa: E402


def main():
    sd_np = synth.make_state_dict(1, 34, 9, 1024)
    sd = {k: torch.tensor(v) for k, v in sd_np.items()}
    model = LocoModel(34, 9, 1024, device='cpu')
    model.load_state_dict(sd)
    model.eval()
    kk = synth.KITTI_K
    res = {}
    for threads, n in ((1, 2048), (2, 2048), (4, 8192), (8, 8192)):
        if threads > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(threads)
        kps = torch.tensor(synth.make_keypoints(n, seed=100))

        def reference():
            with torch.no_grad():
                x = preprocess_monoloco(kps, kk)
                out = extract_outputs(model(x))
                uv = get_keypoints(kps, mode='center')
                xy = pixel_to_camera(uv, torch.tensor(kk), 1)
                xyz = xyz_from_distance(out['d'], xy)
                return torch.cat((xyz, out['d'], out['bi']), dim=1)

        def port():
            return O.forward_mono(sd, kps, kk)['xyzds']

        same = float((reference() - port()).abs().max())
        t = {}
        for name, fn in (('reference', reference), ('port', port), ('reference2', reference), ('port2', port)):
            for _ in range(3):
                fn()
            ts = []
            for _ in range(7):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            t[name] = n / statistics.median(ts)
        ref_ps = 0.5 * (t['reference'] + t['reference2'])
        port_ps = 0.5 * (t['port'] + t['port2'])
        res[str(threads)] = {"rows": n, "reference_persons_per_s": round(ref_ps, 1), "port_persons_per_s": round(port_ps, 1),
                             "port_vs_reference": round(port_ps / ref_ps, 4), "max_abs_difference_of_outputs": same}
        print(threads, res[str(threads)], flush=True)
    ratios = [v["port_vs_reference"] for v in res.values()]
    out = {"what": "oracle (bench.py's cpu_baseline, kind 'port') vs the real reference on the same CPU, threads and batch",
           "cpu_model": next((ln.split(':', 1)[1].strip() for ln in open('/proc/cpuinfo') if ln.startswith('model name')), '?'),
           "host_cores": os.cpu_count(), "torch": torch.__version__, "by_threads": res,
           "port_vs_reference": round(statistics.median(ratios), 4),
           "note": "ratio < 1: the port is a little slower than the reference on this CPU (same arithmetic, bit-identical outputs; it builds a few more "
                   "intermediate tensors); reference persons/s on the GPU box = cpu_baseline.value / ratio"}
    path = os.path.join(ROOT, 'profiles', 'r05_port_vs_reference.json')
    json.dump(out, open(path, 'w'), indent=1)
    print("wrote", path, "ratio", out["port_vs_reference"])


if __name__ == '__main__':
    main()
