"""End-to-end latency of the drop-in Loco.forward (+ post_process) for one image: Python lists in, dictionary of CPU
tensors out -- host overhead included."""
import copy, json, os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch, synth
from monoloco_amd.network import Loco, load_calibration, preprocess_pifpaf
from monoloco_amd.network.architectures import LocoModel
dev = torch.device('cuda', 0)
model = LocoModel(34, 9, 1024)
model.load_state_dict({k: torch.tensor(v) for k, v in synth.make_state_dict(1).items()})
net = Loco(model=model, mode='mono', device=dev)
ann = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'pifpaf_002282.json')))
boxes, kps = preprocess_pifpaf(copy.deepcopy(ann), im_size=(1238, 374), enlarge_boxes=False)
kk = load_calibration('kitti', (1238, 374))
for _ in range(50):
    dic = net.forward(kps, kk)
torch.cuda.synchronize()
n = 500
t0 = time.perf_counter()
for _ in range(n):
    dic = net.forward(kps, kk)
t1 = time.perf_counter()
for _ in range(n):
    out = net.post_process(dic, boxes, kps, kk)
t2 = time.perf_counter()
print("Loco.forward: %d persons, %.1f us per call; post_process %.1f us per call" % (len(kps), (t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6))
if len(sys.argv) > 1:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): dic = net.forward(kps, kk)
    pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): out = net.post_process(dic, boxes, kps, kk)
    pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
