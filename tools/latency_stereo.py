"""End-to-end latency of the drop-in Loco.forward for one STEREO image pair (MonStereo): lists in, dictionary out, host side included."""
import copy, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch, synth
from monoloco_amd.network import Loco, load_calibration, preprocess_pifpaf
from monoloco_amd.network.architectures import LocoModel
dev = torch.device('cuda', 0)
model = LocoModel(68, 10, 1024)
model.load_state_dict({k: torch.tensor(v) for k, v in synth.make_state_dict(3, 68, 10, 1024).items()})
net = Loco(model=model, mode='stereo', device=dev)
ann = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'pifpaf_002282.json')))
boxes, kps = preprocess_pifpaf(copy.deepcopy(ann), im_size=(1238, 374), enlarge_boxes=False)
kps_r = [[[u - 12.0 for u in k[0]], k[1], k[2]] for k in kps[:5]]
kk = load_calibration('kitti', (1238, 374))
for _ in range(50):
    dic = net.forward(kps, kk, keypoints_r=kps_r)
torch.cuda.synchronize()
n = 500
t0 = time.perf_counter()
for _ in range(n):
    dic = net.forward(kps, kk, keypoints_r=kps_r)
t1 = time.perf_counter()
for _ in range(n):
    out = net.post_process(dic, boxes, kps, kk)
t2 = time.perf_counter()
print("stereo Loco.forward: %d x %d persons, %.1f us per call; post_process %.1f us per call" % (len(kps), len(kps_r), (t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6))
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): dic = net.forward(kps, kk, keypoints_r=kps_r)
    pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
