import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, synth
from monoloco_amd import engine
from oracle import monoloco_oracle as O
G = os.path.join(ROOT, 'tests', 'golden')
gold = dict(np.load(os.path.join(G, 'golden_path.npz')))
sd = {k: torch.tensor(v) for k, v in np.load(os.path.join(G, 'ckpt_mono_h256.npz')).items()}
kps = torch.tensor(gold['mono_kps'][:256])
dev = torch.device('cuda', 0)
eng = engine.LocoEngine(sd, device=dev)
kinv = engine.inverse_intrinsics(synth.KITTI_K)
NP = 300
epi, passes = eng.epistemic_mono(kps, kinv, NP, 0.2, want_passes=True)
passes = passes.cpu(); epi = epi.cpu()
x = O.preprocess_monoloco(kps, synth.KITTI_K)
torch.manual_seed(0)
ref_passes = torch.stack([O.loco_forward_mc(sd, x, 0.2) for _ in range(NP)])
for col, name in ((2, 'd'), (3, 's')):
    a, b = passes[:, :, col], ref_passes[:, :, col]
    print(name, 'mean-over-passes: max rel diff of means %.3f ; std ratio (HIP/oracle) median %.3f min %.3f max %.3f' % (
        ((a.mean(0) - b.mean(0)).abs() / b.std(0)).max(), (a.std(0) / b.std(0)).median(), (a.std(0) / b.std(0)).min(), (a.std(0) / b.std(0)).max()))
# closed form of the estimator given pass statistics and an ideal Laplace (var 2): var = Var(mu) + 2 E[b^2]
mu, bb = ref_passes[:, :, 2], (torch.exp(ref_passes[:, :, 3]) * ref_passes[:, :, 2]).abs()
ideal = torch.sqrt(mu.var(0) + 2 * (bb ** 2).mean(0))
mu2, bb2 = passes[:, :, 2], (torch.exp(passes[:, :, 3]) * passes[:, :, 2]).abs()
ideal2 = torch.sqrt(mu2.var(0) + 2 * (bb2 ** 2).mean(0))
torch.manual_seed(0)
ref = O.epistemic_uncertainty(sd, x, NP, 0.2)
print('oracle MC / ideal(oracle passes): median %.3f range %.2f..%.2f' % ((ref / ideal).median(), (ref / ideal).min(), (ref / ideal).max()))
print('HIP MC / ideal(HIP passes):     median %.3f range %.2f..%.2f' % ((epi / ideal2).median(), (epi / ideal2).min(), (epi / ideal2).max()))
print('ideal(HIP passes)/ideal(oracle passes): median %.3f range %.2f..%.2f' % ((ideal2 / ideal).median(), (ideal2 / ideal).min(), (ideal2 / ideal).max()))
print('fraction of variance that is aleatoric (oracle): median %.2f' % ((2 * (bb ** 2).mean(0)) / ideal ** 2).median())
