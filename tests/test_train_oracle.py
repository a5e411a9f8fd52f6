"""Pin the training-step oracle against the real reference's loop (tests/golden/golden_train.npz). CPU."""
import json
import os

import numpy as np
import pytest
import torch

import synth
from oracle.train_oracle import OracleTrainer

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def fixture_batch(mode):
    g = dict(np.load(os.path.join(G, 'golden_train_inputs.npz')))
    return torch.tensor(g[mode + '_x']), torch.tensor(g[mode + '_y'])


@pytest.mark.parametrize("mode,in_f,out_f,seed", [('mono', 34, 9, 7), ('stereo', 68, 10, 8)])
def test_oracle_training_steps_match_reference(mode, in_f, out_f, seed):
    g = dict(np.load(os.path.join(G, 'golden_train.npz')))
    x, y = fixture_batch(mode)
    tr = OracleTrainer(synth.make_state_dict(seed, in_f, out_f, 128), lr=0.001, sched_step=2, sched_gamma=0.5)
    for step in range(3):
        res, out = tr.step(x, y)
        ref = g['%s_loss%d' % (mode, step)]
        assert abs(res['loss'] - ref[0]) <= 2e-6 * abs(ref[0])
        if step == 0:
            assert np.abs(out.numpy() - g[mode + '_out0']).max() <= 1e-5
            for k, v in tr.grads().items():
                ref_g = g['%s_grad0/%s' % (mode, k)]
                assert np.abs(v.numpy() - ref_g).max() <= 1e-6 + 1e-4 * np.abs(ref_g).max(), k
    sd = tr.state_dict()
    for k, v in sd.items():
        ref_v = g['%s_final/%s' % (mode, k)]
        d = np.abs(v.numpy() - ref_v)
        assert (d > 2e-5).mean() < 0.005 and d.max() <= 3.5e-3, (k, d.max())


@pytest.mark.parametrize("mode,in_f,out_f,seed", [('mono', 34, 9, 7), ('stereo', 68, 10, 8)])
def test_oracle_autotune_loss_matches_reference(mode, in_f, out_f, seed):
    """AutoTuneMultiTaskLoss (reference train/losses.py:17-43): weighted task values, total incl. the log_sigmas, and the
    log_sigma trajectory under the shared Adam / StepLR (tests/golden/golden_train_autotune.npz)."""
    g = dict(np.load(os.path.join(G, 'golden_train_autotune.npz')))
    x, y = fixture_batch(mode)
    tr = OracleTrainer(synth.make_state_dict(seed, in_f, out_f, 128), lr=0.001, sched_step=2, sched_gamma=0.5, auto_tune_mtl=True)
    names = ['loss', 'd', 'x', 'y', 'h', 'w', 'l', 'ori'] + (['aux'] if mode == 'stereo' else [])
    for step in range(3):
        res, out = tr.step(x, y)
        ref = g['%s_loss%d' % (mode, step)]
        got = np.array([res[n] for n in names])
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (step, got, ref)
        if step == 0:
            assert np.abs(out.numpy() - g[mode + '_out0']).max() <= 1e-5
        assert np.abs(tr.log_sigmas.detach().numpy() - g['%s_log_sigmas%d' % (mode, step)]).max() <= 2e-6, step
