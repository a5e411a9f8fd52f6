"""Activity rules (reference monoloco/activity.py) against goldens from the reference's own functions
(oracle/make_golden.py activity).  The deterministic rules run on CPU; the probabilistic social-distance flag
samples on the device and is marked gpu."""
import json
import os
from types import SimpleNamespace

import pytest

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_activity.json')))


def test_is_raising_hand_matches_reference():
    from monoloco_amd.activity import is_raising_hand
    outs = [is_raising_hand(c['kp']) for c in G['raising']]
    assert outs == [c['out'] for c in G['raising']]
    assert {None, 'left', 'right', 'both'} == set(outs)


def test_check_f_formations_matches_reference():
    from monoloco_amd.activity import check_f_formations
    for c in G['fform']:
        got = check_f_formations(c['i'], c['j'], c['centers'], c['angles'], tuple(c['radii']), social_distance=c['sd'])
        assert bool(got) == c['out']


def test_social_interactions_deterministic_matches_reference():
    from monoloco_amd.activity import social_interactions
    for c in G['social_det']:
        n = len(c['centers'])
        got = [bool(social_interactions(i, c['centers'], c['angles'], c['dds'], stds=[0.1] * n, n_samples=1,
                                        threshold_dist=2.5, radii=(0.3, 0.5, 1))) for i in range(n)]
        assert got == c['out']


@pytest.mark.gpu
def test_social_distance_probabilistic_and_loco_hooks(hip_lib, cuda_device):
    """Loco.social_distance / Loco.raising_hand on a post_process-like dictionary: clear-cut scenes (facing each
    other at 1 m / back to back) give the reference's flags although the Laplace draws come from another generator."""
    from monoloco_amd.network import Loco
    args = SimpleNamespace(threshold_prob=0.25, threshold_dist=2.5, radii=(0.3, 0.5, 1))
    for c in G['social_prob']:
        dic = {'xyz_pred': [[x, 1.0, z] for x, z in c['centers']], 'angles': c['angles'], 'dds_pred': c['dds'],
               'stds_ale': c['stds']}
        assert Loco.social_distance(dic, args)['social_distance'] == c['out']
    kps = [c['kp'] for c in G['raising'][:20]]
    assert Loco.raising_hand({}, kps)['raising_hand'] == [c['out'] for c in G['raising'][:20]]
