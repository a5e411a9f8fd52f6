"""Activity rules that consume ``Loco.post_process`` output (reference monoloco/activity.py:17-165): raised hands
from the 2D pose, and social-distance / F-formation flags from the 3D centres, orientations and the Laplace
spread of the distance.  Host logic on a handful of persons; the only device work is the Laplace sampling
(``ml_laplace_sampling``), so the probabilistic flag agrees with the reference statistically, not draw by draw.
"""
import math

import numpy as np
import torch

from .network.process import laplace_sampling

# COCO keypoint indices used by the rules
_NOSE, _L_EAR, _R_EAR, _L_SHOULDER, _R_SHOULDER, _L_ELBOW, _R_ELBOW, _L_HAND, _R_HAND = 0, 3, 4, 5, 6, 7, 8, 9, 10


def _elbow_angle(kp, hand, elbow, shoulder):
    """(90/pi) * angle at the elbow between forearm and upper arm (reference activity.py:88-98)."""
    forearm = [kp[0][hand] - kp[0][elbow], kp[1][hand] - kp[1][elbow]]
    arm = [kp[0][shoulder] - kp[0][elbow], kp[1][shoulder] - kp[1][elbow]]
    return (90 / np.pi) * np.arccos(np.dot(forearm / np.linalg.norm(forearm), arm / np.linalg.norm(arm)))


def is_raising_hand(kp):
    """'left' / 'right' / 'both' / None for one person's keypoints [xs, ys, cs] (reference activity.py:70-118): a hand
    counts as raised when it is above its shoulder, the elbow is opened by at least 30 (in the reference's 90/pi
    units) and the hand is not tucked in between shoulder and head top."""
    head_top = kp[1][_NOSE] - (kp[0][_L_EAR] - kp[0][_R_EAR])
    flags = {}
    for side, hand, elbow, shoulder, inward in (('left', _L_HAND, _L_ELBOW, _L_SHOULDER, lambda h, s: h <= s),
                                                ('right', _R_HAND, _R_ELBOW, _R_SHOULDER, lambda h, s: h >= s)):
        up = kp[1][hand] < kp[1][shoulder]
        too_close = inward(kp[0][hand], kp[0][shoulder]) and kp[1][hand] >= head_top
        flags[side] = bool(up and _elbow_angle(kp, hand, elbow, shoulder) >= 30 and not too_close)
    if flags['left'] and flags['right']:
        return 'both'
    if flags['left']:
        return 'left'
    if flags['right']:
        return 'right'
    return None


def check_f_formations(idx, idx_t, centers, angles, radii, social_distance=False):
    """Do persons idx and idx_t form an F-formation (reference activity.py:121-165)?  For each o-space radius: the
    o-space centre is the midpoint of the two points one radius ahead of each person along its orientation; the pair
    qualifies when those two points are at most as far apart as either person is from the centre (they look
    inwards) and no third person stands inside the radius."""
    others = np.array([c for k, c in enumerate(centers) if k not in (idx, idx_t)], dtype=float)
    x_0 = np.array([float(centers[idx][0]), float(centers[idx][1])])
    x_1 = np.array([float(centers[idx_t][0]), float(centers[idx_t][1])])
    for radius in radii:
        mu_0 = x_0 + radius * np.array([math.cos(angles[idx]), -math.sin(angles[idx])])
        mu_1 = x_1 + radius * np.array([math.cos(angles[idx_t]), -math.sin(angles[idx_t])])
        o_c = (mu_0 + mu_1) / 2
        d_new = np.linalg.norm(mu_0 - mu_1)
        if social_distance:
            d_new = d_new / 2
        nearest_other = np.min(np.linalg.norm(others - o_c.reshape(1, -1), axis=1)) if others.size else 100.0
        if d_new <= min(np.linalg.norm(x_0 - o_c), np.linalg.norm(x_1 - o_c)) and nearest_other > radius:
            return True
    return False


def social_interactions(idx, centers, angles, dds, stds=None, social_distance=False, n_samples=100, threshold_prob=0.25,
                        threshold_dist=2, radii=(0.3, 0.5)):
    """Alert flag of person idx (reference activity.py:17-67): among the persons within threshold_dist (nearest
    first), does one form an F-formation with idx -- deterministically (n_samples < 2), or in at least
    threshold_prob of n_samples scenes in which both persons are moved along their viewing rays by Laplace-sampled
    distance errors."""
    xx, zz = centers[idx][0], centers[idx][1]
    distances = [math.sqrt((xx - c[0]) ** 2 + (zz - c[1]) ** 2) for c in centers]
    close = [int(k) for k in np.argsort(distances)[1:] if distances[k] <= threshold_dist]
    if n_samples < 2:
        return any(check_f_formations(idx, k, centers, angles, radii=radii, social_distance=social_distance) for k in close)
    dds_t = torch.tensor(dds).view(-1, 1)
    samples = laplace_sampling(torch.cat((dds_t, torch.tensor(stds).view(-1, 1)), dim=1), n_samples=n_samples)
    for k in close:
        hits = 0
        for s in range(n_samples):
            scene = [list(c) for c in centers]
            for el in (idx, k):
                delta = float(dds_t[el]) - float(samples[s, el])
                theta = math.atan2(scene[el][1], scene[el][0])
                scene[el][0] += delta * math.cos(theta)
                scene[el][1] += delta * math.sin(theta)
            hits += bool(check_f_formations(idx, k, scene, angles, radii=radii, social_distance=social_distance))
        if hits / n_samples >= threshold_prob:
            return True
    return False
