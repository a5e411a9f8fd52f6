"""A CPU stand-in for `monoloco_amd.engine.LocoEngine` -- TEST INFRASTRUCTURE ONLY.

`bench.py --stub-engine` swaps it in so that the N > 1 launch contract of bench.main() (self-launch under
torch.distributed.run, rank / shard bookkeeping, the one gather, `config4_strong`, `ranks_seen`, JSON emission) can be
executed end to end on a box without a GPU (gloo backend, world size 2).  It computes NOTHING of the hot path: the
(x, y, z, d, sigma) block it returns is a cheap deterministic function of the keypoints, the line it produces is marked
`"data": "stub"` and its `value` measures nothing.  The product (`monoloco_amd/`) never imports this file.
"""
import torch


class StubEngine:
    out_features = 9
    num_layers = 8

    def __init__(self, state_dict, device=None, precision='f16x2', merge_w2w3=True, reserve_rows=0):
        self.device = torch.device('cpu')
        self.closed = False
        self.reserved = int(reserve_rows)
        self.calls = 0

    def set_tuning(self, **kw):
        pass

    def reserve(self, rows):
        self.reserved = max(self.reserved, int(rows))

    def profile_begin(self, n):
        pass

    def profile_end(self):
        return None

    def forward_mono(self, kps, kinv, box_conf=None, out=None, xyzds=None, raw=None):
        assert not self.closed
        assert kps.shape[0] <= self.reserved, "forward on more rows than were reserved"
        self.calls += 1
        # fault injection for the launch-contract test: this rank dies (no clean-up, like a crashed process) at its n-th call
        import os
        if os.environ.get('ML_STUB_DIE_RANK') == os.environ.get('RANK', '0') and self.calls == int(os.environ.get('ML_STUB_DIE_AT_CALL', '0')):
            os._exit(17)
        if xyzds is not None:   # rows stay identifiable: the gathered block can be checked against the shards
            xyzds.copy_(kps[:, 0, :5])
        if raw is not None:
            raw.copy_(kps[:, 1, :9])
        return {'xyzds': xyzds}

    def close(self):
        self.closed = True
