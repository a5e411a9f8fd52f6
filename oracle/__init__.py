"""TEST INFRASTRUCTURE ONLY.

CPU restatement ("oracle") of the monoloco keypoint->3D hot path.  Nothing in
``monoloco_amd`` may import from here; only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` do, and only as the checker.
"""
