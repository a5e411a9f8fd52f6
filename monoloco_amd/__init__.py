"""monoloco_amd -- MI355X (gfx950) implementation of monoloco's keypoint -> 3D inference hot path.

Hand-written HIP kernels behind a C ABI (``include/monoloco_hip.h``) and a Python host side that
mirrors the reference surface (``monoloco_amd.network``, ``monoloco_amd.utils``).  See DESIGN.md.
"""
__version__ = '0.1.0'
