"""Build-time guards of the hand-scheduled kernels (CPU only: hipcc cross-compiles gfx950 here, nothing runs).

dense_kernel_w4 (csrc/dense_kernel_w4.h) feeds its LDS ring with LDS-DMA instructions issued from inline asm -- invisible to
hipcc's vmcnt bookkeeping -- and waits for them with hand-counted `s_waitcnt vmcnt(16 | 12 | 10)`; its accumulators fill all 256
AGPRs and its epilogue pads MFMA -> AGPR-read hazards by hand.  A compiler point release (or an innocent source change) that
spills a register, hoists a vector load into the loop or inserts its own `s_waitcnt vmcnt(0)` turns this into a data race that
the parity tests catch only when timing cooperates (profiles/r03_xgemm_occupancy.md is the record of one such race).  These tests
read the assembly and the resource remarks hipcc emits for the very sources and flags the library is built from
(tools/check_loops.py; cached under monoloco_amd/lib/asm/ until a source changes) and assert the loop's instruction mix
exactly.  The layer these kernels compute: monoloco/network/architectures.py:88-102 (Linear + BatchNorm + ReLU of a stage)."""
import concurrent.futures
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import check_loops as C  # noqa: E402


@pytest.fixture(scope='module')
def reports():
    """{translation unit: {kernel: {'loop': instruction mix of the main loop, 'resources': hipcc's remarks}}} -- the two units are
    compiled side by side (about 45 s when the cache is cold)."""
    with concurrent.futures.ThreadPoolExecutor(2) as ex:
        futs = {u: ex.submit(C.report, u, (), 'dense_') for u in ('monoloco_hip', 'train')}
        return {u: f.result() for u, f in futs.items()}


def _w4(reports):
    return {k: v for u in reports.values() for k, v in u.items() if k.startswith('dense_kernel_w4<')}


def _args(name):
    a = re.match(r'dense_kernel_w4<(-?\d+), (\w+), (\w+), (-?\d+), (\w+), (\d+)(?:, \w+)?>', name).groups()   # (7th: non-temporal residual loads)
    return int(a[0]), a[1] == 'true', a[2] == 'true', int(a[3]), a[4] == 'true', int(a[5])


def test_every_w4_instantiation_is_found(reports):
    names = set(_w4(reports))
    assert len(names) >= 36, len(names)
    # the headline's six long-K layers, the half-size tile of the upper mid window, the training GEMMs (fp32 out, split-K, reduction-major)
    # (7th template argument, round 6: non-temporal residual loads -- the two residual layers of a batch whose activations outgrow the
    #  Infinity Cache)
    for need in ('dense_kernel_w4<3, true, false, 0, false, 4, false>', 'dense_kernel_w4<3, true, true, -1, false, 4, false>',
                 'dense_kernel_w4<3, true, true, 0, false, 4, true>', 'dense_kernel_w4<3, true, true, -1, false, 4, true>',
                 'dense_kernel_w4<3, true, false, 0, false, 2, false>', 'dense_kernel_w4<3, false, false, -2, false, 4, false>',
                 'dense_kernel_w4<3, false, false, -3, true, 4, false>'):
        assert need in names, need


def test_no_dense_kernel_spills_or_uses_scratch(reports):
    """A spill store is a vector-memory operation: it would sit in the counted vmcnt queue of the LDS-DMA ring."""
    for unit in reports.values():
        for name, v in unit.items():
            r = v['resources']
            assert r, name
            # (SGPR spills go to VGPR lanes -- v_writelane, no memory operation -- and are allowed; scratch 0 proves none reached memory)
            assert int(r.get('ScratchSize [bytes/lane]', 0)) == 0 and int(r.get('VGPRs Spill', 0)) == 0, (name, r)
            if v['loop'] is not None:
                assert v['loop'].get('scratch', 0) == 0, name


def test_w4_register_budget(reports):
    """One wave per SIMD with the whole register file: the full-size tile's 256 accumulators fill the AGPRs exactly, the half-size
    tile's 128 (+ the head variants' staging) stay below; VGPR + AGPR <= 512."""
    for name, v in _w4(reports).items():
        nsplit, relu, res, head, trans, nj = _args(name)
        r = v['resources']
        agpr, vgpr = int(r['AGPRs']), int(r['VGPRs'])
        assert vgpr + agpr <= 512, (name, vgpr, agpr)
        assert int(r['Occupancy [waves/SIMD]']) == 1, (name, r['Occupancy [waves/SIMD]'])
        if nj == 4:
            assert agpr == 256, (name, agpr)
        else:
            assert 128 <= agpr <= 192, (name, agpr)
        assert int(r['LDS Size [bytes/block]']) == 163840, (name, r['LDS Size [bytes/block]'])   # the 4-slot ring + epilogue buffers = all 160 KiB


def test_w4_main_loop_instruction_mix(reports):
    """Per loop iteration = two k32 steps.  3-product mode, full tile: 192 MFMAs, 32 LDS-DMA (2 steps x 2 slots x 8 quarters), four
    barriers each behind a counted vmcnt(16); half tile: 96 MFMAs, 24 LDS-DMA, vmcnt(12) before phase A and vmcnt(10) before phase
    B; single-product modes: one phase per step, the slot requested one phase earlier -> vmcnt(0).  No other vector-memory
    instruction, no LDS write, no scratch access, and no wait hipcc added on its own."""
    for name, v in _w4(reports).items():
        nsplit, relu, res, head, trans, nj = _args(name)
        loop = v['loop']
        assert loop is not None, name
        if nsplit == 3:
            want = {'mfma': 48 * nj, 'lds_dma': 4 * (4 + nj), 'barrier': 4, 'ds_read': (32 if trans else 16) * 4 * (4 + nj) // 8,
                    'vmcnt_waits': {16: 4} if nj == 4 else {10: 2, 12: 2}}
        else:
            want = {'mfma': 16 * nj, 'lds_dma': 2 * (4 + nj), 'barrier': 2, 'ds_read': 4 * (4 + nj), 'vmcnt_waits': {0: 2}}
        assert loop == want, (name, loop, want)


def test_pp_and_mid_loops_keep_their_shape(reports):
    """dense_kernel_pp (input layer + fused-head layer) also counts its LDS-DMA by hand: per k32 step 14 DMA instructions, four
    barriers, waits vmcnt(2) once and vmcnt(0) twice; dense_mid_kernel uses ordinary loads hipcc counts itself -- its loop must
    hold one barrier per step and no LDS-DMA."""
    seen_pp = seen_mid = 0
    for name, v in reports['monoloco_hip'].items():
        loop = v['loop']
        if name.startswith('dense_kernel_pp<') and loop is not None:
            nsplit = int(name.split('<')[1].split(',')[0])
            assert loop.get('lds_dma') == 14 and loop.get('barrier') == 4 and loop.get('vmcnt_waits') == {0: 2, 2: 1}, (name, loop)
            assert loop.get('mfma') == {3: 48, 1: 16, 0: 16}[nsplit] and 'other_vmem' not in loop and 'ds_write' not in loop, (name, loop)
            seen_pp += 1
        targs = name[name.index('<') + 1:name.rindex('>')].split(', ') if name.startswith('dense_mid_kernel<') else []
        is_dma = len(targs) >= 7 and targs[6] == 'true'        # <NSPLIT, RELU, RES, TM, HEAD, SPLITK, DMA, PREP>
        is_prep = len(targs) >= 8 and targs[7] == 'true'
        if is_prep and loop is not None:
            # the input layer with the pre-process inside (K = 64: both stages requested in front of the loop): no request, no vector-memory
            # instruction inside the loop at all
            assert 'lds_dma' not in loop and 'other_vmem' not in loop and loop.get('barrier') == 2, (name, loop)
            continue
        if name.startswith('dense_mid_kernel<') and loop is not None and is_dma:
            # round 6, the LDS-DMA loader (last template argument): two steps per iteration (the two fragment register sets), per step
            # NI = (128 + TM) / 32 requests per wave, ONE counted wait vmcnt(NI), one barrier, the next step's fragment reads (2 half-steps
            # x (4 W + 2 TM / 64 X) ds_read_b128); no register-staged load or LDS store left, and no wait of hipcc's own in front of the
            # MFMAs (a second LDS object in the kernel once made it guard every fragment read with vmcnt(0); a branch around the reads
            # made it wait lgkmcnt(0) at the join -- both seen here first)
            tm = int(name.split(',')[3])
            ni = (128 + tm) // 32
            ring = 3                         # stages of the LDS ring; the counted wait leaves ring - 2 stages' requests in flight
            assert loop.get('barrier') == 2 and loop.get('lds_dma') == 2 * ni and loop.get('vmcnt_waits') == {(ring - 2) * ni: 2}, (name, loop)
            assert int(v['resources']['LDS Size [bytes/block]']) == ring * (128 + tm) * 128, (name, v['resources']['LDS Size [bytes/block]'])
            assert loop.get('mfma') == 2 * 3 * 4 * tm // 64 and loop.get('ds_read') == 2 * 2 * (4 + 2 * tm // 64), (name, loop)
            assert 'other_vmem' not in loop and 'ds_write' not in loop, (name, loop)
            seen_mid += 1
            continue
        if name.startswith('dense_mid_kernel<') and loop is not None:
            nsplit, tm = int(name.split('<')[1].split(',')[0]), int(name.split(',')[3])
            steps = loop.get('barrier')    # hipcc keeps one k32 step per iteration, or two (the split-K variants): one barrier each
            assert steps in (1, 2) and 'lds_dma' not in loop and loop.get('mfma') == steps * (3 if nsplit == 3 else 1) * 4 * tm // 64, (name, loop)
            seen_mid += 1
    assert seen_pp >= 12 and seen_mid >= 8, (seen_pp, seen_mid)


def test_the_guard_bites(tmp_path):
    """The same analysis on an ablation build whose loop really differs (-DML_W4_ABL=4: no LDS-DMA in the loop): the mix test's
    expectation must fail on it -- a guard that cannot fail guards nothing.  (One more ~20 s device-only compile of train.hip,
    the smaller unit.)"""
    rep = C.report('train', ('-DML_W4_ABL=4',), 'dense_kernel_w4')
    loop = rep['dense_kernel_w4<3, false, false, -2, false, 4, false>']['loop']
    assert loop['mfma'] == 192 and loop.get('lds_dma', 0) == 0, loop
