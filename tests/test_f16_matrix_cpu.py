"""Floors under the generated (shape, tile variant) matrix of the fp16 convolution tests (tests/f16_matrix.py).

The GPU tests collect only the pairs `mp_f16_conv_supported` reports as served, so a kernel family that silently stopped serving its
shapes would shrink the matrix instead of failing it.  These host-only checks (the query runs the entry point's own checks and the
family's dispatch without a launch - no GPU needed) pin the matrix from below:
  * every forced variant id serves at least one case of its family's table, every case is served by at least one variant;
  * the (shape, variant) pairs the tuner picked for the three bench plans on an MI355X (tests/golden/bench_plan_picks.json, written
    by tools/dump_plan_picks.py) are still served by the forced-variant entry;
  * the statistics builds the weight-stationary kernel leaves out for register budget are exactly the listed ones.
"""
import json
import os

import pytest

from tests import f16_matrix as fm

HERE = os.path.dirname(os.path.abspath(__file__))


def _coverage(cases, variants, desc_of, res_of=lambda c: 0, stats=0, **env):
    pairs = fm.served_pairs(cases, variants, desc_of, res_of, stats, **env)
    by_variant = {v: 0 for v in variants}
    by_case = {i: 0 for i in range(len(cases))}
    for pr in pairs:
        case, v = pr.values
        by_variant[v] += 1
        by_case[cases.index(case)] += 1
    return pairs, by_variant, by_case


FAMILIES = [
    # name, cases, variants, desc_of, res_of, knob sets (a pair counts when ANY knob set serves it), floor on the matrix size
    ("one-tile", fm.CONV_CASES, fm.TILE_VARIANTS, fm.conv_case_desc, fm.conv_case_res, [{}], 120),
    ("multi-tile", fm.CONV_CASES, fm.MT_VARIANTS, fm.conv_case_desc, fm.conv_case_res, [dict(MP_F16_MT_GROUPS=1), dict(MP_F16_MT_GROUPS=3)], 95),
    ("weights-in-registers", fm.WREG_CASES, fm.WREG_VARIANTS, fm.conv_case_desc, fm.conv_case_res, [{}], 90),
    ("weights-in-registers stride 2", fm.WREG_S2_CASES, fm.WREG_S2_VARIANTS, fm.s2_case_desc, fm.s2_case_res, [{}], 28),
    ("weight-stationary", fm.WS_CASES, fm.WS_VARIANTS, fm.conv_case_desc, fm.conv_case_res, [dict(MP_F16_WS_GROUPS=2), dict(MP_F16_WS_GROUPS=5)], 17),
    ("four-phase data gradient", fm.PHASES4_CASES, [0, 4], fm.phases4_case_desc, lambda c: 0, [{}], 8),
]


@pytest.mark.parametrize("family", FAMILIES, ids=[f[0] for f in FAMILIES])
def test_every_forced_variant_serves_a_case_and_every_case_is_served(family):
    name, cases, variants, desc_of, res_of, knob_sets, floor = family
    v_total = {v: 0 for v in variants}
    c_total = {i: 0 for i in range(len(cases))}
    for env in knob_sets:
        pairs, by_variant, by_case = _coverage(cases, variants, desc_of, res_of, 0, **env)
        assert len(pairs) >= floor, f"{name} {env}: the served matrix shrank to {len(pairs)} pairs (floor {floor})"
        for v, k in by_variant.items():
            v_total[v] += k
        for i, k in by_case.items():
            c_total[i] += k
    assert not [v for v, k in v_total.items() if k == 0], f"{name}: variants that serve none of the family's cases"
    assert not [i for i, k in c_total.items() if k == 0], f"{name}: cases no variant of the family serves"


def test_weight_stationary_statistics_builds():
    """Training builds (mp_f16_conv2d_fwd_stats): every weight-stationary shape has the forward-statistics build without a residual
    (what Chain16Fn launches); the builds left out because they do not fit the register file (conv_f16_ws.hip ws_build_fits) are
    exactly: backward statistics of 38 / 39 / 42 / 44, forward statistics + residual of 39 / 42 / 44, and the residual form of 44."""
    got = {}
    for stats, n_res in [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1)]:
        served = set()
        for g in (2, 5):
            for pr in fm.served_pairs(fm.WS_CASES, fm.WS_VARIANTS, fm.conv_case_desc, lambda c: n_res, stats, MP_F16_WS_GROUPS=g):
                served.add(pr.values[1])
        got[(stats, n_res)] = sorted(set(fm.WS_VARIANTS) - served)
    assert got == {(0, 0): [], (0, 1): [44], (1, 0): [], (1, 1): [39, 42, 44], (2, 0): [38, 39, 42, 44], (2, 1): [38, 39, 42, 44]}, got


def test_four_phase_data_gradient_is_not_offered_by_the_other_families():
    for fam in (fm.WREG_VARIANTS, fm.WS_VARIANTS):
        assert not fm.served_pairs(fm.PHASES4_CASES, fam, fm.phases4_case_desc)


def test_bench_plan_picks_are_still_served():
    """The tuner's choices for the fp16 conv launches of the bench plans (amp-O2 HRNet-W32 N = 128; config 5: HRNet-W48 384x288, 2N = 128;
    the amp-O2 training step N = 128 incl. statistics modes) as recorded on an MI355X: each (shape, variant, residuals, statistics mode)
    must still be accepted by the forced-variant entry.  A kernel that drops out of the candidate set would otherwise only make the
    bench slower under a green suite."""
    path = os.path.join(HERE, "golden", "bench_plan_picks.json")
    with open(path) as fh:
        picks = json.load(fh)
    assert len(picks["f16"]) >= 40
    families = set()
    for pk in picks["f16"]:
        d = fm._lib.ConvDesc(**pk["desc"])
        assert fm.supported(d, pk["variant"], pk["n_res"], pk["stats"]), f"no longer served: {pk}"
        v = pk["variant"]
        families.add("ws" if v in fm.WS_VARIANTS else "wreg" if v in fm.WREG_VARIANTS else "mt" if v in fm.MT_VARIANTS else "tile")
    assert {"ws", "wreg"} <= families, families  # the round-4 kernels carry the bench plans
