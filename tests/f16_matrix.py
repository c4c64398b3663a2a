"""The (shape, tile variant) matrix of the fp16 convolution tests, generated from the library's own answer.

`mp_f16_conv_supported` (include/mindpose_hip.h) runs the entry point's checks and the kernel family's dispatch without the launch -
host-only, so the matrix can be built at collection time and on a machine without a GPU.  A pair the library does not serve is never
collected (no skip), every collected pair must return MP_OK (a HIP error or a pair that stopped being served is a FAILURE), and
tests/test_f16_matrix_cpu.py holds the floors: every forced variant serves at least one of its family's cases and the pairs the
tuner picked for the three bench plans (tests/golden/bench_plan_picks.json) are still served.
"""
import contextlib
import ctypes
import os

import pytest

from mindpose_amd import _lib

# tile-variant families (csrc/conv_f16.h): one-tile 0-9 + 20, 21, 24; persistent multi-tile 10-19 + 22, 23; weights in registers
# 25-36 + 45-47; weight-stationary 37-44
TILE_VARIANTS = list(range(10)) + [20, 21, 24]
MT_VARIANTS = list(range(10, 20)) + [22, 23]
WREG_VARIANTS = list(range(25, 37)) + [45, 46, 47]
WREG_S2_VARIANTS = list(range(25, 37))
WS_VARIANTS = list(range(37, 45))

CONV_CASES = [
    # n, cin, cout, k, s, h, w, relu, n_res
    (2, 32, 32, 3, 1, 64, 48, True, 1),     # W32 branch 0
    (3, 64, 64, 3, 1, 32, 24, True, 1),     # branch 1 (two chunks with the 64-cout tile)
    (3, 128, 128, 3, 1, 16, 12, True, 0),
    (5, 256, 256, 3, 1, 8, 6, True, 2),     # multi-image tiles, many chunks
    (2, 3, 64, 3, 2, 64, 48, True, 0),      # stem conv1: 3 channels in one block, zero planes
    (2, 64, 64, 3, 2, 32, 24, True, 0),     # stem conv2 (stride 2)
    (2, 48, 48, 3, 1, 24, 16, True, 1),     # W48 widths: cin padded to 64, cout tile 48
    (2, 96, 192, 3, 2, 16, 12, False, 2),   # fuse-layer down-sampling conv with both residuals
    (2, 64, 256, 1, 1, 16, 12, True, 1),    # bottleneck 1x1
    (2, 256, 64, 1, 1, 16, 12, True, 0),
    (3, 32, 17, 1, 1, 64, 48, False, 0),    # head: 17 couts + bias
    (1, 40, 24, 3, 1, 9, 7, False, 0),      # ragged everything
]

WREG_CASES = [
    # n, cin, cout, k, s, h, w, relu, n_res  - stride-1 "same" convs with >= 64 input channels (conv_f16_wreg.hip)
    (3, 64, 64, 3, 1, 32, 24, True, 1),     # branch 1: row bands of 4, 8 bands per image
    (3, 128, 128, 3, 1, 16, 12, True, 1),   # branch 2: half-image bands (P6) / quarter bands (P3)
    (5, 256, 256, 3, 1, 8, 6, True, 1),     # branch 3: two images per tile, odd batch -> a half-empty last tile
    (2, 192, 192, 3, 1, 16, 12, True, 0),   # W48 branch 2: three cout tiles per wave
    (3, 96, 192, 3, 1, 12, 9, False, 1),    # ragged: 10-row bands of a 12-row map, 6 padding lanes per tile
    (2, 72, 64, 3, 1, 10, 7, True, 0),      # input channels padded 72 -> 96: three zero planes from the range check
    (2, 256, 64, 1, 1, 16, 12, True, 0),    # 1x1 (no halo column), 8 k-steps
    (2, 64, 256, 1, 1, 16, 12, True, 1),    # 1x1, four cout tiles per wave
    (130, 128, 128, 3, 1, 16, 12, True, 1), # more workgroups than CUs
    (3, 32, 32, 3, 1, 64, 48, True, 1),     # branch 0: pixel-split waves (W4), one k-step (no refill), 8-row bands
    (2, 48, 48, 3, 1, 24, 16, True, 1),     # W48 widths: three cout tiles, cin padded 48 -> 64
    (2, 96, 96, 3, 1, 24, 18, True, 0),     # W48 branch 1: W2 with three cout tiles per wave
    (3, 192, 192, 3, 1, 24, 18, True, 1),   # W48 branch 2 at config 5's map: P7C3 = 6-row bands, four per image
    (3, 384, 384, 3, 1, 12, 9, True, 1),    # W48 branch 3 at config 5's map: P4C3 = 7-row bands (7 + 5 rows), two cout slices
    (2, 256, 256, 3, 1, 10, 8, False, 1),   # P5C4: 256 couts per workgroup, 80-px bands
]

WS_CASES = [
    # n, cin, cout, k, s, h, w, relu, n_res - 3x3 stride-1 layers of the 32 ... 128-channel branches (W32 and W48 widths), ragged
    # bands (h not a multiple of the rows per tile), padded channel counts, long tile runs (MP_F16_WS_GROUPS)
    (5, 32, 32, 3, 1, 64, 48, True, 1),
    (3, 48, 48, 3, 1, 96, 72, True, 1),
    (3, 48, 48, 3, 1, 23, 72, True, 0),
    (6, 64, 64, 3, 1, 32, 24, True, 1),
    (5, 64, 64, 3, 1, 21, 17, False, 1),
    (6, 96, 96, 3, 1, 48, 36, True, 1),
    (4, 96, 96, 3, 1, 11, 36, True, 0),
    (9, 128, 128, 3, 1, 16, 12, True, 1),
    (3, 40, 48, 3, 1, 30, 33, True, 1),
    (3, 72, 96, 3, 1, 19, 20, False, 1),
    (2, 128, 64, 3, 1, 16, 12, True, 0),
    (4, 128, 128, 3, 1, 24, 16, True, 0),   # 128 couts x 96 px (variant 44): two-row bands
]

WREG_S2_CASES = [
    # n, cin, cout, h, w, n_res - 3x3 stride-2 pad-1 convs of the transition / exchange-unit layers (conv_f16_wreg.hip, S = 2)
    (3, 32, 64, 64, 48, 2),     # fuse down-sampling conv with running sum + identity
    (3, 64, 128, 32, 24, 1),
    (5, 128, 256, 16, 12, 0),   # two images per tile, odd batch
    (2, 32, 32, 64, 48, 0),
    (2, 64, 64, 30, 22, 2),     # ragged: odd output extents after the stride
    (2, 96, 192, 24, 18, 1),    # W48 widths
    (2, 256, 128, 17, 13, 0),   # odd input extents: the last tap column reads the shared zero slot (round 4's 256 -> 64 form of this
                                # case was served by NO variant of the family and had been skipping since it was written)
    (2, 48, 96, 48, 36, 1),     # W48 transition 48 -> 96: the three-cout-tile two-wave shape (variant 32)
    (2, 32, 96, 32, 24, 0),     # ... and the four-wave one (variant 35)
]

PHASES4_CASES = [(2, 32, 64, 32, 24), (3, 64, 128, 16, 12), (2, 48, 96, 24, 16), (5, 32, 32, 64, 48)]  # n, cin, cout, h, w
PHASES4_VARIANTS = [0, 4, 10, 13, 17]


def conv_desc(n, cin, cout, k, s, h, w, relu=0, flags=0):
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    return _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=k, kw=k, stride=s, pad_top=pad, pad_left=pad, conv_h=ho, conv_w=wo,
                         out_h=ho, out_w=wo, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=int(relu), flags=flags)


def phases4_desc(n, cin, cout, h, w):
    """The merged four-phase launch of a 3x3 stride-2 data gradient: input = the [n, cout, h/2, w/2] gradient, output [n, cin, h, w]."""
    ho, wo = h // 2, w // 2
    return _lib.ConvDesc(n=n, cin=cout, h=ho, w=wo, cout=cin, kh=2, kw=2, stride=1, pad_top=0, pad_left=0, conv_h=ho, conv_w=wo,
                         out_h=h, out_w=w, out_mul=2, out_rep=1, out_off_y=0, out_off_x=0, relu=0, flags=_lib.MP_CONV_PHASES4)


@contextlib.contextmanager
def knobs(**env):
    """MP_* experiment knobs for the duration of a query (tests set the same ones with monkeypatch for the launch)."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def supported(desc, variant, n_res=0, stats=0, **env) -> bool:
    with knobs(**env):
        return bool(_lib.load().mp_f16_conv_supported(ctypes.byref(desc), int(variant), int(n_res), int(stats)))


def served_pairs(cases, variants, desc_of, n_res_of=lambda c: 0, stats=0, **env):
    """[(case, variant)] the library serves, as pytest params with readable ids."""
    out = []
    for ci, case in enumerate(cases):
        d = desc_of(case)
        for v in variants:
            if supported(d, v, n_res_of(case), stats, **env):
                out.append(pytest.param(case, v, id=f"case{ci}-v{v}"))
    return out


def conv_case_desc(case):
    n, cin, cout, k, s, h, w, relu, _ = case
    return conv_desc(n, cin, cout, k, s, h, w, relu)


def conv_case_res(case):
    return case[8]


def s2_case_desc(case):
    n, cin, cout, h, w, _ = case
    return conv_desc(n, cin, cout, 3, 2, h, w, 1)


def s2_case_res(case):
    return case[5]


def phases4_case_desc(case):
    return phases4_desc(*case)
