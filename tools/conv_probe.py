"""Per-workgroup phase breakdown of single conv layers (diagnostic library with MP_CONV_STAMPS=1).

    tools/build_stamps.sh && MINDPOSE_HIP_LIB=build/stamps/libmindpose_hip.so python tools/conv_probe.py

Prints, per layer: event-timed duration / TFLOP/s, the launch geometry the library chose, and the mean
cycles a workgroup's wave 0 spent in: prologue (tables + first chunk), issuing next-chunk loads, the MFMA
loop, writing the next chunk to LDS (incl. the wait for its global loads), the barrier, the epilogue.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindpose_amd import _lib  # noqa: E402
from mindpose_amd.models.layers import BatchNorm2d, Conv2d, Plan  # noqa: E402

LAYERS = [
    # name, n, cin, cout, k, s, h, w, residual
    ("b0 32@64x48", 128, 32, 32, 3, 1, 64, 48, True),
    ("b1 64@32x24", 128, 64, 64, 3, 1, 32, 24, True),
    ("b2 128@16x12", 128, 128, 128, 3, 1, 16, 12, True),
    ("b3 256@8x6", 128, 256, 256, 3, 1, 8, 6, True),
    ("s1 1x1 64->256", 128, 64, 256, 1, 1, 64, 48, True),
    ("s1 1x1 256->64", 128, 256, 64, 1, 1, 64, 48, False),
    ("s1 3x3 64@64x48", 128, 64, 64, 3, 1, 64, 48, False),
    ("fuse down 32->64 s2", 128, 32, 64, 3, 2, 64, 48, True),
]


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    dbg = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
    have_stamps = lib.mp_debug_set_stamp_buffer(dbg.data_ptr(), dbg.numel() * 8) == 0
    for name, n, cin, cout, k, s, h, w, res in LAYERS:
        conv = Conv2d(cin, cout, k, stride=s, padding=k // 2)
        torch.nn.init.normal_(conv.weight, std=(2.0 / (cin * k * k)) ** 0.5)
        bn = BatchNorm2d(cout)
        x = torch.randn(n, cin, h, w, device=dev)
        ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
        r = torch.randn(n, cout, ho, wo, device=dev) if res else None
        plan = Plan(dev)
        plan.conv(x, conv, bn, relu=True, res1=r)
        info = plan.entry_info(0)
        for _ in range(3):
            plan.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            plan.run()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        tf = 2.0 * info["macs"] / us / 1e6
        print(f"{name:22s} {us:8.1f} us {tf:7.2f} TF  wgs={info['workgroups']} lds={info['lds_bytes']} "
              f"ct={info['cout_tile']} pt={info['pixel_tile']} ck={info['cin_chunk']} G={info['images_per_tile']} "
              f"R={info['rows_per_tile']}")
        if have_stamps:
            dbg.zero_()
            plan.run()
            torch.cuda.synchronize()
            d = dbg[: info["workgroups"] * 8].reshape(-1, 8).cpu().numpy().astype(np.float64)
            tot = d[:, 0].mean()
            names = ["total", "prologue", "ld_issue", "mfma_loop", "st+wait", "barrier", "epilogue"]
            parts = "  ".join(f"{nm}={d[:, i].mean():9.0f} ({100 * d[:, i].mean() / tot:4.1f}%)" for i, nm in enumerate(names))
            start = d[:, 7] - d[:, 7].min()
            print(f"    cycles/WG: {parts}")
            print(f"    WG start spread: p50={np.percentile(start, 50):.0f} p90={np.percentile(start, 90):.0f} "
                  f"max={start.max():.0f} cycles; total min/max={d[:, 0].min():.0f}/{d[:, 0].max():.0f}")


if __name__ == "__main__":
    main()
