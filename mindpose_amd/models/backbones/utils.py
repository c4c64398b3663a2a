"""Pretrained-weight loading (reference: mindpose/models/backbones/utils.py:10-42).

The reference downloads a MindSpore ``.ckpt``; there is no network here and the ``.ckpt`` reader is a
"next" row.  A local ``.pt`` / ``.npz`` state dict with the reference's parameter
names can be loaded; anything else raises instead of silently skipping.
"""
import os

import numpy as np
import torch


def load_pretrained(net: torch.nn.Module, ckpt_url: str = "") -> None:
    if not ckpt_url or not os.path.exists(ckpt_url):
        raise FileNotFoundError(
            f"pretrained weights `{ckpt_url}` not found: downloading is unavailable (no network); pass a local "
            ".pt/.npz state dict that uses the reference's parameter names")
    if ckpt_url.endswith(".npz"):
        state = {k: torch.from_numpy(v) for k, v in np.load(ckpt_url).items()}
    else:
        state = torch.load(ckpt_url, map_location="cpu")
    net.load_state_dict(state, strict=False)  # strict_load=False in the reference (utils.py:40)
