import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the library honours its MP_* experiment knobs (forced tile runs / forms) only in processes that ask for them (csrc/common.h)
os.environ.setdefault("MINDPOSE_EXPERIMENT_KNOBS", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped (not failed) on a machine without a HIP device."""
    import torch
    reason = None
    if not torch.cuda.is_available():
        reason = "no HIP device (run on the GPU box: pytest -m gpu)"
    # (a GPU box WITHOUT the built library is not a reason to skip: the gpu tests then fail at `_lib.load()` - loudly, as they should)
    if reason:
        skip = pytest.mark.skip(reason=reason)
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
