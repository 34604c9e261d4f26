import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests are skipped (not failed) when collected on a box without a GPU
    # and no marker expression was given.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


_DT = {"float16": torch.float16, "bfloat16": torch.bfloat16}


def _t16(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(dtype)


class Golden:
    """One committed fixture produced by tests/golden/make_golden.py from the
    reference's own Python code."""

    def __init__(self, path):
        z = np.load(path)
        self.name = os.path.basename(path)[:-4]
        self.kind = str(z["kind"])
        self.num_bits = int(z["num_bits"])
        self.tile_p = int(z["tile_p"])
        self.group_size = int(z["group_size"])
        self.dtype = _DT[str(z["dtype"])]
        self.Q = z["Q"]
        self.S = _t16(z["S"], self.dtype)
        self.table = _t16(z["table"], self.dtype)
        self.table2 = torch.from_numpy(z["table2"].view(np.int32)).view(torch.float32)
        self.D_identity = _t16(z["D_identity"], self.dtype)
        if self.kind == "kernel":
            self.W = z["W"]
            self.A = _t16(z["A"], self.dtype)
            self.What = _t16(z["What"], self.dtype)
            self.D = _t16(z["D"], self.dtype)
        else:
            self.vector_size = int(z["vector_size"])
            self.weight_higgs = torch.from_numpy(z["weight_higgs"])
            self.scales_higgs = _t16(z["scales_higgs"], self.dtype)
            self.grid = _t16(z["grid"], self.dtype)

    def __repr__(self):
        return self.name


def golden_paths():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


@pytest.fixture(params=golden_paths(), ids=lambda p: os.path.basename(p)[:-4])
def golden(request):
    return Golden(request.param)
