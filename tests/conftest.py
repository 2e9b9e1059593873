"""pytest configuration: registers the `gpu` marker and shared helpers for the parity tests."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    """dense-batch fixtures"""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and not f.startswith("varlen_"))


def varlen_golden_names():
    """packed variable-length fixtures (reference varlen kernels)"""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("varlen_"))


def load_golden(name):
    """-> (params dict, arrays dict).  `o` is decoded to fp32 from the stored fp16/bf16 bit patterns."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    p = json.loads(str(z["params"]))
    arr = {k: z[k] for k in z.files if k not in ("params",)}
    bits = arr["o_bits"]
    if p["dtype"] == "fp16":
        arr["o"] = bits.view(np.float16).astype(np.float32)
    else:
        arr["o"] = (bits.astype(np.uint32) << 16).view(np.float32)
    arr["input_sha256"] = str(z["input_sha256"])
    return p, arr


def golden_inputs(orc, p):
    """The seeded inputs a dense fixture was made from (tests/golden/make_golden.py::run_case): distribution `dist`, queries
    optionally multiplied by `q_mul` (peaky scores) and rounded to the storage dtype again."""
    q, k, v = orc.make_inputs(p["B"], p["H"], p["S"], p["D"], seed=p["seed"], layout=p["layout"], dtype=p["dtype"],
                              Hkv=p["Hkv"], Sk=p["Sk"], k_bias=p["k_bias"], dist=p.get("dist", "normal"))
    if p.get("q_mul", 1.0) != 1.0:
        q = orc.to_storage(q * np.float32(p["q_mul"]), p["dtype"])
    return q, k, v


@pytest.fixture(scope="session")
def oracle():
    from oracle import lowbit_fa_oracle
    return lowbit_fa_oracle
