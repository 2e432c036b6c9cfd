import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.set_device(0)
    from tests import xcheck

    xcheck.load()  # the cross-check kernel generations (test build): `GpuModel.set_kernel(1 | 2)` works from here on
    return torch.device("cuda", 0)


def record_margin(test: str, **values) -> None:
    """Append the observed error statistics of a parity test to gpurun_out/test_margins.jsonl (scratch; read back in the build container to set / tighten the
    stated tolerances from measurements).  Never fails a test."""
    import json

    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "test_margins.jsonl"), "a") as f:
            f.write(json.dumps({"test": test, **{k: (float(v) if not isinstance(v, (str, int)) else v) for k, v in values.items()}}) + "\n")
    except Exception:
        pass
