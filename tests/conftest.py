import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def fs2_option():
    """Set kernel-choice switches of libfs2_hip for one test (fs2_set_option); restored to automatic afterwards.
    (The library reads the FS2_* environment variables only once, so tests cannot switch them with setenv.)"""
    from fastspeech2_amd import _lib
    touched = []

    def set_(name, value):
        _lib.set_option(name, int(value))
        touched.append(name)

    yield set_
    for name in touched:
        _lib.set_option(name, 0 if name in ("FS2_NOSPLITK", "FS2_F32_ROWS", "FS2_OP_ATT_PLANES") else (1 if name == "FS2_FUSE_VAR" else -1))


def record_measurement(name, value):
    """Append a measured error to gpurun_out/measured_errors.jsonl (merged back from the GPU box): the numbers the test
    tolerances are derived from (<= 5x measured)."""
    import json
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "measured_errors.jsonl"), "a") as f:
            f.write(json.dumps({"name": name, "value": float(value)}) + "\n")
    except OSError:
        pass
