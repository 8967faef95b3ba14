import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """GPU suite: the machine-readable parity report (tests/parity_log.py)."""
    try:
        import parity_log
        parity_log.dump(os.path.join(ROOT, "gpurun_out", "parity_report.json"))
    except Exception as e:   # the report must never turn a green run red
        print(f"[parity] report not written: {e!r}")
