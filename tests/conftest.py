"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
if os.path.join(ROOT, "oracle") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def pytest_sessionfinish(session, exitstatus):
    """Measured parity errors of a `-m gpu` session -> gpurun_out/parity_measured.json (tests/parity.py)."""
    try:
        import json

        import parity

        if parity.MEASURED:
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "parity_measured.json"), "w") as f:
                json.dump(dict(sorted(parity.MEASURED.items())), f, indent=1)
    except Exception as e:  # never turn a green run red because the record could not be written
        print("parity record not written:", e)
