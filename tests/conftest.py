import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # make sure the in-tree HIP library exists (hipcc cross-compiles without a GPU)
    from v1t_amd.build import build

    build(force=False, verbose=False)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    d = {}
    gdir = os.path.join(ROOT, "tests", "golden")
    for f in sorted(os.listdir(gdir)):
        if f.endswith(".npz"):
            with np.load(os.path.join(gdir, f)) as z:
                d.update({k: z[k] for k in z.files})
    return d


def pytest_terminal_summary(terminalreporter):
    """Achieved error / bound of every tolerance check that went through tests.helpers (the tests print their
    margins): worst first; also written to gpurun_out/parity_margins.json when that directory exists."""
    from tests.helpers import MARGINS

    if not MARGINS:
        return
    rows = sorted(MARGINS.items(), key=lambda kv: -(kv[1][0] / max(kv[1][1], 1e-300)))
    tr = terminalreporter
    tr.write_sep("-", f"parity margins: achieved error / bound ({len(rows)} checks, worst first)")
    for name, (err, bound) in rows[:60]:
        tr.write_line(f"{err / max(bound, 1e-300):7.3f}x  err {err:9.3e}  bound {bound:8.1e}  {name}")
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        import json

        try:
            with open(os.path.join(out, "parity_margins.json"), "w") as f:
                json.dump({k: {"err": e, "bound": b, "ratio": e / max(b, 1e-300)} for k, (e, b) in rows}, f, indent=0)
        except OSError:
            pass
