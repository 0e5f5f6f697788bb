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
