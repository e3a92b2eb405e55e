import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built library (the .so is git-ignored): build it once, the same way
    __graft_entry__.build() does.  A failed build is left to the tests to report (the package
    raises ImportError -- there is no CPU fallback)."""
    so = os.path.join(ROOT, "introtocomputervision_amd", "libmicv.so")
    if not os.path.exists(so):
        import shutil
        import subprocess
        if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
            subprocess.run(["bash", os.path.join(ROOT, "introtocomputervision_amd", "csrc", "build.sh")],
                           check=False)


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def ctx():
    from introtocomputervision_amd._capi import Context
    c = Context(0)
    yield c
    c.close()
