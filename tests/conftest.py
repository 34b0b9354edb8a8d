import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    # device_count() does not initialise the GPU on this image
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def knob():
    """sets measurement switches of the kernel launchers for one test (fx_set_knob: which kernel of the library serves a geometry);
    usage: knob("ADVECT_LDS", "0"); every switch touched goes back to its default behind the test"""
    from fluidx12_amd import capi
    touched = []

    def setter(name, value):
        if name not in capi.knob_names():
            # the shipped library offers ten switches; the launchers' A/B switches (and the superseded kernels behind some of them) exist in
            # lab builds only: FLUIDX_BUILD_LAB=1 python -m fluidx12_amd.build, then the whole suite runs (profiles/r10_pytest_gpu_lab.txt)
            pytest.skip("fx_set_knob(%s): a lab-build switch (-DFX_LAB)" % name)
        touched.append(name)
        capi.set_knob(name, value)

    yield setter
    for name in touched:
        capi.set_knob(name, None)
