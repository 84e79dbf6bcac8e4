"""Shared test plumbing.

Markers: `gpu` = needs a real MI355X (run by the driver with `-m gpu`); everything
else runs on CPU.  The package directory is `ocean-bgc_amd/` (hyphenated, as the
project layout requires), so it is imported under the name `ocean_bgc_amd`.
"""
import importlib.util
import os
import sys

import numpy as np
import pytest

# The library takes the separate kernels for small native-layout batches (below 700 000 cells) and the column-block kernel
# above; most tests here are small and are about the column-block kernel, so the suite lowers the threshold to zero.
# tests/test_gpu_column_kernel.py::test_small_batches_take_the_separate_kernels_by_default covers the default.
os.environ.setdefault("OBGC_COLUMN_MIN_CELLS", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_pkg():
    if "ocean_bgc_amd" in sys.modules:
        return sys.modules["ocean_bgc_amd"]
    spec = importlib.util.spec_from_file_location(
        "ocean_bgc_amd", os.path.join(ROOT, "ocean-bgc_amd", "__init__.py"),
        submodule_search_locations=[os.path.join(ROOT, "ocean-bgc_amd")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ocean_bgc_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU")


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def synth(pkg):
    from ocean_bgc_amd import synth as s
    return s


@pytest.fixture(scope="session")
def oracle():
    from oracle import orabind
    return orabind.Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle import refbind
    if not refbind.available():
        pytest.skip("oracle/_ref/libbgc_ref.so not built (needs /root/reference)")
    return refbind.Ref()


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# ---- parity metrics ------------------------------------------
def e_norm(x, xref):
    m = np.abs(xref).max()
    d = np.abs(np.asarray(x) - np.asarray(xref)).max()
    return d / m if m > 0 else d


def e_pt(x, xref, floor=1e-6):
    xref = np.asarray(xref)
    m = np.abs(xref).max()
    if m == 0:
        return 0.0
    sel = np.abs(xref) > floor * m
    if not sel.any():
        return 0.0
    return (np.abs(np.asarray(x)[sel] - xref[sel]) / np.abs(xref[sel])).max()
