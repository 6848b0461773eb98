import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """Build the product library and the oracle once per session."""
    lib = ROOT / "path_tracer_c_amd" / "lib" / "libptgpu.so"
    ora = ROOT / "oracle" / "liboracle.so"
    exe = ROOT / "bin" / "ptrender"
    if not lib.exists() or not ora.exists() or not exe.exists():
        import __graft_entry__
        __graft_entry__.build()
    return True


@pytest.fixture(scope="session")
def pt(built):
    import path_tracer_c_amd
    path_tracer_c_amd.lib()
    return path_tracer_c_amd


@pytest.fixture(scope="session")
def oracle(built):
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def gpu(pt):
    """The C-ABI on device 0; fails (not skips) if the library cannot see it."""
    n = pt.lib().ptgpu_device_count()
    assert n > 0, "GPU test but no HIP device: " + pt.lib().ptgpu_last_error().decode()
    pt._abi.check(pt.lib().ptgpu_set_device(0))
    return pt
