import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def graft():
    import __graft_entry__ as g
    return g


@pytest.fixture(scope="session")
def pkg(graft):
    return graft.load_package()


@pytest.fixture(scope="session")
def O():
    import oracle
    oracle.port()        # builds libgc_oracle.so on demand
    return oracle


@pytest.fixture(scope="session")
def emu_lib_path(graft):
    """The product's HIP sources compiled against the SIMT emulator (tests/emu) -- CPU bring-up only."""
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "emu"), "-j8"], check=True, capture_output=True)
    return os.path.join(ROOT, "tests", "emu", "_build", "libgpucodec_emu.so")


@pytest.fixture(scope="session")
def emu_enc(pkg, emu_lib_path):
    enc = pkg.ZstdEncoder(lib_path=emu_lib_path)
    yield enc
    enc.close()


@pytest.fixture(scope="session")
def gpu_enc(pkg, graft):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    enc = pkg.ZstdEncoder(device=0)       # raises loudly if libgpucodec.so is missing or no gfx950 device opens
    yield enc
    enc.close()
