import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _no_gpu_here():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:
        return True


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """CPU runs (no GPU in the machine) spread the tests over a few worker processes (pytest-xdist, if installed): the SIMT
    emulator is slow and single-threaded.  GPU runs stay in one process (one device, and the driver records which libraries that
    process loads).  GC_TEST_WORKERS overrides (0 = off)."""
    want = os.environ.get("GC_TEST_WORKERS")
    n = int(want) if want is not None else (max(2, min(6, (os.cpu_count() or 4) - 2)) if _no_gpu_here() else 0)
    if os.environ.get("PYTEST_XDIST_WORKER") or hasattr(config, "workerinput"):
        return                                     # inside a worker: never nest
    if n > 0 and getattr(config.option, "numprocesses", None) is None and config.pluginmanager.hasplugin("xdist"):
        config.option.numprocesses = n
        config.option.dist = "load"               # (per test: the slowest files would otherwise be the critical path; module fixtures are cheap)
        config.option.tx = ["popen"] * n          # (what -n would have filled in)


class _BuildLock:
    """Serialises on-demand builds of the checkers between xdist workers."""
    def __enter__(self):
        import fcntl
        self.f = open(os.path.join(ROOT, "tests", ".build.lock"), "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
    def __exit__(self, *a):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN); self.f.close()


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
    with _BuildLock():
        oracle.port()        # builds libgc_oracle.so on demand
    return oracle


@pytest.fixture(scope="session")
def emu_lib_path(graft):
    """The product's HIP sources compiled against the SIMT emulator (tests/emu) -- CPU bring-up only."""
    import subprocess
    with _BuildLock():
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


@pytest.fixture(scope="session")
def gpu_hooks_kw(pkg, graft):
    """Constructor arguments for GPU tests that steer the library through GC_* environment hooks: those exist only in the test build
    (csrc/libgpucodec_hooks.so, the same objects with gc_api.hip compiled -DGC_TEST_HOOKS); the shipped library reads no environment."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    return dict(device=0, lib_path=pkg.HOOKS_LIB_PATH)
