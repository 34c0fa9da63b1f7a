"""The Delta filter on the device against the reference's C/Delta.c: every kind of delta, sizes around the delta and around the
64 KiB chunk of the decoder, the state carried from one buffer to the next, encode -> decode = identity."""
import numpy as np
import pytest

DELTAS = [1, 2, 3, 4, 7, 16, 100, 255, 256]


def _run(pkg, lib_path, x, delta, enc, state, dev=False):
    out = np.empty(max(1, x.size), dtype=np.uint8)
    if not dev:
        st = pkg.delta_convert_device(x.ctypes.data if x.size else 0, out.ctypes.data if x.size else 0, x.size, delta, enc, state, lib_path)
        return out[: x.size], st
    import torch
    d_in = torch.from_numpy(x).cuda() if x.size else torch.empty(1, dtype=torch.uint8, device="cuda")
    d_out = torch.empty(max(1, x.size), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    st = pkg.delta_convert_device(d_in.data_ptr(), d_out.data_ptr(), x.size, delta, enc, state)
    return d_out[: x.size].cpu().numpy(), st


@pytest.mark.parametrize("delta", DELTAS)
def test_emu_delta_matches_the_reference(pkg, O, emu_lib_path, delta):
    if O.ref("bra") is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(delta)
    for n in (0, 1, max(1, delta - 1), delta, delta + 1, 65535, 65536, 65537, 200_003):
        x = rng.integers(0, 256, size=n, dtype=np.uint8)
        state = bytes(rng.integers(0, 256, size=256, dtype=np.uint8)) if n % 2 else None
        for enc in (True, False):
            want, wst = O.ref_delta_convert(x, delta, enc, state)
            got, gst = _run(pkg, emu_lib_path, x, delta, enc, state)
            assert np.array_equal(got, want), (delta, n, enc, np.nonzero(got != want)[0][:8])
            assert gst[:delta] == wst[:delta], (delta, n, enc)
    # two buffers in a row, the state carried over, equal one pass over their concatenation; and the way back
    a, b = rng.integers(0, 256, size=70_001, dtype=np.uint8), rng.integers(0, 256, size=33, dtype=np.uint8)
    ya, s1 = _run(pkg, emu_lib_path, a, delta, True, None)
    yb, s2 = _run(pkg, emu_lib_path, b, delta, True, s1)
    whole, _ = O.ref_delta_convert(np.concatenate([a, b]), delta, True, None)
    assert np.array_equal(np.concatenate([ya, yb]), whole)
    za, t1 = _run(pkg, emu_lib_path, ya, delta, False, None)
    zb, _ = _run(pkg, emu_lib_path, yb, delta, False, t1)
    assert np.array_equal(za, a) and np.array_equal(zb, b)


@pytest.mark.gpu
def test_gpu_delta_matches_the_reference(pkg, O, graft):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    rng = np.random.default_rng(1)
    x = rng.integers(0, 256, size=100_000_003, dtype=np.uint8)
    for delta in (1, 2, 3, 4, 100, 256):
        for enc in (True, False):
            want, wst = O.ref_delta_convert(x, delta, enc, None)
            got, gst = _run(pkg, None, x, delta, enc, None, dev=True)
            assert np.array_equal(got, want), (delta, enc)
            assert gst[:delta] == wst[:delta]
