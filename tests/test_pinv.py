"""P5: the 12 x 12 V S' U^T of calculateImageHessianInverse (src/PointCloudFactory.cu:1511-1824; cuSOLVER gesvd + cuBLAS
upstream).  Three independent computations must agree: the numpy fixture (tests/golden/pinv12.npz, made by
make_pinv_golden.py), the oracle's restatement (Jacobi eigen-decomposition of H^T H) and the product's host routine
(ssrlcv::pseudoInverse, one-sided Jacobi on H; reached through the C++ host-mirror test binary).  No GPU involved."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fixture():
    z = np.load(os.path.join(ROOT, "tests", "golden", "pinv12.npz"))
    return z["H"], z["pinv"], z["singular_values"]


def _rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def test_oracle_pinv_matches_numpy_fixture(oracle_lib):
    hs, want, sv = _fixture()
    assert len(hs) >= 7 and (sv < 1e-4).any() and (sv > 1e3).any()   # both sides of the cutoff are exercised
    for h, w in zip(hs, want):
        out = np.zeros((12, 12), np.float32)
        oracle_lib.oracle_pinv(H.P(np.ascontiguousarray(h)), ctypes.c_int(12), H.P(out))
        assert _rel(out.astype(np.float64), w) <= 2e-5, _rel(out.astype(np.float64), w)
    # the cutoff rule itself: a singular value below 1e-4 keeps its own value as the factor (:1698), it is not zeroed
    d = np.diag(np.array([2.0] * 11 + [5e-5], np.float32))
    out = np.zeros((12, 12), np.float32)
    oracle_lib.oracle_pinv(H.P(d), ctypes.c_int(12), H.P(out))
    assert np.allclose(np.diag(out)[:11], 0.5) and abs(out[11, 11] - 5e-5) < 1e-9


def test_host_mirror_pseudo_inverse_matches_oracle_and_fixture(oracle_lib, tmp_path):
    exe = os.path.join(ROOT, "ssrlcv_amd", "host", "_build", "host_mirror_test")
    if not os.path.exists(exe):
        pytest.skip("host mirror binary not built (python -c 'import __graft_entry__ as g; g.build()')")
    hs, want, _ = _fixture()
    rng = np.random.default_rng(3)
    extra = np.stack([(lambda a: a @ a.T)(rng.normal(size=(12, 12))) for _ in range(8)]).astype(np.float32)
    allh = np.concatenate([hs, extra])
    (tmp_path / "in.bin").write_bytes(allh.tobytes())
    subprocess.check_call([exe, "pinv", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    got = np.frombuffer((tmp_path / "out.bin").read_bytes(), np.float32).reshape(-1, 12, 12)
    assert len(got) == len(allh)
    for i, h in enumerate(allh):
        ref = np.zeros((12, 12), np.float32)
        oracle_lib.oracle_pinv(H.P(np.ascontiguousarray(h)), ctypes.c_int(12), H.P(ref))
        assert _rel(got[i].astype(np.float64), ref.astype(np.float64)) <= 2e-5, i
        if i < len(want):
            assert _rel(got[i].astype(np.float64), want[i]) <= 2e-5, i
