"""Generates tests/golden/pinv12.npz: 12 x 12 matrices and the V S' U^T that calculateImageHessianInverse defines
(src/PointCloudFactory.cu:1511-1824: SVD, S' = 1/S for S >= 1e-4 and S itself below, V S' U^T), computed here with
numpy.linalg.svd in float64.  The reference's cuSOLVER / cuBLAS are not available anywhere off an NVIDIA box and no
reference fixture reaches their output, so this numpy restatement of the published definition is the known answer the
oracle's independent restatement (oracle_pinv) is pinned to.  Run: python tests/golden/make_pinv_golden.py
"""
import os

import numpy as np


def ref_pinv(h):
    u, s, vt = np.linalg.svd(h.astype(np.float64))
    f = np.where(s >= 1e-4, 1.0 / np.where(s == 0, 1.0, s), s)
    return (vt.T * f) @ u.T, s


def main():
    rng = np.random.default_rng(12)
    mats = []
    for cond in (1e2, 1e5, 1e9):          # symmetric, BA-Hessian-like: eigenvalues spread over `cond`, some below 1e-4
        q, _ = np.linalg.qr(rng.normal(size=(12, 12)))
        ev = np.exp(rng.uniform(np.log(1e3 / cond), np.log(1e3), 12))
        mats.append((q * ev) @ q.T)
    mats.append(rng.normal(size=(12, 12)) * 50)                       # general, well conditioned
    q, _ = np.linalg.qr(rng.normal(size=(12, 12)))
    ev = np.array([900, 500, 120, 40, 9, 2, 0.5, 0.01, 3e-5, 2e-6, 0, 0], float)   # rank deficient + sub-cutoff values
    mats.append((q * ev) @ q.T)
    mats.append(np.diag(np.array([1e4, 1e3, 1e2, 10, 1, 0.1, 1e-2, 1e-3, 2e-4, 5e-5, 1e-6, 0.0])))
    mats.append(np.eye(12) * 3.0)
    hs = np.stack(mats).astype(np.float32)
    outs, svals = [], []
    for h in hs:
        p, s = ref_pinv(h)
        # keep clear of the cutoff: a singular value within 2 % of 1e-4 would make the answer depend on rounding
        assert not np.any(np.abs(s - 1e-4) < 2e-6), s
        outs.append(p)
        svals.append(s)
    np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pinv12.npz"), H=hs, pinv=np.stack(outs),
             singular_values=np.stack(svals))


if __name__ == "__main__":
    main()
