"""Pose refinement terms: oracle self-checks on CPU, HIP vs oracle on the GPU.
Reference: src/PoseEstimator.cu:349-393 (LM_iteration), :647-844 (kernels).  The reference has no fixture for this
path (parity unpinned); the oracle restates its formulas and the HIP kernel is held to the oracle."""
import numpy as np
import pytest

import helpers as H


def _fixture_matches(n=None):
    v = H.load_view("Pipeline2View")
    m = H.matches_from_matchset(v["kp0"])
    if n:
        m = m[:n]
    return v["cameras"], m


def test_oracle_residual_vanishes_for_consistent_rays(oracle_lib):
    """Two cameras looking at synthetic 3-D points: at the true relative pose the closest-point gap is ~0, and the cost
    grows when the relative rotation is perturbed."""
    rng = np.random.default_rng(5)
    cams = np.zeros(2, H.CAMERA)
    cams["foc"], cams["size"] = 0.16, 1024
    cams["dpix"] = 0.16 * np.tan(0.2) / 512
    cams["fov"] = 0.4
    true_pose = np.array([0.01, -0.02, 0.015, 0.3, 0.02, -0.01], np.float32)  # target relative to query
    pts = np.stack([rng.uniform(-1, 1, 400), rng.uniform(-1, 1, 400), rng.uniform(8, 12, 400)], 1)

    def rot(a):
        x, y, z = a
        cx, sx, cy, sy, cz, sz = np.cos(x), np.sin(x), np.cos(y), np.sin(y), np.cos(z), np.sin(z)
        return np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx],
                         [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx], [-sy, cy * sx, cy * cx]])
    R = rot(true_pose[:3].astype(np.float64))
    m = np.zeros(len(pts), H.MATCH)
    dp, foc = float(cams["dpix"][0][0]), 0.16
    m["kp0_loc"] = np.stack([pts[:, 0] / pts[:, 2] * foc / dp + 512, pts[:, 1] / pts[:, 2] * foc / dp + 512], 1)
    local = (pts - true_pose[3:].astype(np.float64)) @ R  # R^T (p - t)
    m["kp1_loc"] = np.stack([local[:, 0] / local[:, 2] * foc / dp + 512, local[:, 1] / local[:, 2] * foc / dp + 512], 1)
    c0 = H.oracle_pose_cost(oracle_lib, m, true_pose, cams[0:1], cams[1:2])
    bad = true_pose.copy()
    bad[1] += 0.01
    c1 = H.oracle_pose_cost(oracle_lib, m, bad, cams[0:1], cams[1:2])
    assert c0 < 1e-4 * c1 and c1 > 0
    # a Gauss-Newton step on the rotation block of the oracle's own terms must reduce the cost
    jtj, jtf, cost = H.oracle_pose_terms(oracle_lib, m, bad, cams[0:1], cams[1:2])
    assert abs(cost - c1) <= 1e-4 * c1
    assert np.all(jtj[3:, :] == 0) and np.all(jtj[:, 3:] == 0) and np.all(jtf[3:] == 0)  # position columns are 0
    step = -np.linalg.solve(jtj[:3, :3].astype(np.float64) + 1e-9 * np.eye(3), jtf[:3].astype(np.float64))
    new = bad.copy()
    new[:3] += step.astype(np.float32)
    c2 = H.oracle_pose_cost(oracle_lib, m, new, cams[0:1], cams[1:2])
    assert c2 < 0.05 * c1


def test_oracle_terms_are_symmetric_on_the_fixture(oracle_lib):
    cams, m = _fixture_matches(2000)
    pose = H.relative_pose(cams)
    jtj, jtf, cost = H.oracle_pose_terms(oracle_lib, m, pose, cams[0:1], cams[1:2])
    assert np.array_equal(jtj, jtj.T) and cost > 0 and np.isfinite(jtj).all() and np.isfinite(jtf).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 63, 257, 13534])
def test_pose_terms_match_oracle(oracle_lib, n):
    """HIP (wave-reduced float sums) vs oracle (sequential float sums): the Jacobian is a central difference of float
    residuals (delta 1e-5), so single entries carry ~1e-3 relative noise from libm differences; the sums over the
    matches must agree to 2e-3 of the largest entry, the cost to 1e-4."""
    from ssrlcv_amd import capi
    cams, m = _fixture_matches()
    m = m[:n]
    pose = H.relative_pose(cams)
    md = capi.to_dev(m) if n else capi.dev_bytes(40)
    jtj, jtf, cost = capi.pose_lm_terms(md, n, pose, cams[0:1], cams[1:2])
    rj, rf, rc = H.oracle_pose_terms(oracle_lib, m, pose, cams[0:1], cams[1:2])
    if n == 0:
        assert not jtj.any() and not jtf.any() and cost == 0
        return
    assert np.array_equal(jtj, jtj.T)
    assert np.all(jtj[3:, :] == 0) and np.all(jtf[3:] == 0)
    assert np.abs(jtj - rj).max() <= 2e-3 * np.abs(rj).max()
    assert np.abs(jtf - rf).max() <= 2e-3 * max(np.abs(rf).max(), np.sqrt(np.abs(rj).max() * rc))
    assert abs(cost - rc) <= 1e-4 * rc
    c = capi.pose_cost(md, n, pose, cams[0:1], cams[1:2])
    assert abs(c - rc) <= 1e-4 * rc
