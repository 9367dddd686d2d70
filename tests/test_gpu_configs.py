"""BASELINE.json configs [3] and [4] on one GPU, plus the pieces around them that only run with a device:

  config[3]  4-view 4096 x 4096: N-view flow (ssrlcv_amd.pipeline at world size 1) on tools/scene.py views -- determinism,
             every pair's uint2_pair list equal to the stand-alone matcher call, BA error sweep against the triangulator's
             own error sum, cloud against the generator's ground truth; the same flow at 4096^2 in TWO processes on the one
             GPU over gloo, and the 1024^2 fixtures in two and four, all equal to the single-process result;
  config[4]  8-view 8192 x 8192 pushbroom strips through the whole flow (generatePushbroomBundle -> N-view triangulate)
             on one GPU against the pushbroom generator's ground truth and for determinism; SIFT properties at that size;
             the 3-strip 2048^2 flow;
  fp16 matcher (SSRLCV_MATCH_F16=1, the formulation the north star names) in a subprocess against the oracle;
  colour input (convertToBW) and the key-point capacity flag.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _ground_truth_error(rig, scene_obj, mm, kp, pts):
    """|triangulated point - generator's ground truth at the bundle's first key point|, km."""
    first = kp[mm["index"]]
    err = np.zeros(len(mm))
    for v in np.unique(first["parentId"]):
        sel = np.nonzero(first["parentId"] == v)[0]
        xs = torch.from_numpy(first["loc"][sel, 0].copy()).to(scene_obj.device)
        ys = torch.from_numpy(first["loc"][sel, 1].copy()).to(scene_obj.device)
        gt, _, _ = rig.ground_points(scene_obj, int(v), xs, ys)
        err[sel] = np.linalg.norm(pts[sel] - gt.cpu().numpy(), axis=1)
    return err


def test_config3_four_view_4096_flow(capi):
    import scene
    from ssrlcv_amd import pipeline, dist as sd
    S, V = 4096, 4
    imgs, cams, rig, sc = scene.pinhole_views(V, S)
    seed, _ = H.load_seed_features()
    ws = pipeline.Workspace()
    res = pipeline.reconstruct(imgs, cams, seed_features=seed, mode=1, ws=ws, ba=True)
    nfeat = [f.numel() // 152 for f in res["features"]]
    assert all(2e5 < n < 2e6 for n in nfeat), nfeat       # ~0.03 features per pixel, like the everest fixtures
    mm, kp, pts = res["matches"], res["keypoints"], res["points"].cpu().numpy()
    assert len(mm) > 50000 and pts.shape == (len(mm), 3) and np.isfinite(pts).all()
    assert set(np.unique(mm["numKeyPoints"])) <= {2, 3, 4} and (mm["numKeyPoints"] > 2).sum() > 1000
    # determinism: a second pass through the same workspace reproduces every array
    res2 = pipeline.reconstruct(imgs, cams, seed_features=seed, mode=1, ws=ws, ba=True)
    assert all(torch.equal(a, b) for a, b in zip(res["features"], res2["features"]))
    assert all(torch.equal(a, b) for a, b in zip(res["pairs"], res2["pairs"]))
    assert np.array_equal(mm, res2["matches"]) and np.array_equal(kp, res2["keypoints"])
    assert np.array_equal(pts, res2["points"].cpu().numpy())
    # every pair's list equals the stand-alone 2-view matcher call on the same features
    seed_d = capi.to_dev(seed)
    for p, (qi, ti) in enumerate(sd.pair_list(V)):
        fq, ft = res["features"][qi], res["features"][ti]
        nq, nt = nfeat[qi], nfeat[ti]
        sdist = capi.seed_distances(fq, nq, seed_d, len(seed))
        params = capi.make_match_params(1, qi, ti, 25.0, 5.0, 0.6, 200.0 * 200.0, cams[qi:qi + 1],
                                        capi.projection_matrix(cams[ti:ti + 1]))
        out = capi.match(fq, nq, ft, nt, params, capi.OUT_UINT2_PAIR, seed_d=sdist)
        n = capi.compact_matches(capi.OUT_UINT2_PAIR, out, nq, capi.match_workspace(nq, nt))
        assert torch.equal(out[: n * 16], res["pairs"][p]), (p, n, res["pairs"][p].numel() // 16)
    # BA error sweep on pair (0, 1): 612 sums; every gradient-stencil point (h = 1e-5) sits within a few per cent of the
    # unperturbed value, which is the triangulator's own sum of squared gaps over the same bundles
    sums = res["ba_sums"].cpu().numpy()
    assert sums.shape == (612,) and np.isfinite(sums).all() and res["ba_bundles"] > 10000
    two = np.nonzero(mm["numKeyPoints"] == 2)[0]
    two = two[(kp["parentId"][mm["index"][two]] == 0) & (kp["parentId"][mm["index"][two] + 1] == 1)]
    assert res["ba_bundles"] == len(two)        # the flow selects the pair's bundles on the device (round 4)
    sub_mm = np.zeros(len(two), H.MULTIMATCH)
    sub_mm["numKeyPoints"], sub_mm["index"] = 2, 2 * np.arange(len(two))
    sub_kp = np.zeros(2 * len(two), H.KEYPOINT)
    sub_kp["parentId"][1::2] = 1
    sub_kp["loc"][0::2], sub_kp["loc"][1::2] = kp["loc"][mm["index"][two]], kp["loc"][mm["index"][two] + 1]
    b_d, l_d = capi.generate_bundles(capi.to_dev(sub_mm), capi.to_dev(sub_kp), len(two), capi.to_dev(cams[:2].copy()), 2, len(sub_kp))
    _, err_d, _ = capi.triangulate(l_d, b_d, len(two), want_errors=True)
    ref = float(err_d.double().sum().item())
    centre = sums[24 + 2]   # the unperturbed point of the first diagonal stencil
    assert abs(centre - ref) <= 2e-3 * ref, (centre, ref)
    assert np.abs(sums[:24] - centre).max() <= 0.2 * centre
    # the cloud against the generator's ground truth: GSD 4.1 m, baselines 35..140 km at 400 km range
    err = _ground_truth_error(rig, sc, mm, kp, pts)
    print("config[3]: %d features/image, %d multi-matches, cloud error vs ground truth: median %.4f km, 90 %% %.4f km" %
          (int(np.mean(nfeat)), len(mm), np.median(err), np.percentile(err, 90)))
    assert np.median(err) < 0.03 and np.percentile(err, 90) < 0.15


def test_scene_pair_1024_hip_equals_oracle_end_to_end(capi, oracle_lib):
    """A 1024^2 two-view scene of the benchmark's generator through both implementations: HIP features bit-equal to the
    oracle's, therefore the same seed distances, the same double-constrained match list and the same cloud -- and that
    cloud sits on the generator's ground truth."""
    import scene
    S = 1024
    imgs, cams, rig, sc = scene.pinhole_views(2, S)
    seed, _ = H.load_seed_features()
    plans = [capi.SiftPlan(S, S) for _ in range(2)]
    feats = []
    for p, im in zip(plans, imgs):
        p.extract(im)
        feats.append(p.features_host(H.FEATURE))
    ofeats = [H.oracle_sift(oracle_lib, im.cpu().numpy()) for im in imgs]
    for g, o in zip(feats, ofeats):
        assert len(g) > 20000
        H.assert_features_equal(g, o)
    n0, n1 = len(feats[0]), len(feats[1])
    sd_d = capi.seed_distances(plans[0].features, n0, capi.to_dev(seed), len(seed))
    osd = H.oracle_seed_distances(oracle_lib, ofeats[0], seed)
    assert np.array_equal(sd_d.cpu().numpy()[:n0], osd)
    params = capi.make_match_params(1, 0, 1, 25.0, 5.0, 0.6, 200.0 * 200.0, cams[0:1], capi.projection_matrix(cams[1:2]))
    out_d = capi.match(plans[0].features, n0, plans[1].features, n1, params, capi.OUT_DMATCH, seed_d=sd_d)
    n = capi.compact_matches(capi.OUT_DMATCH, out_d, n0, capi.match_workspace(n0, n1))
    dm = capi.to_host(out_d, H.DMATCH, n)
    proj = H.oracle_projection(oracle_lib, cams[1:2])
    odm = H.oracle_match_dmatch(oracle_lib, 1, 0, ofeats[0], 1, ofeats[1], cams[0:1], proj, 25.0, 5.0, osd, 0.6, 200.0 * 200.0)
    odm = odm[odm["invalid"] == 0]
    assert n == len(odm) > 3000
    for name in ("kp0_loc", "kp1_loc", "distance"):
        assert np.array_equal(dm[name], odm[name]), name
    mm = np.zeros(n, H.MULTIMATCH)
    mm["numKeyPoints"], mm["index"] = 2, 2 * np.arange(n)
    kp = np.zeros(2 * n, H.KEYPOINT)
    kp["parentId"][1::2] = 1
    kp["loc"][0::2], kp["loc"][1::2] = dm["kp0_loc"], dm["kp1_loc"]
    b_d, l_d = capi.generate_bundles(capi.to_dev(mm), capi.to_dev(kp), n, capi.to_dev(cams), 2, len(kp))
    pts = capi.triangulate(l_d, b_d, n)[0].cpu().numpy().reshape(-1, 3)
    ob, ol, _ = H.oracle_bundles(oracle_lib, mm, kp, cams)
    opts, _, _ = H.oracle_triangulate(oracle_lib, False, ob, ol)
    assert np.array_equal(pts, opts)            # bit-equal clouds
    err = _ground_truth_error(rig, sc, mm, kp, pts)
    print("scene 1024^2 pair: %d features, %d matches, cloud vs ground truth: median %.4f km (GSD %.4f km)" %
          (n0, n, np.median(err), rig.gsd))
    assert np.median(err) < 3 * rig.gsd and np.percentile(err, 90) < 0.3


def _gloo_worker(rank, world, port, out_dir, workload):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)   # all ranks on the one GPU
        from ssrlcv_amd import capi, pipeline
        seed, _ = H.load_seed_features()
        if workload == "everest":
            pix = [capi.to_dev(p).view(1024, 1024) for p in H.load_everest_pixels()]
            cams = H.load_view("Pipeline3View")["cameras"]
        else:  # config[3]: the benchmark's 4-view 4096^2 scene (every rank renders it: the generator is deterministic)
            import scene
            pix, cams, _, _ = scene.pinhole_views(4, 4096)
        res = pipeline.reconstruct(pix, cams, seed_features=seed, ba=True)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), mm=res["matches"], kp=res["keypoints"],
                 pts=res["points"].cpu().numpy(), ba=res["ba_sums"].cpu().numpy())
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _spawn_ranks(world, out_dir, workload):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_gloo_worker, args=(world, port, str(out_dir), workload), nprocs=world, join=True)
    return [np.load(out_dir / ("rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_on_one_gpu_equal_single_process(capi, tmp_path, world):
    """The sharded flow with world size 2 and 4 (all ranks on device 0, gloo collectives staged through the host) against
    world size 1 and the reference's 3-view fixture: same MatchSet on every rank, same cloud; BA sums to float
    accumulation accuracy (the all-reduce adds the ranks' partial sums).  With 4 ranks and 3 views / 3 pairs the last rank
    owns no image and no pair -- what ranks 4..7 see when the 4-view leg of bench.py runs on 8 GPUs."""
    from ssrlcv_amd import pipeline
    ranks = _spawn_ranks(world, tmp_path, "everest")
    r0 = ranks[0]
    pix = [capi.to_dev(p).view(1024, 1024) for p in H.load_everest_pixels()]
    seed, _ = H.load_seed_features()
    v = H.load_view("Pipeline3View")
    one = pipeline.reconstruct(pix, v["cameras"], seed_features=seed, ba=True)
    for r in ranks:
        assert np.array_equal(r["mm"], one["matches"]) and np.array_equal(r["kp"], one["keypoints"])
        assert np.array_equal(r["pts"], one["points"].cpu().numpy())
        assert np.allclose(r["ba"], one["ba_sums"].cpu().numpy(), rtol=1e-4)
    assert len(r0["mm"]) == 21177 and np.array_equal(r0["kp"]["loc"], v["kp0"]["loc"])


def test_config3_two_ranks_on_one_gpu_at_4096(capi, tmp_path):
    """config[3] at its size in the sharded form: the 4-view 4096^2 flow with TWO ranks (both on device 0, gloo) -- two
    images, three of the six pairs (balanced by nq * nt) and half of the bundles each -- gives every rank the MatchSet and
    cloud of the single-process run, entry for entry."""
    import scene
    from ssrlcv_amd import pipeline
    ranks = _spawn_ranks(2, tmp_path, "scene4096")
    imgs, cams, _, _ = scene.pinhole_views(4, 4096)
    seed, _ = H.load_seed_features()
    one = pipeline.reconstruct(imgs, cams, seed_features=seed, ba=True)
    assert len(one["matches"]) > 50000
    for r in ranks:
        assert np.array_equal(r["mm"], one["matches"]) and np.array_equal(r["kp"], one["keypoints"])
        assert np.array_equal(r["pts"], one["points"].cpu().numpy())
        assert np.allclose(r["ba"], one["ba_sums"].cpu().numpy(), rtol=1e-4)


def test_config4_sift_8192_properties(capi):
    import scene
    S = 8192
    rig = scene.PushbroomRig(2, S)
    sc = scene.Scene(S, rig.gsd_x, anisotropy=rig.gsd_y / rig.gsd_x)
    img = rig.render(sc, 0)
    plan = capi.SiftPlan(S, S)
    plan.extract(img)
    n1 = plan.count()
    f1 = plan.features[: n1 * 152].clone()
    plan.extract(img)
    assert plan.count() == n1 and torch.equal(f1, plan.features[: n1 * 152])
    f = capi.to_host(f1, H.FEATURE, n1)
    assert 5e5 < n1 < 8e6, n1
    assert (f["parent"] == -1).all() and (f["sigma"] > 0).all()
    assert (f["theta"] >= 0).all() and (f["theta"] < 2 * np.pi + 1e-6).all()
    assert (f["loc"] >= 0).all() and (f["loc"] < S).all()
    norms = np.sqrt((f["values"].astype(np.float64) ** 2).sum(1))
    assert np.percentile(norms, 0.1) > 245 and norms.max() < 265
    # octave-major, blur-major; inside a blur segment the stable sort after refinement leaves up to three raster-ordered
    # runs (key points that arrived from the blur below, stayed, arrived from the blur above): y falls back to the top of
    # the image at most 4 x 3 x 3 - 1 times
    dy = np.diff(f["loc"][:, 1])
    assert (dy < -100.0).sum() <= 35


def test_config4_pushbroom_flow_against_ground_truth(capi):
    """Three pushbroom strips (2048^2 here: the flow, not the size, is what this covers) through SIFT, brute-force
    matching with the seed ratio test, the host merge, generatePushbroomBundle and N-view triangulation; the cloud
    against the pushbroom generator's ground truth."""
    import scene
    from ssrlcv_amd import pipeline
    S, V = 2048, 3
    imgs, pbs, rig, sc = scene.pushbroom_views(V, S)
    seed, _ = H.load_seed_features()
    res = pipeline.reconstruct(imgs, None, seed_features=seed, mode=0, pushbroom=pbs)
    mm, kp, pts = res["matches"], res["keypoints"], res["points"].cpu().numpy()
    assert len(mm) > 2000 and np.isfinite(pts).all()
    err = _ground_truth_error(rig, sc, mm, kp, pts)
    good = float((err < 0.2).mean())
    print("config[4] flow: %d multi-matches, cloud error vs ground truth: median %.4f km, %.1f %% within 0.2 km" %
          (len(mm), np.median(err), 100 * good))
    # rolls 2..16 degrees: a 0.25 rad base angle at 400 km; a one-pixel (8 m across track) mismatch is ~0.03 km.  The
    # matcher has no geometric constraint for pushbroom cameras (mode 0 + seed ratio test), so a tail of wrong matches
    # lands kilometres away -- upstream removes it with the statistical filters after triangulation
    assert np.median(err) < 0.05 and good > 0.7
    # ... and does here, on the device, inside the flow: twelve passes of the 3 sigma / 10 % statistical filter
    res_f = pipeline.reconstruct(imgs, None, seed_features=seed, mode=0, pushbroom=pbs, filters=[("statistical", 3.0, 0.1)] * 12)
    fmm, fkp, fpts = res_f["matches"], res_f["keypoints"], res_f["points"].cpu().numpy()
    ferr = _ground_truth_error(rig, sc, fmm, fkp, fpts)
    print("config[4] flow, filtered: %d of %d multi-matches kept, %.1f %% within 0.2 km" % (len(fmm), res_f["matches_unfiltered"], 100 * (ferr < 0.2).mean()))
    assert res_f["matches_unfiltered"] == len(mm) and len(fmm) < len(mm)
    assert (ferr < 0.2).mean() > 0.9 and (ferr < 0.2).sum() > 0.97 * (err < 0.2).sum()


def test_config4_eight_view_8192_pushbroom_flow(capi):
    """BASELINE config[4] at its size on ONE GPU: eight 8192 x 8192 pushbroom strips through ssrlcv_amd.pipeline -- SIFT
    (one plan reused: eight workspaces of 22 GB would not be sensible), the 28 image pairs (brute force + seed ratio
    test), the merge, generatePushbroomBundle (src/PointCloudFactory.cu:875-903, :4201-4283) and N-view triangulation --
    checked for determinism and against the pushbroom generator's ground truth."""
    import scene
    from ssrlcv_amd import pipeline
    S, V = 8192, 8
    imgs, pbs, rig, sc = scene.pushbroom_views(V, S)
    seed, _ = H.load_seed_features()
    ws = pipeline.Workspace()
    res = pipeline.reconstruct(imgs, None, seed_features=seed, mode=0, pushbroom=pbs, ws=ws)
    assert len(ws.plans) == 1                      # the plan was reused, not one per strip
    nfeat = [f.numel() // 152 for f in res["features"]]
    assert len(nfeat) == V and all(5e5 < n < 8e6 for n in nfeat), nfeat
    assert len(res["pairs"]) == 28
    mm, kp, pts = res["matches"], res["keypoints"], res["points"].cpu().numpy()
    assert len(mm) > 100000 and pts.shape == (len(mm), 3) and np.isfinite(pts).all()
    assert mm["numKeyPoints"].max() >= 4 and set(np.unique(kp["parentId"])) == set(range(V))
    err = _ground_truth_error(rig, sc, mm, kp, pts)
    good = float((err < 0.2).mean())
    print("config[4] at size: %s features, %d multi-matches (up to %d views), cloud error vs ground truth: median %.4f km, "
          "%.1f %% within 0.2 km; stages %s" % (nfeat, len(mm), int(mm["numKeyPoints"].max()), np.median(err), 100 * good,
                                                  {k: round(v * 1e3) for k, v in ws.times.items()}))
    # Brute-force matching without a geometric constraint over 2-3 million features per strip leaves many more wrong
    # matches than at 2048^2 (there 71 % of the bundles land within 0.2 km), and in an 8-view merge one wrong member spoils
    # a bundle: upstream removes these with its statistical filters after triangulation.  What this checks is the geometry
    # at size: a large consistent set whose pushbroom bundles intersect on the generator's ground truth.
    inl = err[err < 0.2]
    print("config[4] at size: %d bundles within 0.2 km, their median error %.4f km" % (len(inl), np.median(inl)))
    assert len(inl) > 500000 and np.median(inl) < 0.03
    # The filtering stage (on the device: csrc/filter.hip): upstream's deterministicStatisticalFilter(3 sigma, 10 %)
    # -- doFiltering runs it once and notes "could increase for more aggressive filtering" -- repeated until nine bundles
    # in ten sit on the ground truth.  Each pass removes the worst tail of what is left and keeps the consistent set.
    fmm, fkp, dev, passes = mm, kp, dict(res["device"]), 0
    fgood = good
    while fgood <= 0.9 and passes < 60:
        fmm, fkp = pipeline.apply_filters(fmm, fkp, dev, None, [("statistical", 3.0, 0.1)] * 4, pushbroom=pbs)
        passes += 4
        fpts = pipeline.triangulate(fmm, fkp, None, nview=True, pushbroom=pbs, dev=dev).cpu().numpy()
        ferr = _ground_truth_error(rig, sc, fmm, fkp, fpts)
        fgood = float((ferr < 0.2).mean())
    print("config[4] at size, filtered: %d passes, %d of %d multi-matches kept, %.1f %% within 0.2 km (%d bundles), median %.4f km" %
          (passes, len(fmm), len(mm), 100 * fgood, int((ferr < 0.2).sum()), np.median(ferr)))
    assert fgood > 0.9, fgood
    assert (ferr < 0.2).sum() > 0.95 * len(inl)     # the consistent set survives the filters
    # determinism: the whole flow again on the same workspace
    res2 = pipeline.reconstruct(imgs, None, seed_features=seed, mode=0, pushbroom=pbs, ws=ws)
    assert all(torch.equal(a, b) for a, b in zip(res["features"], res2["features"]))
    assert np.array_equal(mm, res2["matches"]) and np.array_equal(kp, res2["keypoints"])
    assert np.array_equal(pts, res2["points"].cpu().numpy())


_F16_SCRIPT = r"""
import os, sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import helpers as H
from ssrlcv_amd import capi
lib = H.oracle()
rng = np.random.default_rng(3)
def feats(n, seed):
    r = np.random.default_rng(seed)
    f = np.zeros(n, H.FEATURE); f["parent"] = -1
    f["values"] = r.integers(0, 256, (n, 128), dtype=np.uint8); f["loc"] = r.uniform(0, 1024, (n, 2)).astype(np.float32)
    return f
q, t = feats(3000, 1), feats(5000, 2)
t["values"][:500] = q["values"][:500]          # exact duplicates: distance 0, ties broken by the winner rule
t["values"][600:700] = 255; q["values"][10:20] = 0   # extreme rows: the largest squared distances
params = capi.make_match_params(0, 0, 1, 0.0, 0.0, 0.6, 3.0e7)
out = capi.to_host(capi.match(capi.to_dev(q), len(q), capi.to_dev(t), len(t), params, capi.OUT_DMATCH), H.DMATCH, len(q))
ref = H.oracle_match_dmatch(lib, 0, 0, q, 1, t, None, None, 0, 0, None, 0.6, 3.0e7)
assert np.array_equal(out["invalid"], ref["invalid"]) and np.array_equal(out["distance"], ref["distance"])
assert np.array_equal(out["kp1_loc"], ref["kp1_loc"])
v = H.load_view("Pipeline2View"); cams = v["cameras"]
params = capi.make_match_params(1, 0, 1, 25.0, 5.0, 0.6, 200.0 * 200.0, cams[0:1], capi.projection_matrix(cams[1:2]))
out = capi.to_host(capi.match(capi.to_dev(q), len(q), capi.to_dev(t), len(t), params, capi.OUT_UINT2_PAIR), H.UINT2_PAIR, len(q))
proj = H.oracle_projection(lib, cams[1:2])
ref = H.oracle_match_pairs(lib, 1, 0, q, 1, t, cams[0:1], proj, 25.0, 5.0, None, 0.6, 200.0 * 200.0)
assert np.array_equal(out["a"], ref["a"]) and np.array_equal(out["b"], ref["b"])
print("F16 OK")
"""


_RELEASE_SCRIPT = r"""
import os, sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import helpers as H
from ssrlcv_amd import capi, _lib
assert _lib.LIB_PATH.endswith("libssrlcv_hip_release.so")
lib = H.oracle()
img = H.synthetic_image(320, 256, seed=5)
plan = capi.SiftPlan(320, 256)
plan.extract(capi.to_dev(img))
g, o = plan.features_host(H.FEATURE), H.oracle_sift(lib, img)
assert len(g) == len(o) and len(g) > 50
H.assert_features_equal(g, o)
print("RELEASE OK", len(g))
"""


def test_release_build_ignores_the_environment_and_is_bit_exact():
    """libssrlcv_hip_release.so (csrc/Makefile `release`: the developer switches compiled out) in a child process whose
    environment sets switches that would change the DEVELOPER build's code path: the features equal the oracle's."""
    rel = os.path.join(ROOT, "ssrlcv_amd", "libssrlcv_hip_release.so")
    if not os.path.exists(rel):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "ssrlcv_amd", "csrc"), "release"])
    env = dict(os.environ, SSRLCV_HIP_LIB=rel, SSRLCV_GAUSS_VALU="1", SSRLCV_DOGX_NPX="1", SSRLCV_SIFT_SERIAL="1", SSRLCV_MATCH_F16="1")
    r = subprocess.run([sys.executable, "-c", _RELEASE_SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RELEASE OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_F16_SCENE_SCRIPT = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import helpers as H
from ssrlcv_amd import pipeline
d = np.load(%(npz)r)
feats = [torch.from_numpy(d["f%%d" %% i]).cuda() for i in range(3)]
pairs = pipeline.match_pairs(feats, d["cams"].view(H.CAMERA), None, 25.0, 5.0, mode=1)
np.savez(%(out)r, **{"p%%d" %% k: v.cpu().numpy() for k, v in pairs.items()})
print("F16 SCENE OK")
"""


def test_band_culled_pairs_at_size_equal_the_fp16_kernels(tmp_path):
    """Three 2048^2 scene views, the three double-constrained pair matches: the default kernel (int8 MFMA; hit groups
    from the middle of the band outwards, one tile in flight by LDS-DMA, shared bounds) against the fp16 kernel's plain
    walk of the same boxes (SSRLCV_MATCH_F16=1, a child process).  Two independent walks and epilogues, real band
    geometry, ~10^5 features per view: every validated pair list must be identical."""
    import scene
    from ssrlcv_amd import pipeline
    imgs, cams, _, _ = scene.pinhole_views(3, 2048)
    res = pipeline.reconstruct(imgs, cams, seed_features=None, mode=1)
    feats = res["features"]
    pairs = pipeline.match_pairs(feats, cams, None, 25.0, 5.0, mode=1)
    assert sum(v.numel() // 16 for v in pairs.values()) > 50000
    npz, out = str(tmp_path / "feats.npz"), str(tmp_path / "pairs.npz")
    np.savez(npz, cams=cams.view(np.uint8), **{"f%d" % i: f.cpu().numpy() for i, f in enumerate(feats)})
    env = H.dev_env(SSRLCV_MATCH_F16="1")
    r = subprocess.run([sys.executable, "-c", _F16_SCENE_SCRIPT % {"root": ROOT, "npz": npz, "out": out}], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "F16 SCENE OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    other = np.load(out)
    for k, v in pairs.items():
        assert np.array_equal(v.cpu().numpy(), other["p%d" % k]), "pair %d" % k


def test_fp16_mfma_matcher_is_bit_exact_too():
    """The matcher's fp16-MFMA formulation (v_mfma_f32_32x32x16_f16 with the norms carried as base-1024 digits; what
    the north star names) is selected once per process by SSRLCV_MATCH_F16=1, so it is exercised in a child process:
    brute force with duplicates / extreme rows and the orbit mode, both equal to the oracle entry for entry."""
    env = H.dev_env(SSRLCV_MATCH_F16="1")
    r = subprocess.run([sys.executable, "-c", _F16_SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "F16 OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_match_arithmetic_is_selectable_at_run_time_in_the_release_build(capi, oracle_lib):
    """ssrlcv_hip_set_match_arithmetic (the product's own switch, present in the release library): the fp16-MFMA formulation
    the north star names and the default int8 one, in ONE process, brute force with duplicates / extreme rows and the
    orbit mode, each equal to the oracle entry for entry -- and therefore to each other."""
    from ssrlcv_amd import _lib
    assert _lib.flavour() == "release"

    def feats(n, seed):
        r = np.random.default_rng(seed)
        f = np.zeros(n, H.FEATURE)
        f["parent"] = -1
        f["values"] = r.integers(0, 256, (n, 128), dtype=np.uint8)
        f["loc"] = r.uniform(0, 1024, (n, 2)).astype(np.float32)
        return f
    q, t = feats(3000, 1), feats(5000, 2)
    t["values"][:500] = q["values"][:500]
    t["values"][600:700] = 255
    q["values"][10:20] = 0
    cams = H.load_view("Pipeline2View")["cameras"]
    proj = H.oracle_projection(oracle_lib, cams[1:2])
    ref0 = H.oracle_match_dmatch(oracle_lib, 0, 0, q, 1, t, None, None, 0, 0, None, 0.6, 3.0e7)
    ref1 = H.oracle_match_pairs(oracle_lib, 1, 0, q, 1, t, cams[0:1], proj, 25.0, 5.0, None, 0.6, 200.0 * 200.0)
    assert capi.get_match_arithmetic() == capi.MATCH_ARITH_I8
    assert capi.LIB.ssrlcv_hip_set_match_arithmetic(ctypes.c_int(7)) != 0   # unknown values are refused, the setting stays
    try:
        for arith in (capi.MATCH_ARITH_F16, capi.MATCH_ARITH_I8):
            capi.set_match_arithmetic(arith)
            assert capi.get_match_arithmetic() == arith
            p0 = capi.make_match_params(0, 0, 1, 0.0, 0.0, 0.6, 3.0e7)
            out = capi.to_host(capi.match(capi.to_dev(q), len(q), capi.to_dev(t), len(t), p0, capi.OUT_DMATCH), H.DMATCH, len(q))
            assert np.array_equal(out["invalid"], ref0["invalid"]) and np.array_equal(out["distance"], ref0["distance"])
            assert np.array_equal(out["kp1_loc"], ref0["kp1_loc"])
            p1 = capi.make_match_params(1, 0, 1, 25.0, 5.0, 0.6, 200.0 * 200.0, cams[0:1], capi.projection_matrix(cams[1:2]))
            out = capi.to_host(capi.match(capi.to_dev(q), len(q), capi.to_dev(t), len(t), p1, capi.OUT_UINT2_PAIR), H.UINT2_PAIR, len(q))
            assert np.array_equal(out["a"], ref1["a"]) and np.array_equal(out["b"], ref1["b"])
    finally:
        capi.set_match_arithmetic(capi.MATCH_ARITH_I8)


@pytest.mark.parametrize("depth", [2, 3, 4])
def test_convert_to_bw_bit_exact(capi, oracle_lib, depth):
    rng = np.random.default_rng(depth)
    n = 300 * 200
    color = rng.integers(0, 256, n * depth, dtype=np.uint8)
    color[: 4 * depth] = 255
    got = capi.convert_to_bw(capi.to_dev(color), depth, n).cpu().numpy()
    ref = np.zeros(n, np.uint8)
    oracle_lib.oracle_convert_to_bw(H.P(color), ctypes.c_uint32(depth), H.P(ref), ctypes.c_size_t(n))
    assert np.array_equal(got, ref)
    px = color.reshape(n, depth).astype(np.int64)
    want = px[:, 0] if depth == 2 else px[:, 0] // 4 + px[:, 1] // 2 + px[:, 2] // 4
    assert np.array_equal(ref, want.astype(np.uint8))


def test_keypoint_capacity_overflow_is_reported_and_truncates(capi, oracle_lib):
    """A plan with a tiny key-point capacity: the lists are truncated at the capacity (in the reference's order), the
    overflow mask names the octaves, count() raises; the same image with the default capacity matches the oracle."""
    img = H.synthetic_image(512, 512, seed=4)
    small = capi.SiftPlan(512, 512, max_keypoints_per_octave=4096)
    small.extract(capi.to_dev(img))
    assert small.overflow_mask() & 1
    with pytest.raises(capi.SsrlcvError):
        small.count()
    torch.cuda.synchronize()
    n = int(small.num_features.item())
    of = H.oracle_sift(oracle_lib, img)
    assert 0 < n < len(of)
    full = capi.SiftPlan(512, 512)
    full.extract(capi.to_dev(img))
    assert full.overflow_mask() == 0
    H.assert_features_equal(full.features_host(H.FEATURE), of)


def _nccl_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        from ssrlcv_amd import dist as sd
        dev = torch.device("cuda", 0)
        items = {k: torch.arange(100 * (k + 1), dtype=torch.uint8, device=dev) for k in range(3)}
        out = sd.exchange_keyed(items, 3, sd.image_owner)
        assert all(o.is_cuda and torch.equal(o, items[k]) for k, o in enumerate(out))
        owners = sd.assign_pairs([1000, 2000, 3000], world)
        out = sd.exchange_keyed(items, 3, lambda p, _w: owners[p])
        assert all(torch.equal(o, items[k]) for k, o in enumerate(out))
        parts = sd.all_gather_bytes(items[2])
        assert len(parts) == 1 and torch.equal(parts[0], items[2])
        assert sd.all_gather_bytes(torch.zeros(0, dtype=torch.uint8, device=dev))[0].numel() == 0
        t = torch.arange(612, dtype=torch.float32, device=dev)
        sd.all_reduce_sum(t)
        assert torch.equal(t, torch.arange(612, dtype=torch.float32, device=dev))
        dist.barrier()
        open(os.path.join(out_dir, "nccl_ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_exchanges_run_on_the_nccl_backend(tmp_path):
    """The collectives of ssrlcv_amd.dist on CUDA tensors under the `nccl` (= RCCL) backend -- the branch the gloo tests
    never take: device-side size all-reduce, padded uint8 all-gather, float all-reduce, barrier.  One rank (one GPU here;
    RCCL refuses two ranks on one device), so the data path is the identity, but every call is the one an 8-GPU run makes."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_nccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "nccl_ok").exists()
