"""Developer tool: config[3] through the C++ mirror's class-level calls (tests/cpp/sharded_match_test.cpp bench-flow) with its
per-call diagnostics (SSRLCV_FLOW_DIAG=1).  usage: python3 tools/flow_diag.py [size] [views] [iters]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import helpers as H
import scene
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
views = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
imgs, cams, _, _ = scene.pinhole_views(views, size)
info = H.host_typeinfo()
d = tempfile.mkdtemp(prefix="ssrlcv_flow_", dir="/tmp")
for i, im in enumerate(imgs):
    H.write_uty(os.path.join(d, "pixels_%d.uty" % i), *info["uchar"], 1, im.cpu().numpy().reshape(-1))
    H.write_cpimg(os.path.join(d, "%d_%s.cpimg" % (i, info["Image"][0])), i, (size, size), cams[i:i + 1])
seed, _ = H.load_seed_features()
H.write_uty(os.path.join(d, "-1_%s.uty" % info["Feature"][0]), *info["Feature"], 2, seed)
mode = os.environ.get("FLOW_DIAG_PARENT", "")
keep = []
if "mem" in mode:      # the parent holds a lot of device memory
    import torch
    keep.append(torch.empty(60 << 30, dtype=torch.uint8, device="cuda"))
if "plans" in mode:    # the parent has run the library (side streams, events, plans alive)
    from ssrlcv_amd import capi
    for im in imgs:
        p = capi.SiftPlan(size, size)
        p.extract(im)
        p.count()
        keep.append(p)
if "nview" in mode:    # the parent has run the Python flow (pinned staging buffers, matcher workspaces, merge)
    from ssrlcv_amd import pipeline
    ws = pipeline.Workspace()
    for _ in range(2):
        res = pipeline.reconstruct(imgs, cams, seed_features=H.load_seed_features()[0], mode=1, ws=ws, ba=True)
    keep.append((ws, res))
if "coop" in mode:     # the parent has made ONE cooperative launch (the device merge), on a toy input
    import numpy as np, torch
    from ssrlcv_amd import capi
    pairs = np.array([[0, 1, 1, 2], [0, 3, 1, 4], [0, 1, 2, 5], [1, 2, 2, 5]], np.uint32)   # pairs (0,1) x2, (0,2), (1,2)
    out = capi.merge_matches_device([10, 10, 10], [2, 1, 1], capi.to_dev(pairs.view(np.uint8).reshape(-1)), None)
    torch.cuda.synchronize()
    print("parent merge:", out[2], out[3])
if "match" in mode:    # ... or only the matcher
    import numpy as np, torch
    from ssrlcv_amd import capi
    f = np.zeros(5000, H.FEATURE); f["values"] = np.random.default_rng(1).integers(0, 256, (5000, 128), dtype=np.uint8)
    params = capi.make_match_params(0, 0, 1, 0.0, 0.0, 0.6, 3.0e7)
    capi.match(capi.to_dev(f), 5000, capi.to_dev(f), 5000, params, capi.OUT_DMATCH)
    torch.cuda.synchronize()
if "limit" in mode:
    H.limit_openmp()
if "free" in mode:
    import torch
    del imgs
    torch.cuda.empty_cache()
else:
    del imgs
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
if "omp" in mode:
    env.update(OMP_NUM_THREADS="16", OMP_WAIT_POLICY="PASSIVE")
if "capture" in mode:
    r = subprocess.run([H.SHARDED_BIN, "bench-flow", d, str(views), str(iters)], env=env, capture_output=True, text=True)
    print(r.stdout[-400:])
else:
    r = subprocess.run([H.SHARDED_BIN, "bench-flow", d, str(views), str(iters)], env=env)
sys.exit(r.returncode)
