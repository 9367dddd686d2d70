"""bench.py -- SIFT extract (Mpix/s) + 128-D brute-force match (Mmatches/s) on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it
under torch.distributed.run, one rank per GPU.  One STEP = one pass of the SIFT hot path
(ssrlcv_hip_sift_extract: pyramid -> DoG -> extrema -> refinement -> orientation -> descriptors) over this rank's
pair of synthetic u8 images already resident in HBM (BASELINE.json config[2]: 2-view 4096 x 4096).  Ranks own different
image pairs (weak scaling, no data-path collective inside the SIFT stage).  value = total input pixels of all ranks /
max-over-ranks time.  Inputs come from tools/scene.py (PCG32 terrain + texture rendered through
Image::Camera, ~0.03 features per pixel like the reference's everest imagery).

Besides the contract keys the JSON line carries
  roofline      : the DoG-pyramid stage (ssrlcv_hip_sift_build_dog) against HBM, algorithmic bytes 362.25*W*H per image
                  (`frac`; `frac_s1_s8` adds 8d's figure for the extrema search the stage also runs),
                  duration from HIP events around the stage inside the timed steps;
  describe      : the key-point stage (ssrlcv_hip_sift_describe: extrema .. orientation .. descriptors), the larger
                  share of a step: ms per image and ns per feature from events in the same steps, and its VALU-issue
                  roofline from the committed PMC reduction (profiles/);
  matcher       : Mmatches/s of ssrlcv_hip_match_u8x128 (pairs compared / time) on Nq = Nt synthetic descriptors, with
                  its int8-MFMA roofline (2*128*Nq*Nt op), and the band-culled orbit mode;
  class_api     : the same extraction through ssrlcv::SIFT_FeatureFactory::generateFeatures from host-state pixels
                  (PCIe included), timed by the C++ host-mirror binary;
  nview         : BASELINE config[3] as one more measured step: V views sharded over the ranks, RCCL all-gather of the
                  feature arrays and of the uint2_pair arrays, replicated merge, bundle-range N-view triangulation,
                  all-gather of the cloud, BA error sweep with its all-reduce -- with the wall share of every stage;
  cpu_baseline  : the CPU oracle (oracle/, a port restating the reference's kernels) on BASELINE config[1] -- the 2-view
                  1024 x 1024 pair of the same generator through the whole flow, stage by stage -- plus SIFT on one
                  2048 x 2048 view: one warm-up + median of 5 repetitions per stage, on rank 0's host cores.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import helpers as _H  # noqa: E402  (test-side helpers: POD dtypes and the loader of the CPU oracle the cpu_baseline leg times)

OMP_THREADS = _H.limit_openmp()  # the OpenMP team = the CPUs this process is granted (cgroup quota), set before torch loads

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak (~2.5 PF)
MFMA_I8_PEAK_TOPS = 5000.0     # dense int8 MFMA peak: twice the bf16 rate per clock (MI355X_MICROARCH.md)
# Vector-ALU issue peak in wave-instructions per ns: 256 CUs x 4 SIMDs x 2.4 GHz, a wave64 instruction every TWO clocks (the
# CDNA4 SIMD is 32 lanes wide: MI355X_MICROARCH.md glossary; tools/valu_rate.hip measures 2.4-2.9 clocks for the simple
# ops and 4.2 for an FMA with an SGPR source).  Round 2 priced against one per four clocks and reported a kernel above 1.
VALU_PEAK_GINST = 1024 * 2.4 / 2.0
# what v_mfma_i32_32x32x32_i8 sustains on this part with random operands, issued back to back from registers with nothing
# else in the way (tools/mfma_i8_peak.hip: 4.96 POP/s on zeros, 3.38 on random bytes -- the clock the chip holds under the
# load depends on the toggling): the practical ceiling beside the data-sheet peak the roofline is priced against
MFMA_I8_SUSTAINED_TOPS = 3380.0


def synth_images(n, w, h, seed, device):
    """Round-1 input (multi-scale noise, ~0.1 features per pixel: 3x denser than real imagery).  Kept for the dense
    full-size parity tests; the benchmark itself renders tools/scene.py scenes."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(0x53524C43 + seed)
    imgs = []
    for _ in range(n):
        img = torch.zeros(1, 1, h, w, device=device)
        amp = 30.0
        for k in range(1, 8):
            gh, gw = max(2, h >> k), max(2, w >> k)
            noise = torch.randn(1, 1, gh, gw, device=device, generator=g)
            img += torch.nn.functional.interpolate(noise, size=(h, w), mode="bicubic", align_corners=False) * amp
            amp *= 0.8
        img = (img + 128.0).round().clamp(0, 255).to(torch.uint8).reshape(h, w).contiguous()
        imgs.append(img)
    return imgs


def synth_descriptors(n, seed):
    """uniform u8 vectors L2-normalised to ~255 like real SIFT descriptors."""
    import helpers as H
    rng = np.random.default_rng(seed)
    v = rng.integers(0, 256, (n, 128)).astype(np.float32)
    v = np.rint(v * (255.0 / np.sqrt((v * v).sum(1, keepdims=True)))).clip(0, 255)
    f = np.zeros(n, H.FEATURE)
    f["parent"] = -1
    f["values"] = v.astype(np.uint8)
    f["loc"] = rng.uniform(0, 4096, (n, 2)).astype(np.float32)
    return f


def git_head():
    """The commit this tree was built from: git when it is there, else the stamp __graft_entry__.build() leaves beside the
    library (the GPU box receives a snapshot without .git)."""
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        pass
    try:
        return open(os.path.join(ROOT, "ssrlcv_amd", "_build_commit.txt")).read().strip() or None
    except OSError:
        return None


def bench_matcher(capi, torch, nq, nt, iters, arithmetic="int8"):
    """ssrlcv_hip_match_u8x128, brute force, Nq x Nt synthetic descriptors.  arithmetic = "int8" (the default formulation:
    v_mfma_i32_32x32x32_i8, priced against the int8 dense peak) or "f16" (v_mfma_f32_32x32x16_f16, the formulation the
    north star names, priced against the fp16 dense peak); both exact, selected by ssrlcv_hip_set_match_arithmetic."""
    q = synth_descriptors(nq, 1)
    t = synth_descriptors(nt, 2)
    rng = np.random.default_rng(3)
    dup = rng.choice(nq, nq // 4, replace=False)   # 25 % planted near-duplicates (+-3)
    tgt = rng.choice(nt, nq // 4, replace=False)
    t["values"][tgt] = np.clip(q["values"][dup].astype(np.int32) + rng.integers(-3, 4, (len(dup), 128)), 0, 255)
    q_d, t_d = capi.to_dev(q), capi.to_dev(t)
    ws = capi.match_workspace(nq, nt)
    out = capi.dev_bytes(nq * 48)
    params = capi.make_match_params(0, 0, 1, 0.0, 0.0, 0.6, 200.0 * 200.0)
    before = capi.get_match_arithmetic()
    capi.set_match_arithmetic(capi.MATCH_ARITH_F16 if arithmetic == "f16" else capi.MATCH_ARITH_I8)
    try:
        capi.match(q_d, nq, t_d, nt, params, capi.OUT_DMATCH, workspace=ws, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            capi.match(q_d, nq, t_d, nt, params, capi.OUT_DMATCH, workspace=ws, out=out)
        e1.record()
        torch.cuda.synchronize()
    finally:
        capi.set_match_arithmetic(before)
    ms = e0.elapsed_time(e1) / iters
    pairs = float(nq) * float(nt)
    tops = 2.0 * 128.0 * pairs / (ms * 1e-3) / 1e12
    res = {"value": pairs / (ms * 1e-3) / 1e6, "unit": "Mmatches/s", "nq": nq, "nt": nt, "ms": ms,
           "output_matches_per_s": nq / (ms * 1e-3), "dtype": arithmetic,
           "checksum": int(torch.sum(out.view(torch.int32)[10::12].to(torch.int64)).item())}  # DMatch.distance bits: equal for both formulations
    if arithmetic == "f16":
        res["roofline"] = {"bound": "mfma", "achieved": tops, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": tops / MFMA_F16_PEAK_TFLOPS, "traffic": None,
                           "kernel": "k_match (v_mfma_f32_32x32x16_f16, exact: norms as base-1024 digits in a ninth K step), whole "
                                     "ssrlcv_hip_match_u8x128 call; flops = 2*128*Nq*Nt (BASELINE.md section 4: the K = 144 the kernel "
                                     "really issues is not credited), priced against the fp16 dense peak"}
    else:
        res["roofline"] = {"bound": "mfma", "achieved": tops, "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                           "frac": tops / MFMA_I8_PEAK_TOPS, "traffic": None,
                           "frac_of_fp16_peak": tops / MFMA_F16_PEAK_TFLOPS,
                           "sustained_peak_random_operands": MFMA_I8_SUSTAINED_TOPS,
                           "frac_of_sustained_peak": tops / MFMA_I8_SUSTAINED_TOPS,
                           "kernel": "k_match_i8 (v_mfma_i32_32x32x32_i8, exact), whole ssrlcv_hip_match_u8x128 call; "
                                     "ops = 2*128*Nq*Nt, priced against the int8 dense peak (2x the fp16 peak the "
                                     "north star names: frac_of_fp16_peak is the same rate against that); "
                                     "sustained_peak_random_operands = the same instruction alone on random bytes "
                                     "(tools/mfma_i8_peak.hip: the part is power-limited there, 4.96 POP/s on zeros); "
                                     "matrix-pipe busy cycles and clock: profiles/r05_matcher_lab_pmc.txt"}
    return res


def bench_matcher_epipolar(capi, torch, n, size, iters):
    """The orbit mode of doFeatureMatching (matchFeaturesDoubleConstrained, epsilon 25 px, delta 5 km), features spread
    uniformly over a size x size image seen by the fixture's camera pair rescaled to that size.  Reported as effective
    pair comparisons per second (Nq*Nt / time): the band-culled path skips most of them without computing a distance."""
    import helpers as H
    q, t = synth_descriptors(n, 1), synth_descriptors(n, 2)
    rng = np.random.default_rng(5)
    q["loc"] = rng.uniform(0, size, (n, 2)).astype(np.float32)
    t["loc"] = rng.uniform(0, size, (n, 2)).astype(np.float32)
    cams = H.load_view("Pipeline2View")["cameras"].copy()
    scale = float(cams["size"][0][0]) / size
    cams["dpix"] = cams["dpix"] * scale
    cams["size"] = int(size)
    params = capi.make_match_params(1, 0, 1, 25.0, 5.0, 0.6, 200.0 * 200.0, cams[0:1], capi.projection_matrix(cams[1:2]))
    q_d, t_d = capi.to_dev(q), capi.to_dev(t)
    ws = capi.match_workspace(n, n)
    out = capi.dev_bytes(n * 48)
    capi.match(q_d, n, t_d, n, params, capi.OUT_DMATCH, workspace=ws, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        capi.match(q_d, n, t_d, n, params, capi.OUT_DMATCH, workspace=ws, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return {"value": float(n) * float(n) / (ms * 1e-3) / 1e6, "unit": "effective Mmatches/s", "nq": n, "nt": n, "ms": ms,
            "mode": "double-constrained, epsilon 25 px, delta 5 km, %dx%d images" % (size, size)}


def cpu_baseline(size=1024, big=2048, reps=5, device=None, match_queries=4096):
    """SURVEY section 8(d) / BASELINE.md section 3: the CPU oracle (a PORT restating the reference's kernels: the reference
    has no CPU compute path) on the host cores of this box, stage by stage with the GPU path's boundaries, on BASELINE
    config[1] -- the 2-view size x size pair of the benchmark's own scene generator, both images, the whole flow -- plus the
    SIFT stages on one big x big view.  One warm-up, then `reps` timed repetitions per stage; medians reported.
    `value` = SIFT extract Mpix/s of the pair (pyramid + key points / descriptors), the unit of the headline metric."""
    import ctypes
    import torch
    import helpers as H
    import scene
    lib = H.oracle()
    lib.oracle_sift_create.restype = ctypes.c_void_p
    lib.oracle_sift_features.restype = ctypes.c_int

    def med(ts):
        return float(np.median(ts))

    def sift_stages(img):
        """-> (seconds scale space S1-S7 + DoG, seconds S8-S14, features)"""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        t0 = time.perf_counter()
        hnd = ctypes.c_void_p(lib.oracle_sift_create(H.P(img), ctypes.c_uint32(w), ctypes.c_uint32(h)))
        t1 = time.perf_counter()
        out = ctypes.c_void_p()
        n = lib.oracle_sift_features(hnd, ctypes.c_uint32(2), ctypes.c_float(0.8), ctypes.c_float(1.5), ctypes.c_float(6.0),
                                     ctypes.byref(out))
        t2 = time.perf_counter()
        f = np.ctypeslib.as_array(ctypes.cast(out, ctypes.POINTER(ctypes.c_uint8)), shape=(n * 152,)).view(H.FEATURE).copy()
        lib.oracle_free(out)
        lib.oracle_sift_destroy(hnd)
        return t1 - t0, t2 - t1, f

    dev = device if device is not None else torch.device("cuda")
    imgs_d, cams, _, _ = scene.pinhole_views(2, size, device=dev)
    imgs = [im.cpu().numpy() for im in imgs_d]
    seed, _ = H.load_seed_features()
    t_pyr, t_kp, feats = [], [], None
    for r in range(reps + 1):                     # repetition 0 is the warm-up
        ts = [sift_stages(im) for im in imgs]
        if r:
            t_pyr.append(sum(t[0] for t in ts))
            t_kp.append(sum(t[1] for t in ts))
        feats = [t[2] for t in ts]
    f0, f1 = feats
    for f, pid in ((f0, 0), (f1, 1)):
        f["parent"] = pid
    # doFeatureMatching's orbit flow (src/Pipeline.cu:150-185): seed distances of the query image, double-constrained match
    # The matcher's work is linear in the queries: a bounded sample of them (every k-th feature of image 0) against ALL
    # features of image 1 keeps the leg inside its time budget (the full 78 k x 78 k pass alone takes ~40 s per repetition).
    proj = H.oracle_projection(lib, cams[1:2])
    fq = np.ascontiguousarray(f0[::max(1, len(f0) // match_queries)][:match_queries])
    t_match, dm = [], None
    for r in range(reps + 1):
        t0 = time.perf_counter()
        sd = H.oracle_seed_distances(lib, fq, seed)
        dm = H.oracle_match_dmatch(lib, 1, 0, fq, 1, f1, cams[0:1], proj, 25.0, 5.0, sd, 0.6, 200.0 * 200.0)
        if r:
            t_match.append(time.perf_counter() - t0)
    valid = dm[dm["invalid"] == 0]
    n = len(valid)
    kp = np.zeros(2 * n, H.KEYPOINT)
    kp["parentId"][0::2], kp["loc"][0::2] = valid["kp0_parent"], valid["kp0_loc"]
    kp["parentId"][1::2], kp["loc"][1::2] = valid["kp1_parent"], valid["kp1_loc"]
    mm = np.zeros(n, H.MULTIMATCH)
    mm["numKeyPoints"], mm["index"] = 2, np.arange(n) * 2
    t_tri = []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        bundles, lines, _ = H.oracle_bundles(lib, mm, kp, cams)
        H.oracle_triangulate(lib, False, bundles, lines)
        if r:
            t_tri.append(time.perf_counter() - t0)
    # one BundleAdjustTwoView iteration's finite-difference sweep: 612 evaluations of f(cameras) (:1059-1504)
    lib.oracle_ba_eval.restype = ctypes.c_float
    base = np.concatenate([np.concatenate([c["cam_pos"], c["cam_rot"]]) for c in cams]).astype(np.float32)
    K = 612
    t_ba = []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        for k in range(K if r else 8):
            p6 = base.copy()
            p6[k % 12] += np.float32(1e-4) * (1 + k // 12)
            lib.oracle_ba_eval(ctypes.c_uint32(n), H.P(mm), H.P(kp), H.P(cams), ctypes.c_uint32(2), H.P(p6))
        if r:
            t_ba.append(time.perf_counter() - t0)
    # SIFT stages on one bigger view
    big_img = scene.pinhole_views(1, big, device=dev)[0][0].cpu().numpy()
    b_pyr, b_kp, fb = [], [], None
    for r in range(reps + 1):
        a, b, fb = sift_stages(big_img)
        if r:
            b_pyr.append(a)
            b_kp.append(b)
    sift_s = med(t_pyr) + med(t_kp)
    stages = {
        "pyramid_S1_S7_dog": {"s": med(t_pyr), "Mpix_per_s": 2 * size * size / med(t_pyr) / 1e6},
        "keypoints_descriptors_S8_S14": {"s": med(t_kp), "features": [int(len(f0)), int(len(f1))],
                                         "us_per_feature": med(t_kp) * 1e6 / max(1, len(f0) + len(f1))},
        "seed_distances_and_match_M1_M5": {"s": med(t_match), "Mmatches_per_s": float(len(fq)) * len(f1) / med(t_match) / 1e6,
                                           "mode": "double-constrained (epsilon 25 px, delta 5 km), %d seed features" % len(seed),
                                           "queries_sampled": int(len(fq)), "queries_total": int(len(f0)), "targets": int(len(f1)),
                                           "s_extrapolated_to_all_queries": med(t_match) * len(f0) / max(1, len(fq)),
                                           "matches": int(n)},
        "bundles_triangulate_P1_P2": {"s": med(t_tri), "Mpoints_per_s": n / med(t_tri) / 1e6},
        "ba_sweep_P4": {"s": med(t_ba), "evaluations": K, "ms_per_evaluation": med(t_ba) / K * 1e3},
        "sift_%dx%d_one_view" % (big, big): {"pyramid_s": med(b_pyr), "keypoints_descriptors_s": med(b_kp), "features": int(len(fb)),
                                             "Mpix_per_s": big * big / (med(b_pyr) + med(b_kp)) / 1e6},
    }
    scale = len(f0) / max(1, len(fq))   # the sampled stages extrapolated to every query of image 0
    return {"value": 2 * size * size / sift_s / 1e6, "unit": "Mpix/s", "cores": OMP_THREADS, "kind": "port", "reps": reps,
            "stages": stages,
            "whole_flow_s_extrapolated": sift_s + (med(t_match) + med(t_tri) + med(t_ba)) * scale,
            "sample": "BASELINE config[1]: the 2-view %dx%d pair of the benchmark's scene generator through the whole flow on the "
                      "CPU oracle (SIFT on both images -> seed distances + double-constrained match of a bounded sample of the queries "
                      "against every target -> generateBundle + two-view triangulation of the matches found -> the 612-evaluation BA sweep), every stage timed with the GPU path's boundaries: one "
                      "warm-up + median of %d repetitions; plus the SIFT stages on one %dx%d view.  `value` = SIFT extract of the pair.  "
                      "OpenMP team = the host CPUs granted to this process (cores); single-threaded stages: bundles, "
                      "triangulation, BA.  The reference itself has no CPU compute path." % (size, size, reps, big, big)}


def class_api_leg(img_u8, size, value_c_abi, iters=5):
    """The same extraction through the drop-in C++ class API (ssrlcv::SIFT_FeatureFactory::generateFeatures on an Image
    whose Unity<unsigned char> pixels sit in host memory), timed by the host-mirror binary: the number a caller of the
    reference sees, H2D of the image (and the restore of its origin state) included; the second figure adds the D2H of
    the feature array."""
    exe = os.path.join(ROOT, "ssrlcv_amd", "host", "_build", "host_mirror_test")
    if not os.path.exists(exe):
        return {"error": "host mirror binary not built"}
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".raw", dir="/tmp", delete=False) as f:
        f.write(img_u8.cpu().numpy().tobytes())
        raw = f.name
    try:
        r = subprocess.run([exe, "bench", raw, str(size), str(size), str(iters)], capture_output=True, text=True, timeout=600)
    finally:
        os.unlink(raw)
    if r.returncode != 0:
        return {"error": (r.stdout + r.stderr)[-400:]}
    j = json.loads(r.stdout.strip().splitlines()[-1])
    mpix = size * size / 1e6
    return {"features": j["features"],
            "generateFeatures_from_host_pixels": {"ms_per_image": j["ms_generateFeatures_from_host_pixels"],
                                                  "value": mpix / (j["ms_generateFeatures_from_host_pixels"] * 1e-3), "unit": "Mpix/s"},
            "with_features_to_host": {"ms_per_image": j["ms_with_features_to_host"],
                                      "value": mpix / (j["ms_with_features_to_host"] * 1e-3), "unit": "Mpix/s"},
            "c_abi_resident_value": value_c_abi,
            "note": "one image at a time, synchronous like the reference's methods; plan, workspace and staging buffer "
                    "are pooled inside the factory; includes the image H2D, the restore of the pixels' origin state "
                    "(no copy back for grey pixels since round 5: the host copy is kept while the kernels read the device one) "
                    "and the exact-size feature copy"}


def class_api_flow_leg(imgs, cams, size, nview_ms, cloud_py, iters=3):
    """BASELINE config[3]'s flow through the class-level calls of the C++ mirror (host/Distributed.hpp at world 1:
    SIFT_FeatureFactory::generateFeatures per image from host pixels -> generateMatchesExhaustiveSharded -> nViewTriangulateSharded
    -> selectPairBundles + the 612-point sweep), timed by tests/cpp/sharded_match_test.cpp `bench-flow`, beside the Python C-ABI
    flow's step (`nview`); the C++ cloud is compared with the Python flow's bit for bit."""
    import shutil
    import tempfile
    import helpers as H
    exe = H.SHARDED_BIN
    if not os.path.exists(exe):
        return {"error": "sharded_match_test not built"}
    info = H.host_typeinfo()
    d = tempfile.mkdtemp(prefix="ssrlcv_flow_", dir="/tmp")
    try:
        for i, im in enumerate(imgs):
            H.write_uty(os.path.join(d, "pixels_%d.uty" % i), *info["uchar"], 1, im.cpu().numpy().reshape(-1))
            H.write_cpimg(os.path.join(d, "%d_%s.cpimg" % (i, info["Image"][0])), i, (size, size), cams[i:i + 1])
        seed, _ = H.load_seed_features()
        H.write_uty(os.path.join(d, "-1_%s.uty" % info["Feature"][0]), *info["Feature"], 2, seed)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        if os.environ.get("SSRLCV_BENCH_FLOW_DIAG"):
            env["SSRLCV_FLOW_DIAG"] = "1"
        r = subprocess.run([exe, "bench-flow", d, str(len(imgs)), str(iters)], capture_output=True, text=True, timeout=900, env=env)
        if r.returncode != 0:
            return {"error": (r.stdout + r.stderr)[-400:]}
        j = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
        if os.environ.get("SSRLCV_BENCH_FLOW_DIAG"):
            j["diag"] = [l for l in r.stderr.splitlines() if l.startswith("diag:")]
        raw = open(os.path.join(d, "300_6float3.uty"), "rb").read()
        pts = np.frombuffer(raw[len(raw) - 12 * j["points"]:], np.float32).reshape(-1, 3)
        same = cloud_py is not None and pts.shape == tuple(cloud_py.shape) and bool(np.array_equal(pts.view(np.uint32), cloud_py.view(np.uint32)))
    finally:
        shutil.rmtree(d, ignore_errors=True)
    j.update({"nview_ms_per_step": nview_ms, "flow_over_nview": j["flow_ms"] / nview_ms if nview_ms else None,
              "cloud_equals_python_flow": same,
              "note": "synchronous class-level calls, one image at a time, pixels in pageable host memory (each generateFeatures uploads "
                      "its image and restores the origin state); features stay on the device between the stages; RCCL communicator of one rank"})
    return j


def describe_roofline(ms_per_image, features, size):
    """Roofline of the key-point stage.  Its kernels are bound by vector-instruction issue (the sampling kernels) or move
    little data, so the stage is priced against the VALU issue peak with the wave-instruction count of the committed PMC
    reduction (instructions per feature do not depend on the run; the time does), and its algorithmic bytes against HBM
    beside it.  Per-kernel fractions come from the PMC run's own image and time (profiles/r05_describe_pmc.json)."""
    path = os.path.join(ROOT, "profiles", "r06_describe_pmc.json")
    if not os.path.exists(path):
        path = os.path.join(ROOT, "profiles", "r05_describe_pmc.json")
    out = {"ms_per_image": ms_per_image, "features_per_image": features,
           "ns_per_feature": ms_per_image * 1e6 / max(features, 1),
           "kernels": "flag-byte compaction, k_refine, k_flag_*, list partitions, k_polar, k_thetas, k_desc_consts, k_descriptors "
                      "(ssrlcv_hip_sift_describe; the extrema search itself runs in the pyramid stage's fused DoG pass); "
                      "from the library's stage-boundary event to the end of the fused extract, on the launching stream"}
    if os.path.exists(path):
        pmc = json.load(open(path))
        per_feature = pmc.get("valu_wave_instructions_per_feature")
        if per_feature:
            # instructions of this run's images: the per-PIXEL kernel (polar tables) as counted, the list and sampling
            # kernels scaled by the feature count (the PMC run's image has pmc["features_per_image"] features)
            total = pmc["valu_wave_instructions_per_image"]
            per_pixel = pmc.get("per_kernel", {}).get("k_polar", {}).get("valu_wave_instructions_per_image", 0.0)
            valu = per_pixel + (total - per_pixel) * features / max(pmc["features_per_image"], 1)
            ginst = valu / (ms_per_image * 1e6)
            out.update({"bound": "valu", "achieved": ginst, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s",
                        "frac": ginst / VALU_PEAK_GINST, "traffic": pmc.get("hbm_bytes_per_image"),
                        "algorithmic_bytes_per_image": pmc.get("algorithmic_bytes_per_image"),
                        "hbm_frac": pmc.get("algorithmic_bytes_per_image", 0) / (ms_per_image * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "per_kernel_from_pmc_run": {k: {"ms": round(v["ms_per_image_under_pmc"], 4),
                                                        "valu_frac": None if v.get("valu_frac") is None else round(v["valu_frac"], 3),
                                                        "hbm_frac": None if v.get("hbm_frac") is None else round(v["hbm_frac"], 3)}
                                                    for k, v in pmc.get("per_kernel", {}).items()},
                        "pmc_source": "profiles/%s @ %s (its own image: %d features)" % (os.path.basename(path), pmc.get("commit"), pmc.get("features_per_image", 0))})
    return out


def checksum64(x):
    """Order-sensitive 64-bit checksum of a byte buffer (torch tensor on any device, or numpy array): sum of the 32-bit words
    times an odd, position-dependent multiplier, in wrapping int64 arithmetic.  The N > 1 legs compare it across ranks and with
    the world-1 value committed under tests/golden/ (the scene and every kernel are deterministic)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x).view(np.uint8).reshape(-1)) if isinstance(x, np.ndarray) else x.contiguous().view(torch.uint8).reshape(-1)
    pad = (-t.numel()) % 4
    if pad:
        t = torch.cat([t, torch.zeros(pad, dtype=torch.uint8, device=t.device)])
    w = t.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    mult = torch.arange(w.numel(), dtype=torch.int64, device=w.device) * 2654435761 + 1
    return int((w * mult).sum().item()) & 0xFFFFFFFFFFFFFFFF


NVIEW_GOLDEN = os.path.join(ROOT, "tests", "golden", "nview_checksums.json")


def run_nview(args, torch, dist, capi, world, rank, dev, views, size, steps, warmup):
    """config[3]: V views, image/pair sharding over the ranks with the RCCL exchanges (ssrlcv_amd/pipeline.py).
    Every step is timed on its own (the flow ends in host arrays: it is synchronous); the leg reports median / min / max per
    step and per stage over `steps` >= 6 steps, and says so (`stall`) when the slowest step took more than twice the median --
    round 5's driver run had a 30 ms host-allocation stall in one of two steps and reported their mean."""
    import helpers as H
    import scene
    from ssrlcv_amd import pipeline
    from ssrlcv_amd import dist as sd
    imgs, cams, _, _ = scene.pinhole_views(views, size, device=dev)
    seed, _ = H.load_seed_features()
    ws = pipeline.Workspace()

    def step():
        return pipeline.reconstruct(imgs, cams, seed_features=seed, mode=1, ws=ws, ba=True)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    # warm-up the way the timed loop runs: the previous result is still alive while the next step is made, so the result
    # pool (pipeline._result_buffer) needs its second buffer before the clock starts
    res = None
    for _ in range(max(warmup, 2)):
        res = step()
    ws.times.clear()
    barrier()
    per_step, per_stage = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        before = dict(ws.times)
        ts = time.perf_counter()
        res = step()
        per_step.append((time.perf_counter() - ts) * 1e3)
        per_stage.append({k: (v - before.get(k, 0.0)) * 1e3 for k, v in ws.times.items()})
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if dist is None or dist.get_backend() != "gloo" else "cpu")
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    stage_names = list(per_stage[0].keys())

    def stats(vals):
        return {"median": float(np.median(vals)), "min": float(np.min(vals)), "max": float(np.max(vals))}
    mine = {k: float(np.median([st[k] for st in per_stage])) for k in stage_names}
    mine_stats = {k: stats([st[k] for st in per_stage]) for k in stage_names}
    step_stats = stats(per_step)
    per_rank = [mine]
    if dist is not None:   # every rank's stage times: the spread is the load imbalance (pairs of different cost, images per rank)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    spread = {k: {"min": min(r.get(k, 0.0) for r in per_rank), "max": max(r.get(k, 0.0) for r in per_rank)} for k in stage_names}
    nf = [int(f.numel() // 152) for f in res["features"]]
    owners = sd.assign_pairs(nf, world)
    pairs = sd.pair_list(views)
    cost = [float(nf[i]) * nf[j] for i, j in pairs]
    load = [sum(c for c, o in zip(cost, owners) if o == r) for r in range(world)]
    step_ms = dt / steps * 1e3
    # ---- result check: the cloud, the MatchSet and the BA sums of this rank, across ranks and against the world-1 golden
    sums = res["ba_sums"].detach().float().cpu().numpy() if "ba_sums" in res else np.zeros(0, np.float32)
    check = {"cloud": checksum64(res["points"]), "multi_matches": checksum64(res["matches"]), "keypoints": checksum64(res["keypoints"]),
             "ba_sums": checksum64(sums)}
    all_checks = [check]
    if dist is not None:
        all_checks = [None] * world
        dist.all_gather_object(all_checks, check)
    same = all(c == all_checks[0] for c in all_checks)
    key = "views%d_size%d" % (views, size)
    golden = json.load(open(NVIEW_GOLDEN)) if os.path.exists(NVIEW_GOLDEN) else {}
    g = golden.get(key)
    exact = ("cloud", "multi_matches", "keypoints")  # the BA sums are float sums in a sharding-dependent order: compared by value
    result_check = {"checksums": {k: "%016x" % v for k, v in check.items()}, "equal_across_ranks": same,
                    "golden": "tests/golden/nview_checksums.json[%s]" % key if g else None,
                    "equal_to_world1": None if not g else all(("%016x" % check[k]) == g[k] for k in exact),
                    "ba_sums_max_rel_dev_from_world1": None if not g or not len(sums) else float(np.max(np.abs(sums[:len(g["ba_sums_head"])] - np.array(g["ba_sums_head"], np.float32)) /
                                                                                                   np.maximum(np.abs(np.array(g["ba_sums_head"], np.float32)), 1e-30))),
                    "bundles": int(len(res["matches"])), "points": int(res["points"].shape[0])}
    if rank == 0 and world == 1 and os.environ.get("SSRLCV_WRITE_NVIEW_GOLDEN"):
        golden[key] = dict({k: "%016x" % check[k] for k in check}, ba_sums_head=[float(x) for x in sums[:8]], bundles=result_check["bundles"],
                           features_per_image=nf, written_by="bench.py --gpus 1 (SSRLCV_WRITE_NVIEW_GOLDEN=1) @ %s" % git_head())
        out_path = os.environ["SSRLCV_WRITE_NVIEW_GOLDEN"]   # "1": in place; anything else: that path (gpurun only brings gpurun_out/ back)
        with open(NVIEW_GOLDEN if out_path == "1" else out_path, "w") as f:
            json.dump(golden, f, indent=1, sort_keys=True)
    if not same or result_check["equal_to_world1"] is False:
        result_check["FAILED"] = "ranks disagree" if not same else "differs from the world-1 result"
    # stages every rank repeats on the whole problem (the rest shrinks with the rank count): the serial fraction of the flow
    replicated = sum(mine.get(k, 0.0) for k in ("merge", "filter"))
    wire = {"exchange_features_bytes": int(sum(f.numel() for f in res["features"])),
            "exchange_pairs_bytes": int(sum(p.numel() for p in res["pairs"])),
            "cloud_all_gather_bytes": int(res["points"].shape[0] * 12),
            "ba_all_reduce_bytes": 612 * 4,
            "mode": sd.exchange_mode(),
            "note": "payload every rank ends up holding, per step; SSRLCV_EXCHANGE=bcast (default): each rank sends its own share once "
                    "at its exact size (one grouped broadcast per rank, dist._gather_segments); =allgather: one all-gather padded to the "
                    "largest share; world 1 moves nothing"}
    comm = {"backend": dist.get_backend() if dist is not None else None, "rccl_world": dist.get_world_size() if dist is not None else 1}
    if dist is not None and dist.get_backend() == "nccl":
        try:
            comm["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
    return {"metric": "Mpix/s N-view reconstruction (SIFT + exhaustive orbit match + merge + N-view triangulate + BA sweep)",
            "value": views * size * size * steps / dt / 1e6, "unit": "Mpix/s", "n_gpus": world, "steps": steps,
            "ms_per_step": step_ms, "ms_per_step_stats_rank0": step_stats, "stall": bool(step_stats["max"] > 2.0 * step_stats["median"]),
            "scaling": "strong", "views": views, "pairs": len(pairs),
            "workload": "%d-view %dx%d scene, image/pair shard over %d GPU(s): all-gather of features, pairs balanced by "
                        "nq*nt, all-gather of uint2_pair arrays, replicated merge on the device (ssrlcv_hip_merge_matches), bundle-range "
                        "triangulation + all-gather of the cloud, 612-point BA error sweep + all-reduce" % (views, size, size, world),
            "stage_ms_per_step_rank0": mine, "stage_ms_stats_rank0": mine_stats,
            "stage_ms_per_step_over_ranks": spread,
            "replicated_stage_ms": replicated, "replicated_share_of_step": replicated / step_stats["median"] if step_stats["median"] else None,
            "pair_cost_share_per_rank": [l / max(sum(cost), 1.0) for l in load],
            "pair_balance_max_over_mean": max(load) / (sum(load) / world) if sum(load) else None,
            "images_per_rank": [sum(1 for v in range(views) if sd.image_owner(v, world) == r) for r in range(world)],
            "wire": wire, "comm": comm, "result_check": result_check,
            "multi_matches": int(len(res["matches"])), "points": int(res["points"].shape[0]),
            "ba_bundles": int(res.get("ba_bundles", 0)),
            "features_per_image": nf,
            "_inputs": (imgs, cams, res["points"].cpu().numpy())}   # for the class-API flow leg (popped before printing)


def run_pushbroom(args, torch, dist, world, rank, dev, size, views):
    """BASELINE config[4]: `views` pushbroom strips of size^2 -> SIFT per strip (one plan reused: a workspace is 22 GB at
    8192^2) -> the exhaustive brute-force + seed-ratio matching of all strip pairs -> merge -> generatePushbroomBundle
    (src/PointCloudFactory.cu:875-903, :4201-4283) + N-view triangulation -> statistical filters on the device.  One warm-up
    step, one timed step (the match stage of 28 pairs of 2-3 million features each is ~20 s on one GPU), per-stage times,
    and the scale-space stage of one strip alone against the HBM roofline (`roofline_8192`)."""
    import helpers as H
    import scene
    from ssrlcv_amd import capi, pipeline
    imgs, pbs, _, _ = scene.pushbroom_views(views, size, device=dev)
    seed, _ = H.load_seed_features()
    ws = pipeline.Workspace()
    filters = [("statistical", 3.0, 0.1)] * 4

    def step():
        return pipeline.reconstruct(imgs, None, seed_features=seed, mode=0, pushbroom=pbs, ws=ws, filters=filters)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    step()
    ws.times.clear()
    barrier()
    t0 = time.perf_counter()
    res = step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if dist is None or dist.get_backend() != "gloo" else "cpu")
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    stages = {k: v * 1e3 for k, v in ws.times.items()}
    nf = [int(f.numel() // 152) for f in res["features"]]
    check = {"cloud": "%016x" % checksum64(res["points"]), "multi_matches": "%016x" % checksum64(res["matches"])}
    out = {"metric": "Mpix/s pushbroom N-view reconstruction (SIFT + exhaustive brute-force match + merge + pushbroom bundles + N-view triangulate + filters)",
           "value": views * size * size / dt / 1e6, "unit": "Mpix/s", "n_gpus": world, "steps": 1, "ms_per_step": dt * 1e3, "scaling": "strong",
           "workload": "config[4]: %d pushbroom strips of %dx%d (tools/scene.py pushbroom_views), image/pair shard over %d GPU(s); %d pairs" %
                       (views, size, size, world, views * (views - 1) // 2),
           "stage_ms": stages, "features_per_strip": nf, "multi_matches_unfiltered": int(res["matches_unfiltered"]),
           "multi_matches": int(len(res["matches"])), "points": int(res["points"].shape[0]), "filters": "4 x statistical (3 sigma, 10 %) on the device",
           "checksums_rank0": check}
    del res
    if rank == 0:
        # the scale-space stage of ONE strip alone (ssrlcv_hip_sift_build_dog, HIP events on the launching stream)
        plan = ws.plan(size, size, 0)
        for _ in range(2):
            plan.build_dog(imgs[0])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            plan.build_dog(imgs[0])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        b_pyr = 362.25 * size * size
        rl = {"bound": "hbm", "achieved": b_pyr / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b_pyr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
              "ms_per_image": ms, "algorithmic_bytes": b_pyr, "traffic": None,
              "kernel": "scale-space stage (ssrlcv_hip_sift_build_dog) of one %dx%d strip alone; same definition as `roofline.stage_alone`" % (size, size)}
        tpath = os.path.join(ROOT, "profiles", "r06_8192_pyramid_traffic.json")
        if size == 8192 and os.path.exists(tpath):
            tj = json.load(open(tpath))
            rl["traffic"], rl["traffic_source"] = tj["pyramid_stage_bytes_per_image"], "profiles/r06_8192_pyramid_traffic.json @ %s" % tj.get("commit")
        out["roofline_8192" if size == 8192 else "roofline_strip"] = rl
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=4096, help="image edge (config[2]: 2-view 4096x4096)")
    ap.add_argument("--images", type=int, default=2, help="images per rank per step (one pair)")
    ap.add_argument("--match-n", type=int, default=1 << 18, help="Nq = Nt of the stand-alone matcher measurement")
    ap.add_argument("--match-iters", type=int, default=3)
    ap.add_argument("--cpu-size", type=int, default=1024, help="edge of the CPU baseline's config[1] pair")
    ap.add_argument("--cpu-big", type=int, default=2048, help="edge of the CPU baseline's single bigger view")
    ap.add_argument("--cpu-reps", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-matcher", action="store_true")
    ap.add_argument("--no-nview", action="store_true")
    ap.add_argument("--no-class-api", action="store_true")
    ap.add_argument("--nview-views", type=int, default=0,
                    help="views of the N-view leg; 0 = max(4, world): config[3]'s four views (6 pairs) up to four ranks, eight views "
                         "(28 pairs, src/MatchFactory.cu:907-1028 order) on eight, so that no rank is left without an image or a pair")
    ap.add_argument("--nview-size", type=int, default=4096, help="edge of the N-view leg's images (config[3]: 4-view 4096x4096)")
    ap.add_argument("--nview-steps", type=int, default=6, help="timed steps of the N-view leg (median / min / max are reported; two warm-up steps)")
    ap.add_argument("--no-pushbroom", action="store_true", help="skip the config[4] leg (eight 8192^2 pushbroom strips: ~1 minute)")
    ap.add_argument("--pushbroom-size", type=int, default=8192)
    ap.add_argument("--pushbroom-views", type=int, default=0, help="0 = max(8, world)")
    ap.add_argument("--noise-input", action="store_true", help="round-1 input: multi-scale noise instead of the scene generator")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # SSRLCV_BENCH_BACKEND=gloo: rehearsal of the N > 1 legs on a box with fewer GPUs than ranks (every collective staged
        # through the host, ranks share the cards); the driver's runs use RCCL, one rank per GPU
        if os.environ.get("SSRLCV_BENCH_BACKEND", "nccl") == "gloo":
            local_rank %= max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    from ssrlcv_amd import capi, _lib  # raises if the HIP library is missing: no CPU fallback (default: the release build)
    import scene

    W = H_ = args.size
    dev = torch.device("cuda", torch.cuda.current_device())
    if args.noise_input:
        imgs = synth_images(args.images, W, H_, seed=rank, device=dev)
        workload = "multi-scale noise"
    else:
        imgs, _, _, _ = scene.pinhole_views(args.images, W, device=dev, seed=scene.SEED + 7919 * rank)
        workload = "tools/scene.py pinhole views (PCG32 terrain + texture through Image::Camera)"
    plans = [capi.SiftPlan(W, H_) for _ in range(args.images)]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(ev=None):
        """One pass of the hot path over this rank's images: ssrlcv_hip_sift_extract (both stages in one call, as
        SIFT_FeatureFactory::generateFeatures makes it).  ev[i] = (before, between the stages, after): the middle event is
        recorded by the library itself on the launching stream (ssrlcv_sift_plan_set_stage_event)."""
        for i, (p, im) in enumerate(zip(plans, imgs)):
            if ev is not None:
                p.set_stage_event(ev[i][1])
                ev[i][0].record()
            p.extract(im)
            if ev is not None:
                ev[i][2].record()

    for _ in range(args.warmup):
        step()
    barrier()
    events = [[[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in plans] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if dist is None or dist.get_backend() != "gloo" else "cpu")
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    for p in plans:
        p.set_stage_event(None)
    nfeat = [p.count() for p in plans]  # raises if a key-point list overflowed its capacity (truncated result)
    pyr_ms = float(np.mean([e[0].elapsed_time(e[1]) for st in events for e in st]))
    desc_ms = float(np.mean([e[1].elapsed_time(e[2]) for st in events for e in st]))
    # The scale-space stage ALONE (ssrlcv_hip_sift_build_dog as its own call, a few back-to-back launches per image, HIP events
    # on the launching stream; outside the timed region): inside the fused extract octave 0's list chain of the key-point
    # stage runs beside the stage's tail since round 5, so the stage-boundary event there sees a longer scale-space stage.
    alone = []
    for p, im in zip(plans, imgs):
        for _ in range(2):
            p.build_dog(im)
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(5):
            p.build_dog(im)
        a1.record()
        torch.cuda.synchronize()
        alone.append(a0.elapsed_time(a1) / 5)
    alone_ms = float(np.mean(alone))
    del plans

    line = None
    if rank == 0:
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r06_pyramid_traffic.json" if os.path.exists(os.path.join(ROOT, "profiles", "r06_pyramid_traffic.json")) else "r05_pyramid_traffic.json")
        if W == 4096 and H_ == 4096 and os.path.exists(tpath):
            # HBM bytes of the pyramid stage per image from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the stage
            # benchmark (collected off-line: counters cannot be read from inside the timed run)
            tj = json.load(open(tpath))
            traffic, traffic_src = tj["pyramid_stage_bytes_per_image"], "profiles/%s @ %s" % (os.path.basename(tpath), tj.get("commit"))
        pixels_per_step = world * args.images * W * H_
        value = pixels_per_step * args.steps / dt / 1e6
        # Algorithmic bytes of the stage.  `frac` is priced on SURVEY.md 8(d)'s / BASELINE.md section 4's own figure,
        # B_pyr = 362.25 W H (S1-S7: u8 once, 6 levels per octave written and read once, 5 DoG levels written) -- the
        # definition of rounds 1-2 and of the 0.70 target.  The stage also runs S8 (findExtrema) in the same pass since
        # round 3; SURVEY's figure for it is 20 B read + 1 B written per scale-space pixel = 111.56 W H, reported as the
        # secondary, separately labelled `frac_s1_s8` (round 3 called THAT figure `frac`: 0.58 there is 0.44 here).
        b_pyr = 362.25 * W * H_
        b_ext = 21.0 * 5.3125 * W * H_
        achieved = b_pyr / (pyr_ms * 1e-3) / 1e9
        line = {
            "metric": "Mpix/s SIFT extract (+ Mmatches/s 128-D brute-force, see `matcher`)",
            "value": value, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "commit": git_head(),
            "config": {"workload": "2-view %dx%d pair per GPU (%s): SIFT_FeatureFactory::generateFeatures (sparse DoG "
                                   "path) on each image, pixels resident in HBM" % (W, H_, workload),
                       "images_per_gpu": args.images, "features_per_image": nfeat, "parallelism": "image-pair shard",
                       "library": "%s build (%s)" % (_lib.flavour(), os.path.basename(_lib.LIB_PATH))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "scale-space stage = ssrlcv_hip_sift_build_dog: S1-S8 (u8 upsample in the first level's loader, "
                                   "24 gaussian levels -- the level-3 launches also write the 2x2 bin --, then per octave ONE pass "
                                   "that forms the 5 DoG levels in registers, reduces their min / max and finds the extrema: the "
                                   "DoG levels are never written); algorithmic bytes = B_pyr = 362.25*W*H per image (SURVEY 8d, S1-S7); "
                                   "the stage is timed inside the fused ssrlcv_hip_sift_extract: HIP events on the launching stream, the one between "
                                   "the two stages recorded by the library itself (ssrlcv_sift_plan_set_stage_event); since round 5 octave 0's key-point list chain "
                                   "runs beside the end of this stage (see stage_alone)",
                         "algorithmic_bytes": b_pyr,
                         "frac_s1_s8": (b_pyr + b_ext) / (pyr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_s1_s8": b_pyr + b_ext,
                         "frac_pyramid_only": achieved / HBM_PEAK_GBS,
                         "ms_per_image": pyr_ms,
                         "stage_alone": {"ms_per_image": alone_ms, "frac": b_pyr / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                         "note": "ssrlcv_hip_sift_build_dog as its own call (5 back-to-back launches per image, HIP events, "
                                                 "outside the timed region).  `frac` / `ms_per_image` above are measured inside the timed "
                                                 "fused extract, where octave 0's list chain of the key-point stage runs beside the "
                                                 "stage's tail (round 5: the step gains what the stage-boundary event loses)"}},
            "describe": describe_roofline(desc_ms, int(np.mean(nfeat)), W),
        }
        fpath = os.path.join(ROOT, "profiles", "r06_pyramid_floor.json")
        if W == 4096 and H_ == 4096 and os.path.exists(fpath):
            # per-launch floor table of the stage (tools/collect_floor.sh + tools/pyramid_floor.py, serial run under rocprofv3):
            # solo time, counter traffic, copy floor (traffic / 5.3 TB/s), issue floor, gap -- summarised here, rows in the file
            fj = json.load(open(fpath))
            rows = [r for r in fj["launches"] if "gap_us" in r]
            line["roofline"]["floor"] = {
                "source": "profiles/r06_pyramid_floor.json @ %s" % fj.get("commit"),
                "serial_sum_us": fj["serial_sum_us"], "copy_floor_sum_us": fj["copy_floor_sum_us"], "floor_sum_us": fj["floor_sum_us"],
                "gap_sum_us": fj["gap_sum_us"], "traffic_bytes": fj["traffic_bytes"], "algorithmic_bytes_as_built": fj["algorithmic_bytes"],
                "octave0_levels_0_3": fj["octave0_levels_0_3"],
                "largest_gaps": [{"kernel": r["kernel"], "octave": r.get("octave"), "level": r.get("level"), "solo_us": r["solo_us"],
                                  "copy_floor_us": r.get("copy_floor_us"), "issue_floor_us": r.get("issue_floor_us"), "gap_us": r["gap_us"]}
                                 for r in sorted(rows, key=lambda r: -r["gap_us"])[:6]]}
    img0 = imgs[0]
    del imgs[1:]
    torch.cuda.empty_cache()
    if not args.no_nview:
        # every rank enters the N-view leg together (the barrier inside run_nview): the rank-0-only legs come after it,
        # so that its stage times are not polluted by rank skew
        nviews = args.nview_views if args.nview_views > 0 else max(4, world)
        nv = run_nview(args, torch, dist, capi, world, rank, dev, nviews, args.nview_size, max(args.nview_steps, 1), 2)
        nv_inputs = nv.pop("_inputs")
        if rank == 0:
            line["nview"] = nv
    if not args.no_pushbroom:
        # config[4]; every rank enters together (its exchanges are collectives), before the rank-0-only legs
        torch.cuda.empty_cache()
        pb = run_pushbroom(args, torch, dist, world, rank, dev, args.pushbroom_size, args.pushbroom_views if args.pushbroom_views > 0 else max(8, world))
        if rank == 0:
            line["pushbroom8"] = pb
        torch.cuda.empty_cache()
    if rank == 0:
        if not args.no_class_api:
            line["class_api"] = class_api_leg(img0, W, line["value"] / world)
            if not args.no_nview and world == 1:
                line["class_api"]["flow"] = class_api_flow_leg(nv_inputs[0], nv_inputs[1], args.nview_size, nv["ms_per_step_stats_rank0"]["median"], nv_inputs[2])
        if not args.no_matcher:
            line["matcher"] = bench_matcher(capi, torch, args.match_n, args.match_n, args.match_iters)
            line["matcher_f16"] = bench_matcher(capi, torch, args.match_n, args.match_n, args.match_iters, "f16")
            line["matcher_f16"]["same_output_as_int8"] = line["matcher_f16"]["checksum"] == line["matcher"]["checksum"]
            line["matcher_epipolar"] = bench_matcher_epipolar(capi, torch, args.match_n, W, args.match_iters)
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is a single-GPU-run figure
            line["cpu_baseline"] = cpu_baseline(args.cpu_size, args.cpu_big, args.cpu_reps)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
