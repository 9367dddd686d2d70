"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/ssrlcv_hip.h declares; the POD layouts match the reference's (sizes measured in SURVEY.md 8a)."""
import ctypes
import os
import re

import numpy as np

import helpers as H

ROOT = H.ROOT


def test_library_exports_every_declared_symbol():
    from ssrlcv_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "ssrlcv_hip.h")).read()
    declared = set(re.findall(r"\b(ssrlcv_(?:hip_)?[A-Za-z0-9_]+)\s*\(", header))
    declared -= {"ssrlcv_stream_t"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "libssrlcv_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTED)
    assert b"gfx950" in lib.ssrlcv_hip_version()
    assert lib.ssrlcv_hip_status_string(0) == b"ok"
    assert lib.ssrlcv_hip_status_string(-3) == b"workspace too small"


def test_missing_extension_fails_loudly(tmp_path, monkeypatch):
    from ssrlcv_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.load()
    except _lib.HipExtensionMissing as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must raise when the HIP library is absent")


def test_host_only_entry_points_match_oracle(oracle_lib):
    """Host arithmetic exported by the C ABI (no GPU needed): Gaussian taps and the projection matrix."""
    from ssrlcv_amd import _lib
    lib = _lib.load()
    for sigma, pw in ((0.70710678, 0.5), (1.0, 0.5), (2.0, 0.5), (4.0, 0.5), (5.656854, 2.0), (16.0, 4.0)):
        w = np.zeros(129, np.float32)
        taps = lib.ssrlcv_gauss_kernel_host(ctypes.c_float(sigma), ctypes.c_float(pw), H.P(w))
        otaps, ow = H.oracle_gauss_kernel(oracle_lib, sigma, pw)
        assert taps == otaps and taps % 2 == 1
        assert np.array_equal(w[:taps], ow)
    v = H.load_view("Pipeline3View")
    for i in range(3):
        cam = v["cameras"][i:i + 1]
        out = np.zeros((3, 4), np.float32)
        lib.ssrlcv_projection_matrix_host(H.P(cam), H.P(out))
        assert np.array_equal(out, H.oracle_projection(oracle_lib, cam))


def test_sift_plan_layout_is_host_only():
    from ssrlcv_amd import _lib
    lib = _lib.load()

    class SiftParams(ctypes.Structure):
        _fields_ = [("maxOrientations", ctypes.c_uint32), ("orientationThreshold", ctypes.c_float),
                    ("orientationContribWidth", ctypes.c_float), ("descriptorContribWidth", ctypes.c_float),
                    ("maxKeyPointsPerOctave", ctypes.c_uint32)]
    p = SiftParams(2, 0.8, 1.5, 6.0, 0)
    plan = ctypes.c_void_p()
    assert lib.ssrlcv_sift_plan_create(ctypes.c_uint32(1024), ctypes.c_uint32(1024), ctypes.byref(p),
                                       ctypes.byref(plan)) == 0
    nbytes = lib.ssrlcv_sift_plan_workspace_bytes(plan)
    # sum P = 5.3125 W H pixels; the 6 gaussian levels + 3 polar tables (8 bytes per pixel) dominate (no DoG levels)
    assert 5.3125 * 1024 * 1024 * (6 * 4 + 3 * 8) < nbytes < 420e6
    lib.ssrlcv_sift_plan_destroy(plan)
    # sizes that need makeBinnable padding (S3) get the reference's border: 1001 x 1024 has an odd side, so the upsampled
    # 2002 x 2048 image is padded to multiples of 32 (2016 x 2048) and the workspace grows with it
    assert lib.ssrlcv_sift_plan_create(ctypes.c_uint32(1001), ctypes.c_uint32(1024), ctypes.byref(p),
                                       ctypes.byref(plan)) == 0
    nb_odd = lib.ssrlcv_sift_plan_workspace_bytes(plan)
    lib.ssrlcv_sift_plan_destroy(plan)
    assert nbytes * 0.97 < nb_odd < nbytes * 1.05  # (mode 2 also keeps the un-padded upsampled image)
    # even sizes are padded to multiples of 8 before the upsample: 1002 -> 1008
    assert lib.ssrlcv_sift_plan_create(ctypes.c_uint32(1002), ctypes.c_uint32(1024), ctypes.byref(p),
                                       ctypes.byref(plan)) == 0
    lib.ssrlcv_sift_plan_destroy(plan)
    # small and thin images are taken like upstream (src/FeatureFactory.cu:364-376) ...
    for (w, h) in ((200, 1024), (64, 64)):
        assert lib.ssrlcv_sift_plan_create(ctypes.c_uint32(w), ctypes.c_uint32(h), ctypes.byref(p), ctypes.byref(plan)) == 0
        lib.ssrlcv_sift_plan_destroy(plan)
    # ... down to 64 pixels: below that the reference's 65-tap mirror indexes outside its smallest octave (undefined
    # upstream), which is refused (-4 = SSRLCV_ERR_UNSUPPORTED), not mishandled
    assert lib.ssrlcv_sift_plan_create(ctypes.c_uint32(48), ctypes.c_uint32(1024), ctypes.byref(p),
                                       ctypes.byref(plan)) == -4
    # contribution widths whose sampling windows would outgrow the kernels' 16-bit window indexing are refused too
    wide = SiftParams(2, 0.8, 1.5, 64.0, 0)
    assert lib.ssrlcv_sift_plan_create(ctypes.c_uint32(1024), ctypes.c_uint32(1024), ctypes.byref(wide),
                                       ctypes.byref(plan)) == -4


def test_release_build_has_no_developer_switches():
    """`make release` (csrc/Makefile, -DSSRLCV_RELEASE): the same exports, and none of the SSRLCV_* switch names of the
    developer build survive in it -- svdev::env() (csrc/dev_switch.h) is a constant nullptr there, so a caller's environment
    cannot choose a code path of the drop-in library."""
    import subprocess
    from ssrlcv_amd import _lib
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "ssrlcv_amd", "csrc"), "release"])
    rel = os.path.join(ROOT, "ssrlcv_amd", "libssrlcv_hip_release.so")
    lib = ctypes.CDLL(rel)
    for name in _lib.EXPORTED:
        assert hasattr(lib, name), name
    blob = open(rel, "rb").read()
    dev = open(os.path.join(ROOT, "ssrlcv_amd", "libssrlcv_hip.so"), "rb").read()
    for name in (b"SSRLCV_GAUSS_VALU", b"SSRLCV_DOGX_NPX", b"SSRLCV_MATCH_F16", b"SSRLCV_SIFT_SERIAL", b"SSRLCV_MERGE_THREADS"):
        assert name in dev and name not in blob, name
    # no raw getenv of a switch is left in the hot-path sources
    for f in ("pyramid.hip", "keypoints.hip", "matcher.hip", "merge.hip", "filter.hip", "pointcloud.hip", "pose.hip", "host_merge.cpp"):
        src = open(os.path.join(ROOT, "ssrlcv_amd", "csrc", f)).read()
        assert not re.search(r'(?<![A-Za-z_:])getenv\(', src), f


def test_the_default_library_is_the_release_build():
    """ssrlcv_amd/_lib.py loads libssrlcv_hip_release.so unless told otherwise, so bench.py, smoke() and the parity tests run
    the library INTEGRATION.md tells a deployer to link; tests/helpers.py dev_env() is how a child process gets the
    developer build (the formulation-by-formulation tests), SSRLCV_HIP_LIB any other one."""
    import subprocess
    import sys
    import helpers as H
    probe = "import sys; sys.path.insert(0, %r); from ssrlcv_amd import _lib; print(_lib.flavour(), _lib.LIB_PATH)" % ROOT
    env = dict(os.environ)
    env.pop("SSRLCV_HIP_LIB", None)
    env.pop("SSRLCV_DEV_BUILD", None)
    out = subprocess.check_output([sys.executable, "-c", probe], env=env, text=True).split()
    assert out[0] == "release" and out[1].endswith("libssrlcv_hip_release.so")
    out = subprocess.check_output([sys.executable, "-c", probe], env=H.dev_env(SSRLCV_GAUSS_VALU="1"), text=True).split()
    assert out[0] == "developer" and out[1].endswith("libssrlcv_hip.so")
    out = subprocess.check_output([sys.executable, "-c", probe], env=dict(env, SSRLCV_HIP_LIB="/tmp/other.so"), text=True).split()
    assert out == ["/tmp/other.so", "/tmp/other.so"]


def test_matcher_kernels_register_budget():
    """The matcher kernels are built for fixed occupancies (__launch_bounds__): the compiler's resource report must stay
    what the timings were taken with -- no scratch in either.  Brute force: four query tiles, two waves per SIMD.
    Band-culled: one query tile, five waves per SIMD (six spilled 4 registers in round 3 and 16 with round 4's third box
    level, which cost more than the sixth wave brings)."""
    import subprocess
    csrc = os.path.join(ROOT, "ssrlcv_amd", "csrc")
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                          "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                          "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(csrc, "matcher.hip"), "-o", os.devnull],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    usage, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and name:
            usage[name][m.group(1).strip()] = int(m.group(2))
    brute = [v for k, v in usage.items() if "k_match_i8ILb0ELi4ELb0E" in k]
    seed = [v for k, v in usage.items() if "k_match_i8ILb0ELi4ELb1E" in k]  # the min-only variant of getSeedDistances
    band = [v for k, v in usage.items() if "k_match_i8ILb1ELi1ELb0E" in k]
    assert len(brute) == 1 and len(band) == 1 and len(seed) == 1, sorted(usage)
    assert brute[0]["ScratchSize [bytes/lane]"] == 0 and brute[0]["VGPRs"] <= 256 and brute[0]["Occupancy [waves/SIMD]"] == 2
    assert seed[0]["ScratchSize [bytes/lane]"] == 0 and seed[0]["Occupancy [waves/SIMD]"] == 2
    assert band[0]["ScratchSize [bytes/lane]"] == 0 and band[0]["Occupancy [waves/SIMD]"] == 5, band[0]


_WRONG_ABI_STUB = r"""
int ssrlcv_hip_abi_version(void) { return 999; }
const char* ssrlcv_hip_version(void) { return "stub (abi 999)"; }
const char* ssrlcv_hip_status_string(int s) { (void)s; return "stub"; }
"""


def _wrong_abi_library(tmp_path):
    import subprocess
    src = tmp_path / "stub.c"
    src.write_text(_WRONG_ABI_STUB)
    so = tmp_path / "libssrlcv_hip_release.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    return so


def test_library_of_another_abi_version_is_refused(tmp_path):
    """include/ssrlcv_hip.h SSRLCV_HIP_ABI_VERSION is bumped on every signature / layout change.  Both binders compare
    it with what the library they loaded reports before the first real call: the Python loader raises HipAbiMismatch, the
    C++ mirror logs and exits(-1) (host/Memory.hpp requireAbi, reached from the first allocation)."""
    import subprocess
    import sys
    from ssrlcv_amd import _lib
    header = open(os.path.join(ROOT, "include", "ssrlcv_hip.h")).read()
    declared = int(re.search(r"#define SSRLCV_HIP_ABI_VERSION (\d+)", header).group(1))
    assert declared == _lib.ABI_VERSION == _lib.load().ssrlcv_hip_abi_version()
    assert ("abi %d" % declared).encode() in _lib.load().ssrlcv_hip_version()
    so = _wrong_abi_library(tmp_path)
    probe = ("import sys; sys.path.insert(0, %r)\nfrom ssrlcv_amd import _lib\n"
             "try:\n    _lib.load()\nexcept _lib.HipAbiMismatch as e:\n    print('refused:', e)\n" % ROOT)
    out = subprocess.check_output([sys.executable, "-c", probe], env=dict(os.environ, SSRLCV_HIP_LIB=str(so)), text=True)
    assert "refused:" in out and "ABI version 999" in out and "version %d" % declared in out
    # the C++ mirror: a program that only allocates, linked against the stub
    cpp = tmp_path / "abi_probe.cpp"
    cpp.write_text('#include "Unity.hpp"\nint main() { ssrlcv::ptr::device<float> d(16); return d.get() ? 0 : 1; }\n')
    exe = tmp_path / "abi_probe"
    host = os.path.join(ROOT, "ssrlcv_amd", "host")
    subprocess.check_call(["g++", "-O0", "-std=c++14", "-I" + os.path.join(ROOT, "include"), "-I" + host, "-o", str(exe), str(cpp),
                           "-L" + str(tmp_path), "-lssrlcv_hip_release", "-Wl,-rpath," + str(tmp_path),
                           "-Wl,--unresolved-symbols=ignore-all"])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 255, (r.returncode, r.stderr)
    assert "ABI version 999" in r.stderr and "version %d" % declared in r.stderr
