// ssrlcv_amd/csrc/dev_switch.h -- developer switches of the library.
//
// The kernels exist in several bit-identical formulations (VALU / MFMA / register-marching Gaussians, pixels per lane of
// the DoG pass, schedules ...); which one runs is decided by size and radius, and -- in a DEVELOPER build, the default
// of csrc/Makefile -- can be forced through SSRLCV_* environment variables read once per process, which is how the
// parity tests hold every formulation to the oracle and how the A/B timings of DESIGN.md were taken.  A drop-in library
// must not let its caller's environment choose its code path: `make -C ssrlcv_amd/csrc release` builds
// libssrlcv_hip_release.so with -DSSRLCV_RELEASE, where svdev::env() is a constant nullptr and every switch is compiled
// down to its default.
#pragma once
#include <stdlib.h>

// Instrumented builds (results unchanged, timings not): s_memtime stamps of the Gaussian kernels (-DSSRLCV_STAMPS, the tools/
// lab programs) and the walk counters of the band-culled matcher (-DSSRLCV_MATCH_STATS, `make instrumented`).  Neither may
// end up in libssrlcv_hip.so / libssrlcv_hip_release.so: the `all` and `release` targets stop here.  (The timing labs of
// rounds 4-5 whose RESULTS were invalid -- matcher without epilogue / operand reads, gathers folded into 32 KB, one byte
// load of eight in the upsampling loader -- served their measurements, profiles/r05_matcher_lab_pmc.txt,
// r05_sampling_gather_lab.txt, r05_kernel_ab.txt section 6, and were removed from the sources in round 6: git history.)
// SSRLCV_TIMING_VARIANT (tools/ lab programs only): builds that leave work OUT to time what remains -- results invalid.
#if (defined(SSRLCV_STAMPS) || defined(SSRLCV_MATCH_STATS) || defined(SSRLCV_TIMING_VARIANT)) && !defined(SSRLCV_INSTRUMENTED_BUILD)
#error "SSRLCV_STAMPS / SSRLCV_MATCH_STATS only build through `make instrumented` or the tools/ lab programs (-DSSRLCV_INSTRUMENTED_BUILD)"
#endif
#if defined(SSRLCV_LAB) || defined(SSRLCV_MATCH_LAB) || defined(SSRLCV_LAB_LOCAL_GATHER) || defined(SSRLCV_LAB_UPS_LOADS)
#error "the timing-lab variants (results invalid) were removed in round 6; see git history before 2026-10-05"
#endif

namespace svdev {
inline const char* env(const char* name) {
#ifdef SSRLCV_RELEASE
  (void)name;
  return nullptr;
#else
  return getenv(name);
#endif
}
}  // namespace svdev
