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
