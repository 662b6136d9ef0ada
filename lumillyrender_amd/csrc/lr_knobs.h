// lr_knobs.h -- diagnostic knobs of liblumilly_hip.so.
//
// Environment variables that change WHAT the library runs (pipeline, banding, chunk schedule, stack placement, builders ...) are
// measurement and test instruments, not product surface: a host that links the library must not have a stale shell
// variable change its renders.  They exist only in a build with -DLR_DIAG_KNOBS (`make -C lumillyrender_amd/csrc knobs` ->
// liblumilly_hip_knobs.so, loaded by the tests that need one and by tools/ through LR_HIP_LIB); the product library compiles
// lr_knob() to "unset" and reads none of them.  lr_build_info() says which of the two a process has loaded (knobs=on|off).
// LR_DEBUG (prints only) is not a knob.
#pragma once
#include <cstdlib>

namespace lr {
#ifdef LR_DIAG_KNOBS
inline const char* lr_knob(const char* name) { return std::getenv(name); }
constexpr const char* kKnobsState = "on";
#else
inline const char* lr_knob(const char*) { return nullptr; }
constexpr const char* kKnobsState = "off";
#endif
}  // namespace lr
