// pulse_internal.h -- shared by the translation units of libpulse_hip.so (not part of the ABI).
#pragma once
#include <cstdint>

#include "../../include/pulse_env.h"

namespace pulse {

// Records a thread-local error message and returns `code` (so callers can `return fail(...)`).
int fail(int code, const char* msg);
int fail_hip(int hip_error, const char* what);

}  // namespace pulse
