// abi.hip -- error reporting and version of libpulse_hip.so (include/pulse_env.h).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "pulse_internal.h"

namespace {
thread_local char g_err[512] = "";
}

namespace pulse {

int fail(int code, const char* msg) {
    std::snprintf(g_err, sizeof g_err, "%s", msg ? msg : "");
    return code;
}

int fail_hip(int hip_error, const char* what) {
    const hipError_t e = (hipError_t)hip_error;
    std::snprintf(g_err, sizeof g_err, "%s: HIP error %d (%s)", what ? what : "hip", hip_error, hipGetErrorString(e));
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorNotInitialized) ? PULSE_ENODEVICE : PULSE_ELAUNCH;
}

}  // namespace pulse

extern "C" {

int pulse_version(void) { return PULSE_ABI_VERSION; }

const char* pulse_last_error(void) { return g_err; }

}  // extern "C"
