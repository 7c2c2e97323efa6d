// pulse_internal.h -- shared by the translation units of libpulse_hip.so (not part of the ABI).
#pragma once
#include <cstdint>

#include "../../include/pulse_env.h"

struct PulseStopRule;
struct ihipStream_t;

namespace pulse {

// Records a thread-local error message and returns `code` (so callers can `return fail(...)`).
int fail(int code, const char* msg);
int fail_hip(int hip_error, const char* what);

// stoprule.hip: a chunk's partial done-counts (one uint32 per wavefront / workgroup) are written in stream order
// into the slot `claim` hands out; `commit` sums, all-reduces and copies them to the host on the rule's side stream.
int stoprule_claim(PulseStopRule* h, int n_partials, uint32_t** partials_out);
int stoprule_commit(PulseStopRule* h, int n_partials, ihipStream_t* stream);

}  // namespace pulse
