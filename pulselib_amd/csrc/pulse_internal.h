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

// stoprule.hip: a check point's partial done-counts (one uint32 per wavefront / workgroup) are written in stream order
// into the slot `claim` hands out.  Their sum is published to the host by the NEXT launch on that stream: `claim` also
// returns the previous check point's counts as a `carry`, which that launch sums and publishes in its first
// workgroup, ahead of that workgroup's tables (no event, no second stream, no kernel of its own on the steps' stream).
struct StopRuleCarry {
    const uint32_t* partials;     // nullptr: nothing to carry
    int n;
    long long* host;              // {local, global, seq} in coherent pinned memory
    long long seq;
};
int stoprule_claim(PulseStopRule* h, int n_partials, uint32_t** partials_out, StopRuleCarry* carry);
int stoprule_commit(PulseStopRule* h, int n_partials, ihipStream_t* stream);

}  // namespace pulse
