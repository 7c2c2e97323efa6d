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

// envs.hip: the current device's row table of the packed 2048 move (tfe_device.h), built on first use
int tfe_row_lut(const uint32_t** out);

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

// Paired launches (lag 1 only): ONE launch runs up to two check intervals ("chunks") and decides by itself how many of
// them the rule allows.  Under the fixed-lag rule chunk c runs iff the counts of the check points c - 2 and earlier did
// not end the episode, so a launch covering the check points [a, a + k), k <= 2, needs the verdicts of a - 2 (run at
// all?) and a - 1 (run the second chunk?).  Both counts belong to launches that have COMPLETED when this one starts: its
// first workgroup sums and publishes them (the carries), the host turns them into verdicts (all-reducing over the ranks
// where there are several) and writes ONE word {launch id, skip_all, stop_mid} into pinned memory; thread 0 of the
// launch relays it into device memory, and every wavefront reads it before its first store.  No wavefront ever waits
// for another wavefront of its own launch, so the scheme needs no co-residency and works at any grid size.
struct StopRulePair {
    uint32_t* wave_done_mid;      // counts after the first chunk (nullptr when the launch covers one chunk)
    uint32_t* wave_done_fin;      // counts after the last step
    StopRuleCarry carry[2];       // check points a - 2 and a - 1, where still unpublished
    const long long* verdict_host;
    long long* verdict_dev;
    long long* verdict_err;       // pinned word a launch sets when it gave up waiting
    long long wait_ticks;         // how long the launch waits for its verdict (100 MHz ticks from its start)
    long long launch_id;
    long long first_check_point;  // a
    int n_chunks;                 // k
};
bool stoprule_pairs_supported(const PulseStopRule* h, int n_partials);
// <0: error; 0: launch; 1: the episode is already known to be over (the count a - 2 had been read earlier): do not launch
int stoprule_pair_claim(PulseStopRule* h, int n_partials, int n_chunks, StopRulePair* plan);
int stoprule_pair_commit(PulseStopRule* h, const StopRulePair* plan, int n_partials, ihipStream_t* stream);
// After the launch was enqueued: waits for the counts it carries, writes its verdict word, tells how many of its chunks run.
// *gave_up = 1: the host was too late, the launch ran nothing, the handle is back before it and pairs no more (stoprule.hip).
int stoprule_pair_verdict(PulseStopRule* h, const StopRulePair* plan, int* chunks_run, int* over, int* gave_up);

}  // namespace pulse
