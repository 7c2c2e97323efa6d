/*
 * pulse_env.h -- C ABI of libpulse_hip.so, the MI355X (gfx950) batched env-step engine.
 *
 * Drop-in boundary for the hot path of cerredz/Pulselib: each entry point replaces one chain of
 * eager torch ops in the reference (file:line given per function, relative to the reference
 * checkout).  Plain pointers and sizes only; all device buffers are allocated and owned by the
 * caller (PyTorch-ROCm in the shipped host code, anything else that can hand out HBM pointers
 * works too).  Every launch is enqueued on the caller's hipStream_t and returns without a host
 * sync.  Return value: 0 on success, negative PULSE_E* on error (message: pulse_last_error()).
 * The library never throws across this boundary and keeps no per-call device allocations.
 *
 * There is no CPU fallback: without a usable HIP device the launch functions return
 * PULSE_ENODEVICE.  (pulse_handranks_generate is a host-side data-file builder, not a fallback.)
 */
#ifndef PULSE_ENV_H
#define PULSE_ENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PULSE_ABI_VERSION 1

#define PULSE_EINVAL    (-1)   /* bad argument (shape / null pointer / unsupported size)            */
#define PULSE_ENODEVICE (-2)   /* no HIP device / HIP runtime error before launch                   */
#define PULSE_ELAUNCH   (-3)   /* hipLaunchKernel / hipGetLastError reported an error               */
#define PULSE_EINTERNAL (-4)

#define PULSE_HANDRANKS_LEN 32487834   /* int32 entries of HandRanks.dat (PokerGPU.py:51-57)        */
#define PULSE_MAX_SEATS     16         /* seats per table the kernels support (reference default 10) */

/* Seat status codes, environments/Poker/PokerGPU.py:11 */
enum { PULSE_ACTIVE = 0, PULSE_FOLDED = 1, PULSE_ALLIN = 2, PULSE_SITOUT = 3 };

/* Agent types for pulse_poker_policy, environments/Poker/utils.py:80-87 (+ Player.py:79-176) */
enum {
    PULSE_AGENT_EXTERNAL = 0,        /* seat is played by the caller (Q-network): action left untouched */
    PULSE_AGENT_RANDOM = 1,          /* utils.py:121                 */
    PULSE_AGENT_HEURISTIC_HANDS = 2, /* Player.py:79-104             */
    PULSE_AGENT_TIGHT_AGGRESSIVE = 3,/* Player.py:106-126            */
    PULSE_AGENT_LOOSE_PASSIVE = 4,   /* Player.py:128-151            */
    PULSE_AGENT_SMALL_BALL = 5       /* Player.py:153-176            */
};

/*
 * Device view of one PokerGPU instance: the reference's own SoA tensors
 * (environments/Poker/PokerGPU.py:81-155), row-major, device pointers.
 *   [N]      int32 : pots stages deck_positions button sb bb idx highest agg acted
 *                    last_raise_size prev_stacks prev_invested
 *   [N]      uint8 : is_done (torch.bool), equity_dirty (torch.bool)
 *   [N,P]    int32 : stacks current_round_bet total_invested status
 *   [N,P,2]  int32 : hands          [N,5] int32 : board        [N,52] int32 : decks (cards 1..52)
 *   [N,A]    fp32  : equities       [N,obs_size] fp32 : obs (obs_size = 13 + 3*(max_players-1))
 *   hand_ranks: int32[hand_ranks_len], the 2+2 table (PokerGPU.py:47-58)
 *   w1,w2 (fp32) K,alpha (int32): 0-d device tensors, read by the kernel at every step because
 *   callers re-assign them (tests/poker/test_poker_gpu_round_progression.py:207-210).
 * is_done_out may alias is_done (in-place) or be a second buffer (the host code ping-pongs two so
 * that the tensor returned by step() is not rewritten by the next step, as in PokerGPU.py:619-623).
 */
/* flags: kernel variants a caller (or a test) selects per view; 0 = the product defaults */
#define PULSE_VIEW_NO_OBS_STAGING 0x1  /* store the observation column by column instead of LDS-staged 16-byte bursts  */
#define PULSE_VIEW_NO_CHUNK       0x2  /* pulse_poker_rollout: one launch per step instead of one launch per chunk     */
#define PULSE_VIEW_FOUR_LANES     0x4  /* chunk launches: four lanes per table also where two are the default (<= 10 seats) */
#define PULSE_VIEW_NO_PAIRS       0x8  /* pulse_poker_rollout_until: one check interval per launch instead of two (lag 1)  */
typedef struct PulsePokerView {
    int32_t n_games, n_players, active_players, max_players;   /* n_games <= 2^24 per view (shard larger batches) */
    int32_t obs_size, hand_ranks_len;
    int32_t flags, reserved0;
    const int32_t* hand_ranks;
    int32_t *pots, *stages, *deck_positions, *button, *sb, *bb, *idx, *highest, *agg, *acted,
            *last_raise_size, *prev_stacks, *prev_invested;
    uint8_t *is_done, *is_done_out, *equity_dirty;
    int32_t *stacks, *current_round_bet, *total_invested, *status;
    int32_t *hands, *board;
    const int32_t* decks;
    float *equities, *obs;
    const float *w1, *w2;
    const int32_t *K, *alpha;
    /* Optional evaluation cache (all four NULL = disabled), owned by the caller like the rest, opaque to
     * it: pulse_poker_reset walks the table once per episode (board first, which reaches the same table
     * state as the reference's hole-cards-first order for any distinct cards) and stores, per table, the
     * board it will deal (pre_board [N], packed cards + valid bit) and, per seat, the hole cards used
     * (pre_hands [N,P]), the three street equities (pre_eq [N,3,P], PokerGPU.py:455-525) and the 7-card
     * rank (pre_rank [N,P], PokerGPU.py:437-444).  The step kernel uses an entry only if the cards it
     * finds in `board` / `hands` at that moment are exactly the cached ones, else it does the reference's
     * literal gather chain -- so poked states stay bit-exact and the common path needs no table gathers. */
    int32_t *pre_board, *pre_hands;
    float *pre_eq;
    int32_t *pre_rank;
} PulsePokerView;

/* Phase bits for pulse_poker_phases: the white-box methods the reference's tests call directly. */
#define PULSE_PH_CAPTURE   0x001u  /* PokerGPU.py:530-539  prev_done / actor / prev_stacks / prev_invested */
#define PULSE_PH_EQUITY    0x002u  /* PokerGPU.py:455-525  calculate_equities                             */
#define PULSE_PH_EXECUTE   0x004u  /* PokerGPU.py:230-303  execute_actions                                */
#define PULSE_PH_ADVANCE   0x008u  /* PokerGPU.py:547-616  next actor, round close, street transition     */
#define PULSE_PH_FOLDWIN   0x010u  /* PokerGPU.py:331-338  resolve_fold_winners                           */
#define PULSE_PH_SHOWDOWN  0x020u  /* PokerGPU.py:380-453  resolve_terminated_games + :340-378 side pots  */
#define PULSE_PH_CLEARDONE 0x040u  /* PokerGPU.py:625-628                                                 */
#define PULSE_PH_REWARD    0x080u  /* PokerGPU.py:305-329,:631-632 poker_reward_gpu                       */
#define PULSE_PH_OBS       0x100u  /* PokerGPU.py:159-179  get_obs                                        */
#define PULSE_PH_STEP      0x1FFu  /* all of the above in reference order = PokerGPU.step :527-633        */

int pulse_version(void);
const char* pulse_last_error(void);

/* Host: build the 2+2 table (replaces the HandRanks.dat download, PokerGPU.py:47-58).
 * out = host int32[PULSE_HANDRANKS_LEN].  Deterministic; n_threads<=0 = all cores. */
int pulse_handranks_generate(int32_t* out, int n_threads);

/* Standalone 5/6/7-card lookup kernel (PokerGPU.py:437-444, :497-500, :518-521):
 * cards device int32[n_hands,n_cards]; out device int32[n_hands]; n_cards 7 -> raw chain value,
 * 6 -> HR[p], 5 -> HR[p] (flop_double=0) or HR[HR[p]] (flop_double=1). */
int pulse_poker_eval_hands(const int32_t* hand_ranks, int32_t hand_ranks_len, const int32_t* cards,
                           int32_t n_hands, int32_t n_cards, int32_t flop_double, int32_t* out, void* stream);

/* The closed-form evaluator the reset kernel fills its evaluation cache with (csrc/hand_eval_device.h: the evaluator
 * the table is generated from), alone, for the tests that hold it to the table walk: cards device
 * int32[n_hands,n_cards] of 5..7 DISTINCT cards in 1..52; out[i] = category << 12 | index, what the walk of
 * PokerGPU.py:437-444 returns for them (n_cards 5 / 6: HR[p] after the last card).  Diagnostic, not drop-in surface. */
int pulse_poker_eval_closed_form(const int32_t* cards, int32_t n_hands, int32_t n_cards, int32_t* out, void* stream);

/* PokerGPU.step (PokerGPU.py:527-633), one fused launch.  actions device int64[N] (any value; <0 =
 * no-op, >12 = raise of 0 as in the reference's masks), rewards device fp32[N] out. */
int pulse_poker_step(const PulsePokerView* v, const int64_t* actions, float* rewards, void* stream);

/* Run a subset of step()'s phases in reference order (white-box methods).  For FOLDWIN/SHOWDOWN
 * without CAPTURE the "newly done" mask is is_done itself, as in PokerGPU.py:333,386.  actor_idx
 * (device int32[N]) is used by REWARD when CAPTURE is absent (PokerGPU.py:305), else NULL. */
int pulse_poker_phases(const PulsePokerView* v, uint32_t phases, const int64_t* actions,
                       const int32_t* actor_idx, float* rewards, void* stream);

/* PokerGPU.reset (PokerGPU.py:73-157) after the host picked active_players (v->active_players).
 *   first          : 1 = no previous episode (stacks := starting_bbs, button := 0)           (:101-102,:121)
 *   prefixed_decks : device int32[N,52] copied into v->decks, or NULL = shuffle on device with
 *                    Philox4x32-10(seed, table id + table_id0, episode) (replaces rand().argsort, :86)
 *   decks_out      : v->decks is const in the view; reset writes through this pointer.
 *   rotation       : torch.roll shift of the stack rows                                     (:104-110)
 *   shuffle_key_bits : the device shuffle orders the 52 cards of a table by the top 26 bits of one Philox word each
 *                    (torch.rand's keys have 24), equal keys keeping card order; 1..25 keeps fewer bits and so
 *                    forces ties: a hook for the tests of the tie-break, not for production use.  0 = 26.
 *   stats_rewards / stats_out : NULL, or the episode statistics of the episode that ENDS here, taken before the state
 *                    is overwritten: the sum of stats_rewards[t] (device fp32[n_games], the last step's rewards) and the
 *                    number of tables with is_done (what a rank all-reduces per episode, scripts/Poker/trainGPU.py:96,
 *                    104) are ADDED into stats_out, device double[PULSE_STATS_SLOTS * PULSE_STATS_STRIDE]: accumulator k
 *                    is {stats_out[k * STRIDE] = reward sum, stats_out[k * STRIDE + 1] = done count}, a wavefront adds to
 *                    accumulator (its index mod SLOTS); the total is the sum over k.  (Spread because same-address
 *                    atomics serialise.)  No launch of its own. */
#define PULSE_STATS_SLOTS 256
#define PULSE_STATS_STRIDE 16
typedef struct PulsePokerResetOpts {
    int32_t first, starting_bbs, max_bbs, rotation;
    uint64_t seed, episode, table_id0;
    const int32_t* prefixed_decks;
    int32_t* decks_out;
    int32_t shuffle_key_bits, reserved0;
    const float* stats_rewards;
    double* stats_out;
} PulsePokerResetOpts;
int pulse_poker_reset(const PulsePokerView* v, const PulsePokerResetOpts* o, void* stream);

/* build_actions + scripted opponents (environments/Poker/utils.py:108-123, Player.py:79-176):
 * obs device fp32[n,obs_stride] (columns 5,6 = hole cards, 9 = pot), seat_idx device int32[n] (the
 * `curr_players` argument).  For every table whose seat has a scripted type, actions[t] is written;
 * EXTERNAL seats are left untouched.  agent_types: host uint8[n_players] (PULSE_AGENT_*).
 * Random draws (the scripted-opponent stream): call = Philox4x32-10(seed, table id + table_id0, step_counter >> 1);
 * an even step_counter takes words (x, y) of the call, an odd one (z, w) -- x/z feed the action's randint, y/w
 * loose_passive's rand() (Player.py:146).  One call serves two steps: the chunked roll-out draws eight steps in one pass. */
int pulse_poker_policy(const float* obs, int32_t obs_stride, const int32_t* seat_idx, int32_t n,
                       const uint8_t* agent_types, int32_t n_players, uint64_t seed, uint64_t step_counter,
                       uint64_t table_id0, int64_t* actions, void* stream);

/* Fused roll-out step for scripted tables: policy (above) + step in ONE launch; actions[] is both
 * input (EXTERNAL seats) and output (what was played).  Same results as policy followed by step. */
int pulse_poker_policy_step(const PulsePokerView* v, const uint8_t* agent_types, uint64_t seed, uint64_t step_counter,
                            uint64_t table_id0, int64_t* actions, float* rewards, void* stream);

/* Roll-out: n_steps fused policy+step transitions enqueued by ONE native call and, by default, executed by ONE
 * launch: every table's state is loaded once, stepped n_steps times in registers and its changed words stored once,
 * while every step still stores its observation, reward, done flag and action -- step i into the even buffers
 * (v_even->obs, v_even->is_done_out, rewards_even) when i is even, else into the odd ones (v_odd->obs,
 * v_odd->is_done_out, rewards_odd).  Memory after the call is bit-identical to calling pulse_poker_policy_step
 * n_steps times on v_even, v_odd, v_even, ... (PULSE_VIEW_NO_CHUNK in v_even->flags does exactly that instead).
 * v_odd must be v_even with is_done <-> is_done_out swapped (obs may differ: double-buffered observations).
 * Draws of step i: the scripted-opponent stream at step_counter0 + i (pulse_poker_policy).
 * timer (NULL or a pulse_timer_create handle): the call is bracketed by one HIP event pair on `stream`; read the
 * summed time, the launches and the steps covered with pulse_timer_collect() AFTER synchronising the stream.
 * stoprule (NULL or a pulse_stoprule_create handle): the tables done after the last step are counted for it by the
 * launch itself (each wavefront stores its count; no extra kernel on `stream`). */
int pulse_poker_rollout(const PulsePokerView* v_even, const PulsePokerView* v_odd, const uint8_t* agent_types,
                        uint64_t seed, uint64_t step_counter0, uint64_t table_id0, int64_t* actions,
                        float* rewards_even, float* rewards_odd, int32_t n_steps, void* timer, void* stoprule,
                        void* stream);
/* The trainer's inner loop (scripts/Poker/trainGPU.py:79-108) for scripted tables, natively: roll-out chunks of
 * chunk_steps (the reference's check interval, 5) are enqueued, each followed by the stop rule's verdict
 * (pulse_stoprule_decide), until the rule ends the episode or max_steps steps have run -- no interpreter between the
 * chunks.  *steps_done: steps executed (the caller's views / reward buffers have swapped roles if it is odd);
 * *over: the rule fired.  timer + time_every > 0: every time_every-th chunk opens a HIP-event bracket over the next
 * eight chunks of the call (an event pair around every single launch costs the stream a fifth of the launch).
 * With a lag-1 rule (and unless the view says PULSE_VIEW_NO_PAIRS) ONE launch runs up to TWO check intervals: under
 * the fixed-lag rule a chunk runs iff the count two check points back did not end the episode, and both counts a
 * two-chunk launch needs belong to launches that have completed when it starts -- its first workgroup sums and
 * publishes them, the host answers with the launch's verdict word in pinned memory, and every wavefront reads the
 * relayed word before its first store (the launch then runs two chunks, one, or none).  Episodes come out step for
 * step as with one chunk per launch; the state is loaded and written back half as often.
 * That launch is the one place where device code waits for the host (the word arrives ~5 us after the launch starts).  The
 * wait is bounded (PULSE_STOPRULE_OPT_VERDICT_WAIT_TICKS, 20 s by default) and the host decides by its own clock which side
 * of the bound it is on: a host that is later than half the wait after the enqueue (a stalled rank, a stopped process) does
 * NOT write the word, lets the launch give up -- it then runs nothing and stores nothing --, counts a time-out
 * (pulse_stoprule_stats) and runs these and all further steps of the handle's life with one check interval per launch, which
 * never waits in a kernel.  Same episodes either way.  Ranks that share one device (shm exchange) never pair. */
int pulse_poker_rollout_until(const PulsePokerView* v_even, const PulsePokerView* v_odd, const uint8_t* agent_types,
                              uint64_t seed, uint64_t step_counter0, uint64_t table_id0, int64_t* actions,
                              float* rewards_even, float* rewards_odd, int32_t chunk_steps, int32_t max_steps, void* timer,
                              int32_t time_every, void* stoprule, void* stream, int32_t* steps_done, int32_t* over);
int pulse_timer_create(void** timer);
int pulse_timer_collect(void* timer, float* sum_ms, int32_t* n_launches, int64_t* n_steps);
int pulse_timer_destroy(void* timer);

/* The trainer's episode stop rule (scripts/Poker/trainGPU.py:27-33,99: every 5th step, more than `threshold` of the
 * tables done ends the episode) without its blocking read, for one GPU or for one process per GPU.
 * A "check point" is one evaluation of the rule.  pulse_poker_rollout (or submit) leaves the check point's done
 * tables counted per wavefront on the device; the NEXT launch on that stream sums them in its first workgroup (before that workgroup's own tables) and
 * writes the count + a sequence number into coherent pinned host memory, which decide()/counts() poll -- no event,
 * no second stream, no kernel of its own between the steps' launches, and the host sees a count a microsecond after
 * it exists.  decide() answers for the check point submitted `lag` check points before the newest one: a FIXED lag,
 * so a run is reproducible and every rank of a job takes every decision on the same check point and ends every
 * episode at the same step (equal collective sequences on all ranks).  lag 0 is the reference's blocking check (the
 * count is then flushed by a kernel of its own); lag 1 (default of the host code) never waits in practice.
 * Between the ranks of a job the counts are exchanged
 *   - through a POSIX shared-memory segment (shm_name, rank, world: one cache line per rank and check point; the
 *     ranks of one node; 8 bytes every 5 steps need no device), or
 *   - by an RCCL all-reduce on a side stream of the rule (comm: a pulse_comm_create handle; an event hands each check
 *     point over to that stream, which costs the steps' stream a barrier per check point), or
 *   - by the caller: counts() returns the check point's local count, to be all-reduced by the host code (gloo).
 * counts(): have = 0 while no check point of this episode is due.  drain() marks an episode boundary: earlier check
 * points decide nothing any more.  n_global = tables of the whole job.  The handle owns < 1 MB of device / pinned
 * memory (and, RCCL only, a stream and 8 events), created on the current device. */
int pulse_stoprule_create(int32_t n_local, int64_t n_global, double threshold, int32_t lag, void* comm, const char* shm_name,
                          int32_t rank, int32_t world, void** handle);
int pulse_stoprule_submit(void* handle, const uint8_t* flags, int32_t n, void* stream);   /* flags: device uint8[n], != 0 = done */
int pulse_stoprule_counts(void* handle, int64_t* local, int64_t* global, int32_t* have);
int pulse_stoprule_decide(void* handle, int32_t* over);
/* Publishes the newest check point's count NOW (a one-workgroup launch on the stream that wrote its partial counts) instead
 * of leaving it to the next launch that carries the rule: for a caller whose next such launch is far away (the trainer: every
 * fifth step), so that the verdict due then finds its count long published and waits for nothing. */
int pulse_stoprule_publish(void* handle);
int pulse_stoprule_drain(void* handle);
int pulse_stoprule_destroy(void* handle);
/* How the handle exchanges its counts: 0 = local (one process), 1 = RCCL side stream (any communicator, also of one
 * rank), 2 = shared memory; side_launches = check points that went through the side stream so far (tests assert the
 * RCCL leg really ran). */
int pulse_stoprule_mode(void* handle);
int64_t pulse_stoprule_side_launches(void* handle);
/* Options of a handle.  VERDICT_WAIT_TICKS: how long a paired launch (pulse_poker_rollout_until) waits for its host's verdict,
 * in ticks of the device's 100 MHz clock (default 2,000,000,000 = 20 s; >= 100,000).  ALLOW_SHARED_DEVICE_PAIRS (0 / 1):
 * with the shm exchange, ranks whose GPU (PCI bus id, told to the segment at create time) is also another rank's do not pair,
 * since their launches wait for hosts that wait for every rank's launch to have started and the grids of several processes
 * need not fit one device together; 1 lifts that for grids known to fit (tests).  DEBUG_LATE_VERDICTS (test hook): the next
 * `value` verdicts are treated as "the host is too late": the launch gives up, the handle falls back.
 * stats: out4 = {paired launches issued, verdict time-outs, 1 if the handle would still pair, side-stream check points}. */
#define PULSE_STOPRULE_OPT_VERDICT_WAIT_TICKS        0
#define PULSE_STOPRULE_OPT_ALLOW_SHARED_DEVICE_PAIRS 1
#define PULSE_STOPRULE_OPT_DEBUG_LATE_VERDICTS       2
int pulse_stoprule_set_option(void* handle, int32_t option, int64_t value);
int pulse_stoprule_stats(void* handle, int64_t* out4);

/* The shared-memory exchange of the stop rule as an object of its own (host code only; the rule creates one itself when
 * given shm_name).  Every rank maps the POSIX segment `name` (created by whoever comes first; unlink it once all ranks
 * have it mapped) and calls all_sum with the same sequence of indices 0, 1, 2, ...: *total = sum of the ranks' values. */
int pulse_shm_create(const char* name, int32_t rank, int32_t world, void** handle);
int pulse_shm_all_sum(void* handle, int64_t index, int64_t value, int64_t* total);
/* The segment also carries one record per rank naming the GPU the rank computes on (any string equal exactly for ranks on
 * one device; the rule stores the PCI bus id).  device_is_private: 1 = every rank has named its GPU within wait_ms and none
 * shares this rank's, 0 otherwise. */
int pulse_shm_set_device(void* handle, const char* bus_id);
int pulse_shm_device_is_private(void* handle, int32_t wait_ms);
int pulse_shm_destroy(void* handle);

/* RCCL communicator of the job (one process per GPU), for the stop rule's 8-byte all-reduce on its side
 * stream (the `comm` exchange above).  librccl is bound at run time (the copy PyTorch-ROCm loaded, else the system one), so the library loads
 * without it.  unique_id: call on rank 0, hand the 128 bytes to every rank (torch.distributed broadcast), then
 * every rank calls create (collective).  all_reduce_i64: sum of int64[count], device pointers, in `stream` order. */
int pulse_comm_unique_id(uint8_t* out128);
int pulse_comm_create(const uint8_t* id128, int32_t rank, int32_t world, void** comm);
int pulse_comm_all_reduce_i64(void* comm, const int64_t* send, int64_t* recv, int32_t count, void* stream);
int pulse_comm_destroy(void* comm);

/* Diagnostic only (tools/ablate_step.py): fused policy+step with phases compiled out, to price them. */
int pulse_poker_ablate(const PulsePokerView* v, uint32_t phases, int64_t* actions, float* rewards, uint64_t types_packed,
                       uint64_t step_counter, void* stream);

/* Diagnostic only (tools/pmc_calibrate.py): stream n_words dwords, one dword per lane, read (write=0) or
 * written (write=1), to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on a known byte count. */
int pulse_calib_stream(int32_t* buf, uint64_t n_words, int32_t write, void* stream);

/* Episode statistics for the trainer's stop rule and returns (scripts/Poker/trainGPU.py:27-33,96):
 * stats device int64[2] += {#tables with is_done, 0}; fstats device double[1] += sum(rewards[mask]).
 * stats NULL: the done count goes to fstats[1] instead (double[2] = {reward sum, done count}: one tensor to all-reduce). */
int pulse_poker_stats(const uint8_t* is_done, const float* rewards, const uint8_t* mask, int32_t n,
                      int64_t* stats, double* fstats, void* stream);

/* Hand-level metrics side-channel (scripts/Poker/trainGPU_performance.py:198-206, utils/performance.py): for every
 * table with dones[t] && !terminated_before[t] (NULL = none terminated) the learner seat's chip delta
 * stacks[t][q_seat] - initial_q_stacks[t] is added to acc[position][bucket][4] = {hands, wins, sum delta, sum delta^2}
 * (device int64[16*5*4], zeroed by the caller when a new aggregation starts); position = (q_seat - button[t]) mod
 * active_players, bucket = clamp(stages[t], 0, 4).  No host sync; BB/100, win rates by street / position and the
 * confidence bound follow from the sums on the host (pulselib_amd/utils/performance.py).
 * The ordered hand log (optional; both NULL or both set): finish_step[t] = step_index and hand_delta[t] = the delta for
 * every table accounted in this call (device int32[n] each; the caller fills finish_step with -1 when an episode
 * starts).  Sorting the finished tables by (finish_step, t) gives the hands in the order the reference's per-step
 * boolean pulls append them -- what its rolling-window average runs over (utils/performance.py:128-135). */
int pulse_poker_hand_metrics(const uint8_t* dones, const uint8_t* terminated_before, const int32_t* stacks, int32_t n_players,
                             const int32_t* initial_q_stacks, const int32_t* stages, const int32_t* button, int32_t q_seat,
                             int32_t active_players, int32_t n, int64_t* acc, int32_t step_index, int32_t* finish_step,
                             int32_t* hand_delta, void* stream);

/* ---- Blackjack (environments/blackjack/blackjack.py) ------------------------------------------ */
typedef struct PulseBlackjackView {
    int32_t batch_size;
    const int32_t* decks;              /* [B,52] cards 0..51 (blackjack.py:24-29)                  */
    int32_t *deck_positions, *players_cards, *players_card_idx, *player_card_sums;
    int32_t *dealer_cards, *dealer_card_idx, *dealer_upcard, *dealer_card_sums;
    uint8_t *terminated, *has_ace, *dealer_has_ace;
    int32_t *rewards, *obs;            /* obs [B,3] int32                                          */
} PulseBlackjackView;
/* reset: decks_src NULL = device shuffle Philox(seed, game id, episode), else copied (blackjack.py:23-101) */
int pulse_blackjack_reset(const PulseBlackjackView* v, const int32_t* decks_src, int32_t* decks_out,
                          uint64_t seed, uint64_t episode, void* stream);
int pulse_blackjack_step(const PulseBlackjackView* v, const int64_t* actions, void* stream);   /* :113-186 */

/* ---- 2048 (environments/2048/TFE.py), batched: boards device int32[B,n,n], n = 2..8 ------------
 * 4 x 4 (config/tfe.yaml) with 16-byte aligned boards runs packed: the board as 64 bits of 4-bit log2 tiles, the move as four
 * lookups in a 65,536-entry row table the library builds on the device at the first call (256 KB of static device memory, one
 * synchronisation of the null stream, once per device); a board holding anything but 0 and the powers of two 2 .. 16,384 is
 * stepped cell by cell like the other sides -- any int32 board gives the reference's result. */
int pulse_tfe_reset(int32_t* boards, int64_t* total_score, int32_t n_boards, int32_t n, uint64_t seed,
                    uint64_t board_id0, void* stream);                                          /* :143-149 */
int pulse_tfe_step(int32_t* boards, int64_t* total_score, const int64_t* actions, int32_t* rewards,
                   uint8_t* dones, int32_t n_boards, int32_t n, uint64_t seed, uint64_t board_id0,
                   uint64_t step_counter, void* stream);                                        /* :152-189 */

/* ---- tabular Q-learning for the batched 2048 roll-out (utils/numba.py:5-39, agents/TemperalDifference/QLearningNumba.py:10-37)
 * The reference's `defaultdict(state -> float64[4])` is an open-addressing hash table of 64-byte ENTRIES in HBM:
 *   entry = { uint64 key (the board packed as 4-bit log2 tiles; 0 = free), double q[4], 24 spare bytes }
 * -- key and values of a state in ONE memory line.  entries: device memory, capacity * 64 bytes, 64-byte aligned,
 * zero-initialised by the caller; capacity and region_slots are powers of two.  region_slots > 0: board g owns entries
 * [g*region_slots, (g+1)*region_slots) (independent learners = copies of the reference agent, race-free, bit-exact);
 * 0: one table shared by all boards.  Shared table: an update whose single compare-and-swap loses against a concurrent
 * update of the same cell is deferred, and the deferred transitions of a launch are combined per cell --
 *   q <- q + (1 - (1 - alpha)^k) (mean of the k targets - q)   (k = 1: numba.py:38-39 itself)
 * -- with plain atomic adds into the per-launch scratch below (linear in the number of contenders; a CAS retry loop is
 * quadratic: all boards leave reset from a few hundred states). */
#define PULSE_QTABLE_ENTRY_BYTES 64
typedef struct PulseQTable {
    void* entries;
    uint64_t capacity, region_slots;
} PulseQTable;
/* Scratch of a shared table (NULL for private regions): caller-owned device memory, zero-initialised once.
 * count uint32[64 + 2 * n / 256]; cells uint64[n]; targets double[n]; owner int32[n]; acc_key uint64[acc_slots]; acc_cnt
 * uint32[acc_slots]; acc_sum double[acc_slots]; n >= n_boards, a multiple of 256 (the deferred list is kept in one segment
 * per workgroup of the launch; at most 1,048,576 boards per launch); acc_slots a power of two >= 2 n.  launch_index: the
 * caller counts its update / rollout_step launches on this scratch (0, 1, 2, ...: its parity picks the list).
 * A long list of deferred transitions is combined by 64 workgroups that meet inside the follow-up launch; if they cannot
 * gather within wait_ticks (a co-tenant on the GPU) the meeting is called off as a whole, that launch's combined updates
 * are dropped, and the next call on a shared table clears the accumulators and returns PULSE_EINTERNAL once. */
typedef struct PulseQTableScratch {
    uint32_t* count;
    uint64_t* cells;
    double* targets;
    int32_t* owner;
    uint64_t* acc_key;
    uint32_t* acc_cnt;
    double* acc_sum;
    uint32_t n, acc_slots;
    int64_t wait_ticks;             /* how long the follow-up launch's workgroups wait for each other (100 MHz ticks); 0 = 3 s */
    int32_t debug_meet_extra;       /* test hook: arrivals that meeting expects beyond the grid's own (> 0: it never comes about) */
    int32_t reserved0;
} PulseQTableScratch;
/* state lookup/insert + epsilon-greedy: actions int64[B] out, slots int64[B] out (-1 = no room: the region is full or
 * 4,096 consecutive slots were taken -- size a shared table for the states a run will visit; such a board acts at
 * random and is not updated) */
int pulse_qtable_select(const PulseQTable* q, const int32_t* boards, int32_t n_boards, int32_t n, double epsilon,
                        uint64_t seed, uint64_t board_id0, uint64_t step_counter, int64_t* actions, int64_t* slots,
                        void* stream);
/* q[s][a] += alpha * ((terminal ? r : r + gamma * max q[s']) - q[s][a]) for every board */
int pulse_qtable_update(const PulseQTable* q, const PulseQTableScratch* scratch, uint64_t launch_index, const int64_t* slots,
                        const int64_t* actions, const int32_t* rewards, const int32_t* next_boards, const uint8_t* terminal,
                        int32_t n_boards, int32_t n, double alpha, double gamma, void* stream);
/* One roll-out step of B learners in ONE launch: pulse_qtable_select + pulse_tfe_step + pulse_qtable_update with the board
 * in registers throughout (same draws, same results: the agent's stream (agent_seed, board, agent_step), the
 * environment's (env_seed, board, env_step >= 1)).  slots_io int64[B]: in = the entry of every board's current state if
 * the previous step found it, -2 = look it up; out = the entry of the state the move led to.  boards / total_score are
 * updated in place, actions / rewards / dones written as the three calls would. */
int pulse_qtable_rollout_step(const PulseQTable* q, const PulseQTableScratch* scratch, uint64_t launch_index, int32_t* boards,
                              int64_t* total_score, int32_t n_boards, int32_t n, double epsilon, double alpha, double gamma,
                              uint64_t agent_seed, uint64_t agent_step, uint64_t env_seed, uint64_t env_step, uint64_t board_id0,
                              int64_t* actions, int32_t* rewards, uint8_t* dones, int64_t* slots_io, void* stream);

/* ---- the learner's action selection (environments/Poker/Player.py:178-253) ---------------------
 * PokerQNetwork.network in eval mode: Linear(state_dim,128) GELU Linear(128,128) GELU [Dropout] Linear(128,64)
 * GELU [Dropout] Linear(64,32) GELU Linear(32,n_actions) (:189-201).  Weights are the module's own tensors:
 * torch.nn.Linear layout w[out][in] row-major fp32 on the device, 16-byte aligned; nothing is packed or cached
 * between calls, so an optimizer step is visible to the next call. */
typedef struct PulseQNet {
    int32_t state_dim, n_actions;                       /* n_actions <= 32 */
    const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4, *w5, *b5;
} PulseQNet;
/* q_out[r, :] = network(states[r, :]) (Player.py:235-240); states fp32 rows of state_dim at row_stride floats. */
int pulse_qnet_forward(const PulseQNet* net, const float* states, int64_t row_stride, int32_t n_rows, float* q_out,
                       void* stream);
/* PokerQNetwork.get_actions (Player.py:242-253) fused with build_actions' mask (utils.py:108-119): for every row
 * r with seat_idx[r] == q_seat (seat_idx NULL: every row) actions[r] = uniform{0..n_actions-1} with probability
 * epsilon, else the first argmax of the Q row; other rows are left untouched.  Draws are words x (explore) and y
 * (action) of Philox4x32-10(seed, table_id0 + r, step), the stream pulse_poker_policy uses for scripted seats.
 * q_out NULL or fp32[n_rows, n_actions] (selected rows written).  row_mask_out (NULL or device uint8[n_rows], needs
 * seat_idx): row_mask_out[r] = (seat_idx[r] == q_seat) && !terminated[r] (terminated NULL = none) for EVERY row --
 * the trainer's `q_mask & ~terminated` (scripts/Poker/trainGPU.py:85), the row_mask of pulse_qnet_train_step.
 * Kernels: with seat_idx, state_dim a multiple of 4 in 13..64, n_actions <= 16 and 16-byte aligned rows (the Poker shapes)
 * sixteen rows per wavefront with the activations in registers (csrc/qnet_rows16.h); otherwise cooperative 32-row tiles
 * (csrc/qnet_device.h).  The two sum a layer's inputs in different orders: Q values agree to rounding (2e-5 at |Q| = O(1)),
 * actions wherever the two best Q values of a row are further apart than that. */
int pulse_qnet_act(const PulseQNet* net, const float* states, int64_t row_stride, int32_t n_rows,
                   const int32_t* seat_idx, int32_t q_seat, float epsilon, uint64_t seed, uint64_t step,
                   uint64_t table_id0, int64_t* actions, float* q_out, const uint8_t* terminated,
                   uint8_t* row_mask_out, void* stream);
/* The same (seat_idx and row_mask_out required) for a trainer that will call pulse_qnet_train_step / _grads next with
 * THIS `states` as its states and THIS row_mask_out as its row_mask: the rows that call trains on (row_mask & seat
 * status ACTIVE / ALLIN) are known already, so their lists are written to the trainer's select_scratch here and the
 * training call skips its selection launch -- set PulseQNetTrain.select_from_act = 1 for that call (and only that one).
 * select_words: at least 259 per 256 rows + 512.  (Shapes that take the 32-row-tile kernels, see pulse_qnet_act: with 517 per
 * 256 rows + 512 and 262,144 rows or more the selection runs as TWO launches -- the windows only list the learner's rows (in
 * the extra words), a second launch runs them in full tiles -- same results.) */
int pulse_qnet_act_select(const PulseQNet* net, const float* states, int64_t row_stride, int32_t n_rows,
                          const int32_t* seat_idx, int32_t q_seat, float epsilon, uint64_t seed, uint64_t step,
                          uint64_t table_id0, int64_t* actions, const uint8_t* terminated, uint8_t* row_mask_out,
                          int32_t* select_scratch, int64_t select_words, void* stream);

/* PokerQNetwork.train_step (Player.py:255-294) as up to three launches: (0) the row filter as lists of row ids per
 * window (skipped when pulse_qnet_act_select wrote them); (1) the listed rows in tiles of 32, dealt to the workgroups: TD target +
 * forward (train mode) + backward on the matrix cores, gradient sums accumulated per workgroup; (2) reduction of the
 * workgroups' slices, then -- in the same launch -- gradient mean / clip_grad_norm_ / AdamW / target sync.
 * The network and its target each live in ONE flat fp32 buffer of pulse_qnet_param_count() floats laid out
 * w1,b1,w2,b2,w3,b3,w4,b4,w5,b5 (torch layouts); `net` / `target` hold the ten views.  grad, exp_avg, exp_avg_sq: flat
 * buffers of the same length (moments zero before the first call).  partials: device fp32[max_blocks *
 * pulse_qnet_slice_floats()] scratch (max_blocks = number of persistent workgroups, 256 = one per CU; a workgroup's
 * slice is laid out for its own stores, not in parameter order).  step: device int64 optimizer step count
 * (bias correction, target sync every update_freq steps).  stats: device fp32[4] scratch.  report: device fp32[4] out:
 * [0] rows trained on, [1] the MSE loss, [2] gradient norm before clipping, [3] 0, or -1: the reduce + AdamW launch's
 * workgroups did not all gather within meet_wait_ticks (a co-tenant on the GPU), the meeting was called off AS A WHOLE and no
 * parameter, moment or step count moved; the next pulse_qnet_train_* call then returns PULSE_EINTERNAL once.  The in-launch
 * meeting is only used where the device holds the launch's whole grid at once (asked once per device), never with
 * separate_apply = 1 (AdamW as a launch of its own, what pulse_qnet_train_grads / _apply always do).
 * Row filter: row_mask[r] != 0 (NULL: all) and states[r][12] in {0, 2} (:261); if no row passes, nothing changes (:262).
 * Trainer bookkeeping folded in (both optional, scripts/Poker/trainGPU.py:86,96): terminated (device uint8[n_rows],
 * NULL = skip) gets terminated[r] |= dones[r]; reward_sum (device double, NULL = skip) += sum of rewards over the
 * row_mask rows (before the status filter).
 * Dropout(.1) after the 2nd and 3rd GELU draws 16-bit uniforms from Philox4x32-10(seed ^ 0xD50F0D50F0, table_id0 + r,
 * 32 * step_counter + unit / 8); dropout_p = 0 disables it.  Sums over rows run in a fixed order: deterministic for
 * a given n_rows and max_blocks (whichever launch made the row lists). */
typedef struct PulseQNetTrain {
    PulseQNet net, target;
    float *params, *target_params, *grad, *exp_avg, *exp_avg_sq;
    int64_t* step;
    float *stats, *report, *partials;
    float lr, weight_decay, beta1, beta2, eps, max_grad_norm, gamma, dropout_p;
    int32_t update_freq, max_blocks;
    int32_t* select_scratch;        /* device int32[select_words]: the row-selection launch's lists */
    int64_t select_words;           /* >= 259 * ceil(n_rows / 256) + 512 for the largest n_rows passed */
    int32_t select_from_act;        /* 1: the lists in select_scratch were written by pulse_qnet_act_select (see there) */
    int32_t separate_apply;         /* 1: AdamW in a launch of its own (no in-launch meeting of the reduce launch's workgroups) */
    int64_t meet_wait_ticks;        /* how long a workgroup waits at that meeting, in ticks of the 100 MHz clock; 0 = 5 s */
    int32_t debug_meet_extra;       /* test hook: arrivals the meeting expects beyond the grid's own (> 0: it never comes about) */
    int32_t reserved0;
} PulseQNetTrain;
int pulse_qnet_param_count(int32_t state_dim, int32_t n_actions);
int pulse_qnet_slice_floats(void);
/* How many reduce + AdamW launches of this process have called their meeting off so far (a count in pinned host memory the
 * launches add to: reading it waits for nothing; it is behind by the launches still in flight). */
int64_t pulse_qnet_called_off_meetings(void);
int pulse_qnet_train_step(const PulseQNetTrain* t, const float* states, int64_t row_stride, const int64_t* actions,
                          const float* rewards, const float* next_states, int64_t next_stride, const uint8_t* dones,
                          const uint8_t* row_mask, int32_t n_rows, uint64_t seed, uint64_t step_counter,
                          uint64_t table_id0, uint8_t* terminated, double* reward_sum, void* stream);
/* The same in two halves, for data-parallel training (one process per GPU): pulse_qnet_train_grads runs launches 1-2
 * and leaves the UNNORMALISED gradient sum in t->grad and {sum g^2, rows, sum td^2} in t->stats[0..2] without touching
 * t->step; the caller all-reduces t->grad and t->stats[1..2] over the ranks (RCCL), stores the squared norm of the
 * reduced gradient in t->stats[0], adds 1 to *t->step if the reduced row count is positive, and calls
 * pulse_qnet_train_apply (launch 3) -- every rank then applies the identical update. */
int pulse_qnet_train_grads(const PulseQNetTrain* t, const float* states, int64_t row_stride, const int64_t* actions,
                           const float* rewards, const float* next_states, int64_t next_stride, const uint8_t* dones,
                           const uint8_t* row_mask, int32_t n_rows, uint64_t seed, uint64_t step_counter,
                           uint64_t table_id0, uint8_t* terminated, double* reward_sum, void* stream);
int pulse_qnet_train_apply(const PulseQNetTrain* t, void* stream);

/* The first half of the trainer's step with the learner in it as ONE launch (scripts/Poker/trainGPU.py:79-83 with the learner at
 * a seat: build_actions -> env.step): pulse_qnet_act_select on the observation act->states (the learner's actions into
 * `actions`, the trainer's mask into act->row_mask_out, the training launch's row lists into act->select_scratch) followed by
 * pulse_poker_policy_step on `v` -- the same results word for word; the workgroup that picks the actions of a window of 128
 * tables steps those tables itself, so nothing grid-wide lies between the two and the tables' state travels from HBM while
 * the forward runs.  Needs n_games % 128 == 0, max_players <= 10, 16-byte aligned fp32 rows of 16..40 inputs (a multiple of
 * 8); PULSE_EINVAL otherwise: make the two calls instead.  act->states must NOT be the buffer v->obs writes (the trainer's
 * double-buffered observations).  stoprule: as pulse_poker_rollout (the done tables after the step are counted for it by the
 * launch itself), or NULL. */
typedef struct PulseQNetAct {
    const float* states; int64_t row_stride;     /* device fp32 rows, row_stride floats apart: the observation the learner acts (and trains) on */
    const int32_t* seat_idx; int32_t q_seat;     /* rows with seat_idx[r] == q_seat are the learner's */
    float epsilon;
    uint64_t seed, step, table_id0;              /* the learner's exploration stream (pulse_qnet_act) */
    const uint8_t* terminated; uint8_t* row_mask_out;
    int32_t* select_scratch; int64_t select_words;
} PulseQNetAct;
int pulse_poker_act_policy_step(const PulsePokerView* v, const uint8_t* agent_types, uint64_t seed, uint64_t step_counter,
                                uint64_t table_id0, int64_t* actions, float* rewards, const PulseQNet* net,
                                const PulseQNetAct* act, void* stoprule, void* stream);

/* ---- Particle2D (environments/Particle2D/Particle2D.py:22-30) ---------------------------------- */
int pulse_particle2d_step(float* state, const float* action, int32_t* steps, float* obs_out, float* rewards,
                          uint8_t* terminated, int32_t n, float dt, int32_t max_steps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PULSE_ENV_H */
