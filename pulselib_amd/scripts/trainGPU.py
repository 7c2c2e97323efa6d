"""Episode driver with the reference trainer's loop contract (scripts/Poker/trainGPU.py:36-145):

  * seat rotation per episode (get_rotated_agents), reset options {rotation, active_players, q_agent_seat};
  * per step: actions.fill_(0) -> build_actions -> env.step -> learner.train_step on
    `q_mask & ~terminated` evaluated BEFORE `terminated |= dones` (trainGPU.py:85-86);
  * stop rule every 5th step at > 80 % terminated (trainGPU.py:27-33); `total_steps += n_games * idx`
    (trainGPU.py:108) -- finished tables count, as in every number the reference publishes;
  * run summary with the reference's keys (`sps`, `total_steps`, ...; utils/benchmarking/benchmarking.py:84-100).

The learner is whatever object sits at the QLEARNING seat (`get_actions`, `train_step`); ours is
environments/Poker/qnetwork.py: PokerQNetwork, whose native kernels `train_agent_fused` drives without host syncs.  With N > 1 ranks each process drives its own shard and episode sums are all-reduced (RCCL)."""
from __future__ import annotations

import time
from pathlib import Path

import torch

from ..environments.Poker.utils import PokerAgentType, build_actions, get_rotated_agents
from ..sharding import EpisodeStats

CHECK_INTERVAL = 5            # trainGPU.py:31
REPORT_EVERY = 10             # trainGPU.py:110: the reference reports every 10th episode
TERMINATION_THRESHOLD = 0.8   # trainGPU.py:76


def _should_stop_loop(step_idx: int, terminated: torch.Tensor, termination_threshold, check_interval: int = CHECK_INTERVAL) -> bool:
    return step_idx % check_interval == 0 and bool(terminated.float().mean() > termination_threshold)


def train_agent(env, agents, agent_types, episodes, n_games, device, results_dir=None, config=None, plotter=None,
                benchmarker=None, build_actions_fn=build_actions, max_episode_steps=None, reduce_stats=True):
    config = config or {}
    total_steps = 0
    start_time = time.time()
    scores, reward_scores = [], []
    actions = torch.zeros(n_games, dtype=torch.long, device=device)
    q_agent_idx = agent_types.index(PokerAgentType.QLEARNING)
    q_agent = agents[q_agent_idx]
    stats = EpisodeStats(device)

    for episode in range(episodes):
        rotated_agents, rotated_types, q_seat, rotations = get_rotated_agents(agents, agent_types, episode_idx=episode,
                                                                              q_agent_idx=q_agent_idx)
        state, info = env.reset(options={"rotation": rotations, "active_players": True, "q_agent_seat": q_seat})
        initial_stacks = info["stacks"][:, q_seat].clone()
        terminated = torch.zeros(n_games, dtype=torch.bool, device=device)
        episode_reward_tensor = torch.tensor(0.0, device=device)
        termination_threshold = torch.tensor(TERMINATION_THRESHOLD, device=device)
        idx = 0
        while True:
            actions.fill_(0)
            q_mask = info["seat_idx"] == q_seat
            build_actions_fn(state, actions, info["seat_idx"], rotated_agents, rotated_types, device)
            # the env reuses its observation buffer (PokerGPU.py:633): keep the pre-step rows for the learner
            state_before = state[q_mask & ~terminated].clone() if hasattr(q_agent, "train_step") else None
            next_state, rewards, dones, truncated, info = env.step(actions)
            del truncated
            active_games = q_mask & ~terminated
            terminated |= dones
            if state_before is not None and active_games.any():
                q_agent.train_step(states=state_before, actions=actions[active_games], rewards=rewards[active_games],
                                   next_states=next_state[active_games], dones=dones[active_games])
            episode_reward_tensor += rewards[active_games].sum()
            state = next_state
            if _should_stop_loop(idx, terminated, termination_threshold):
                break
            idx += 1
            if max_episode_steps is not None and idx >= max_episode_steps:
                break
        final_stacks = info["stacks"][:, q_seat]
        stats.set(terminated.sum(), episode_reward_tensor, (final_stacks - initial_stacks).sum())
        totals = (stats.all_reduce_async() if reduce_stats else stats).wait()
        reward_scores.append(float(totals[1].item()))
        scores.append(float(totals[2].item()))
        total_steps += n_games * idx

    end_time = time.time()
    elapsed = end_time - start_time
    summary = {"env": config.get("ENV_ID", "Pulse-Poker-GPU-v1"), "total_steps": total_steps, "start_time": start_time,
               "end_time": end_time, "total_training_seconds": elapsed, "sps": total_steps / elapsed if elapsed > 0 else 0.0,
               "episode_rewards": reward_scores, "episode_profits": scores, "config": dict(config)}
    if plotter is not None and results_dir is not None:
        plotter.plot_learning_curve(scores=reward_scores, file_path=str(Path(results_dir) / "rewards_learning_curve"), window_size=10,
                                    title="Poker Q-Learning - Total Reward per Episode Batch")
        plotter.plot_learning_curve(scores=scores, file_path=str(Path(results_dir) / "total_chips_curve"), window_size=10,
                                    title="Poker Q-Learning - Total Chip Profit per Episode Batch")
    if benchmarker is not None:
        benchmarker.create_benchmark_file(env_name=summary["env"], episodes_return=reward_scores, start_time=start_time,
                                          end_time=end_time, total_steps=total_steps, config=config)
    return summary


def train_agent_fused(env, agents, agent_types, episodes, n_games, device, results_dir=None, config=None, plotter=None,
                      benchmarker=None, max_episode_steps=None, reduce_stats=True, stop_rule="lagged", host_seed=0,
                      step_hook=None, learner="native", hand_metrics=None, prefixed_decks=None, fuse_act_step=False):
    """train_agent with nothing in the step waiting on the host: the loop contract above (rotation, masks evaluated
    before `terminated |= dones`, stop cadence, step accounting) on four launches-groups per step --
      learner's actions (pulse_qnet_act, masked by seat) -> scripted opponents + env step (pulse_poker_policy_step) ->
      the learner's update (`learner="native"`: train_step_native, three HIP launches; `"torch"`: train_step_masked on
      PyTorch autograd) -> every 5th step the lagged done-count.
    `active_players` is drawn from a host RNG (the reference reads a device randint back, PokerGPU.py:76-77) and the
    stop rule is decided one check point late, on the done count of the WHOLE job (stoprule.py: with one process per
    GPU the count is all-reduced on a side stream, so every rank ends every episode at the same step and the learner's
    gradient all-reduces and the episode statistics stay aligned); `stop_rule="sync"` restores the reference's blocking
    check (lag 0), `"steps"` runs every episode to `max_episode_steps`.  Needs a learner with `act_into` /
    `train_step_masked` (qnetwork.py).
    `hand_metrics` (utils.performance.HandMetrics) adds the BB/100 side-channel of trainGPU_performance.py:192-206 as
    one more launch per step; its per-episode summaries come back under "hand_metrics"."""
    import random

    from ..environments.Poker.utils import native_types
    from ..stoprule import LaggedDoneCount
    config = config or {}
    q_agent_idx = agent_types.index(PokerAgentType.QLEARNING)
    q_agent = agents[q_agent_idx]
    if not (hasattr(q_agent, "act_into") and hasattr(q_agent, "train_step_masked") and hasattr(q_agent, "train_step_native")):
        raise TypeError("train_agent_fused needs a learner with act_into / train_step_native (PokerQNetwork)")
    native = learner == "native"
    host_rng = random.Random(host_seed)
    if stop_rule not in ("lagged", "sync", "steps"):
        raise ValueError(f"stop_rule must be 'lagged', 'sync' or 'steps', got {stop_rule!r}")
    if stop_rule == "steps" and max_episode_steps is None:
        raise ValueError("stop_rule='steps' needs max_episode_steps")
    done_count = LaggedDoneCount(device, n_games, TERMINATION_THRESHOLD, lag=0 if stop_rule == "sync" else 1)
    # the done tables of a check point are counted by the step's own launch where that launch takes the rule (act_policy_step)
    fused_count = native and stop_rule != "steps" and hasattr(env, "act_policy_step") and done_count.handle is not None and done_count.exchange != "host"
    fuse_act_step = bool(fuse_act_step) and native and hasattr(env, "act_policy_step")
    actions = torch.zeros(n_games, dtype=torch.long, device=device)
    # the env alternates two observation buffers from here on: the tensor a step returned is still intact while the
    # next step runs, so the pre-step observation the learner needs is simply the previous `state` (no copy per step)
    double_buffered = hasattr(env, "double_buffer_obs")
    if double_buffered:
        env.double_buffer_obs = True
    state_before = None if double_buffered else torch.empty((n_games, env.obs_size), dtype=torch.float32, device=device)
    terminated = torch.zeros(n_games, dtype=torch.bool, device=device)
    active_games = torch.zeros(n_games, dtype=torch.bool, device=device)
    episode_reward = torch.zeros((), dtype=torch.float64, device=device)
    stats = EpisodeStats(device)
    total_steps, global_step = 0, 0
    scores, reward_scores, episode_metrics = [], [], []
    pending = []

    def drain_pending():
        if pending:
            host = torch.stack(pending).cpu()                                         # one read-back for all of them
            reward_scores.extend(float(x) for x in host[:, 1])
            scores.extend(float(x) for x in host[:, 2])
            pending.clear()
    start_time = time.time()
    for episode in range(episodes):
        _, rotated_types, q_seat, rotations = get_rotated_agents(agents, agent_types, episode_idx=episode, q_agent_idx=q_agent_idx)
        native_seats = native_types(rotated_types)               # the learner's seat is EXTERNAL: its action is taken as given
        A = host_rng.randint(2, env.n_players)
        options = {"rotation": rotations, "active_players": int(A), "q_agent_seat": q_seat}
        if prefixed_decks is not None:                           # USE_PREFIXED_DECKS: episode -> int32[n_games, 52] (trainGPU_performance.py:52)
            options["prefixed_decks"] = prefixed_decks(episode)
        state, info = env.reset(options=options)
        initial_stacks = info["stacks"][:, q_seat].clone()
        terminated.zero_()
        episode_reward.zero_()
        done_count.drain()
        if hand_metrics is not None:
            hand_metrics.begin_episode(env, q_seat)
        idx = 0
        while True:
            seat_idx = info["seat_idx"]
            if double_buffered:
                state_before = state
            else:
                state_before.copy_(state)                                             # the env reuses its obs buffer
            if native:
                # active_games = q_mask & ~terminated (trainGPU.py:85) comes out of the act launch, and so do the lists of the
                # rows the training launch will take (it trains on this very observation); `terminated |= dones` (:86) and
                # the episode reward (:96) ride on the training launches
                # every CHECK_INTERVAL-th step the step's launch also counts the done tables for the stop rule (no launch of its own)
                check = fused_count and idx % CHECK_INTERVAL == 0
                if fuse_act_step:
                    # ... the act launch IS the env step's launch (PokerGPU.act_policy_step: the workgroup that picks the actions
                    # of 128 tables steps them itself) -- measured no faster than the two launches: opt-in
                    next_state, rewards, dones, _, info = env.act_policy_step(q_agent, q_seat, native_seats, actions, global_step, state_before,
                                                                              seat_idx, terminated, active_games, stop_rule=done_count if check else None)
                else:
                    q_agent.act_into(state_before, seat_idx, q_seat, actions, step_counter=global_step, terminated=terminated,
                                     row_mask_out=active_games, select_for_training=True)
                    if check:
                        next_state, rewards, dones, _, info = env.rollout(native_seats, actions, 1, global_step, stop_rule=done_count)
                    else:
                        next_state, rewards, dones, _, info = env.policy_step(native_seats, actions, global_step)
                if hand_metrics is not None:
                    hand_metrics.update(env, dones, terminated)                       # before terminated |= dones
                q_agent.train_step_native(state_before, actions, rewards, next_state, dones, active_games, step_counter=global_step,
                                          terminated=terminated, reward_sum=episode_reward)
            else:
                torch.logical_and(seat_idx == q_seat, ~terminated, out=active_games)      # :85, before the step
                q_agent.act_into(state, seat_idx, q_seat, actions, step_counter=global_step)
                next_state, rewards, dones, _, info = env.policy_step(native_seats, actions, global_step)
                if hand_metrics is not None:
                    hand_metrics.update(env, dones, terminated)
                terminated |= dones                                                   # :86
                q_agent.train_step_masked(state_before, actions, rewards, next_state, dones, active_games)
                episode_reward += (rewards * active_games).sum()                      # :96
            if step_hook is not None:
                step_hook(episode, idx, state_before, actions, rewards, next_state, dones, active_games)
            state = next_state
            global_step += 1
            if idx % CHECK_INTERVAL == 0 and stop_rule != "steps":                    # :27-33 cadence
                if fused_count:
                    done_count.publish()                                              # (not with the next check point's launch: the verdict due then must not wait)
                if not fused_count:                                                   # (else: counted by the step's launch -- the env's done
                    done_count.submit(terminated)                                     #  flags ARE `terminated`: they never clear inside an episode)
                if done_count.over():                                                 # the same verdict on every rank
                    break
            idx += 1
            if max_episode_steps is not None and idx >= max_episode_steps:
                break
        final_stacks = info["stacks"][:, q_seat]
        stats.set(terminated.sum(), episode_reward, (final_stacks - initial_stacks).sum())
        totals = (stats.all_reduce_async() if reduce_stats else stats).wait()
        # the episode's sums stay on the device; they are read back at the reference's reporting cadence (every REPORT_EVERY-th
        # episode, trainGPU.py:110) and at the end -- a read-back per episode stalls the queue ~80 us at every boundary
        pending.append(totals.clone())
        if native:
            q_agent.check_native_report(wait=False)                                   # an update called off inside its launch must not pass silently
        if len(pending) >= REPORT_EVERY:
            drain_pending()
        if hand_metrics is not None:
            episode_metrics.append(hand_metrics.end_episode())
        total_steps += done_count.n_global * idx                                      # :108 (the tables of the whole job: every rank steps in lock step)

    drain_pending()
    if native:
        q_agent.check_native_report()
    torch.cuda.synchronize(device)
    done_count.close()
    if double_buffered:
        env.double_buffer_obs = False
    end_time = time.time()
    elapsed = end_time - start_time
    summary = {"env": config.get("ENV_ID", "Pulse-Poker-GPU-v1"), "total_steps": total_steps, "start_time": start_time,
               "end_time": end_time, "total_training_seconds": elapsed, "sps": total_steps / elapsed if elapsed > 0 else 0.0,
               "episode_rewards": reward_scores, "episode_profits": scores, "config": dict(config), "env_step_calls": global_step}
    if hand_metrics is not None:
        summary["hand_metrics"] = {"episodes": episode_metrics, "final": hand_metrics.summary()}
    if plotter is not None and results_dir is not None:
        plotter.plot_learning_curve(scores=reward_scores, file_path=str(Path(results_dir) / "rewards_learning_curve"), window_size=10,
                                    title="Poker Q-Learning - Total Reward per Episode Batch")
        plotter.plot_learning_curve(scores=scores, file_path=str(Path(results_dir) / "total_chips_curve"), window_size=10,
                                    title="Poker Q-Learning - Total Chip Profit per Episode Batch")
    if benchmarker is not None:
        benchmarker.create_benchmark_file(env_name=summary["env"], episodes_return=reward_scores, start_time=start_time,
                                          end_time=end_time, total_steps=total_steps, config=config)
    return summary


def main(argv=None):
    """python -m pulselib_amd.scripts.trainGPU [--config PATH] [--tables N] [--episodes E] [--results DIR]: the reference's
    entry point (scripts/Poker/trainGPU.py:148-214) on the fused loop.  The configuration is the reference's own schema
    (config/pokerGPU.yaml, flat keys; utils/config.py) -- its file can be passed as it is -- plus the optional engine keys
    N_GPUS / SEED / USE_PREFIXED_DECKS / MAX_EPISODE_STEPS.  Writes the run summary results/PokerGPU/runs/run_N.yaml
    (utils/benchmarking.py: YamlBenchmarker, the reference writer's keys and numbering) and poker_qnet_final.pth."""
    import argparse
    import os

    from ..environments.Poker import PokerGPU, PokerQNetwork, load_gpu_agents
    from ..sharding import shard_tables
    from ..utils.benchmarking import YamlBenchmarker, result_folder_for
    from ..utils.config import get_config_file, poker_gpu_arguments
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="pokerGPU.yaml", help="file name under pulselib_amd/config/, or a path (e.g. the reference's config/pokerGPU.yaml)")
    ap.add_argument("--tables", type=int, default=None, help="override N_GAMES")
    ap.add_argument("--episodes", type=int, default=None, help="override EPISODES")
    ap.add_argument("--results", type=Path, default=Path.cwd() / "results")
    args = ap.parse_args(argv)
    config = get_config_file(args.config)
    if config is None:
        raise FileNotFoundError(f"no configuration file {args.config!r}")
    if args.tables is not None:
        config["N_GAMES"] = int(args.tables)
    if args.episodes is not None:
        config["EPISODES"] = int(args.episodes)
    a = poker_gpu_arguments(config)
    engine = a["engine"]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:                                           # launched as one process per GPU (torchrun): N_GPUS must agree
        import torch.distributed as dist
        if int(engine["N_GPUS"]) != world:
            raise ValueError(f"N_GPUS is {engine['N_GPUS']} but the job has {world} ranks")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        if not dist.is_initialized():
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    elif int(engine["N_GPUS"]) != 1:
        raise ValueError(f"N_GPUS is {engine['N_GPUS']}: start one process per GPU (python -m torch.distributed.run --nproc-per-node {engine['N_GPUS']} -m pulselib_amd.scripts.trainGPU ...)")
    device = torch.device("cuda", torch.cuda.current_device())
    n_local, table_id0 = shard_tables(int(a["env"]["n_games"]), world, rank)
    agents, types = load_gpu_agents(device, *a["agents_args"])                                    # trainGPU.py:156-162
    results_dir = result_folder_for(a["env_id"], args.results)
    q_net = PokerQNetwork(weights_path=results_dir / "poker_qnet_final.pth", device=device, seed=int(engine["SEED"]), **a["q_network"])
    agents.insert(0, q_net)                                                                       # :174-175
    types.insert(0, PokerAgentType.QLEARNING)
    env = PokerGPU(device=device, agents=agents, seed=int(engine["SEED"]), table_id0=table_id0, **dict(a["env"], n_games=n_local))   # :177-188
    benchmarker = YamlBenchmarker.from_config(a["benchmarking"])                                  # :153
    benchmarker.results_dir_resolver = lambda env_name: results_dir
    if rank != 0:
        benchmarker.enabled = False
    prefixed = None
    if engine["USE_PREFIXED_DECKS"]:
        from ..utils.performance import build_prefixed_deck_batch
        prefixed = lambda episode: build_prefixed_deck_batch(n_games=n_local, seed=int(engine["SEED"]) + episode, device=device)   # noqa: E731
    out = train_agent_fused(env, agents, types, int(a["train"]["episodes"]), n_local, device, results_dir=results_dir, config=config,
                            benchmarker=benchmarker, max_episode_steps=engine["MAX_EPISODE_STEPS"], prefixed_decks=prefixed)
    if rank == 0:
        torch.save(q_net.network.state_dict(), results_dir / "poker_qnet_final.pth")             # trainGPU.py:118
        print({k: out[k] for k in ("total_steps", "total_training_seconds", "sps")})
    return out


if __name__ == "__main__":
    main()
