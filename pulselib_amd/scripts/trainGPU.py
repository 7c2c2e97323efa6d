"""Episode driver with the reference trainer's loop contract (scripts/Poker/trainGPU.py:36-145):

  * seat rotation per episode (get_rotated_agents), reset options {rotation, active_players, q_agent_seat};
  * per step: actions.fill_(0) -> build_actions -> env.step -> learner.train_step on
    `q_mask & ~terminated` evaluated BEFORE `terminated |= dones` (trainGPU.py:85-86);
  * stop rule every 5th step at > 80 % terminated (trainGPU.py:27-33); `total_steps += n_games * idx`
    (trainGPU.py:108) -- finished tables count, as in every number the reference publishes;
  * run summary with the reference's keys (`sps`, `total_steps`, ...; utils/benchmarking/benchmarking.py:84-100).

The learner is whatever object sits at the QLEARNING seat (`get_actions`, `train_step`): the reference's
PokerQNetwork runs unchanged on PyTorch-ROCm; `SimpleQNetwork` below is a minimal stand-in with the same
interface.  With N > 1 ranks each process drives its own shard and episode sums are all-reduced (RCCL)."""
from __future__ import annotations

import time
from pathlib import Path

import torch

from ..environments.Poker.utils import PokerAgentType, build_actions, get_rotated_agents
from ..sharding import EpisodeStats

CHECK_INTERVAL = 5            # trainGPU.py:31
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
                      step_hook=None, learner="native"):
    """train_agent with nothing in the step waiting on the host: the loop contract above (rotation, masks evaluated
    before `terminated |= dones`, stop cadence, step accounting) on four launches-groups per step --
      learner's actions (pulse_qnet_act, masked by seat) -> scripted opponents + env step (pulse_poker_policy_step) ->
      the learner's update (`learner="native"`: train_step_native, three HIP launches; `"torch"`: train_step_masked on
      PyTorch autograd) -> every 5th step the lagged done-count.
    `active_players` is drawn from a host RNG (the reference reads a device randint back, PokerGPU.py:76-77) and the
    stop rule is decided on the newest count that already reached the host (stoprule.py); `stop_rule="sync"`
    restores the reference's blocking check.  Needs a learner with `act_into` / `train_step_masked` (qnetwork.py)."""
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
    done_count = LaggedDoneCount(device, n_games, TERMINATION_THRESHOLD)
    actions = torch.zeros(n_games, dtype=torch.long, device=device)
    state_before = torch.empty((n_games, env.obs_size), dtype=torch.float32, device=device)
    terminated = torch.zeros(n_games, dtype=torch.bool, device=device)
    active_games = torch.zeros(n_games, dtype=torch.bool, device=device)
    episode_reward = torch.zeros((), dtype=torch.float64, device=device)
    stats = EpisodeStats(device)
    total_steps, global_step = 0, 0
    scores, reward_scores = [], []
    start_time = time.time()
    for episode in range(episodes):
        _, rotated_types, q_seat, rotations = get_rotated_agents(agents, agent_types, episode_idx=episode, q_agent_idx=q_agent_idx)
        native_seats = native_types(rotated_types)               # the learner's seat is EXTERNAL: its action is taken as given
        A = host_rng.randint(2, env.n_players)
        state, info = env.reset(options={"rotation": rotations, "active_players": int(A), "q_agent_seat": q_seat})
        initial_stacks = info["stacks"][:, q_seat].clone()
        terminated.zero_()
        episode_reward.zero_()
        done_count.drain()
        idx = 0
        while True:
            seat_idx = info["seat_idx"]
            state_before.copy_(state)                                                 # the env reuses its obs buffer
            if native:
                # active_games = q_mask & ~terminated (trainGPU.py:85) comes out of the act launch; `terminated |= dones`
                # (:86) and the episode reward (:96) ride on the training launches
                q_agent.act_into(state, seat_idx, q_seat, actions, step_counter=global_step, terminated=terminated,
                                 row_mask_out=active_games)
                next_state, rewards, dones, _, info = env.policy_step(native_seats, actions, global_step)
                q_agent.train_step_native(state_before, actions, rewards, next_state, dones, active_games, step_counter=global_step,
                                          terminated=terminated, reward_sum=episode_reward)
            else:
                torch.logical_and(seat_idx == q_seat, ~terminated, out=active_games)      # :85, before the step
                q_agent.act_into(state, seat_idx, q_seat, actions, step_counter=global_step)
                next_state, rewards, dones, _, info = env.policy_step(native_seats, actions, global_step)
                terminated |= dones                                                   # :86
                q_agent.train_step_masked(state_before, actions, rewards, next_state, dones, active_games)
                episode_reward += (rewards * active_games).sum()                      # :96
            if step_hook is not None:
                step_hook(episode, idx, state_before, actions, rewards, next_state, dones, active_games)
            state = next_state
            global_step += 1
            if idx % CHECK_INTERVAL == 0:                                             # :27-33 cadence
                done_count.submit(terminated)
                if done_count.over(blocking=stop_rule == "sync"):
                    break
            idx += 1
            if max_episode_steps is not None and idx >= max_episode_steps:
                break
        final_stacks = info["stacks"][:, q_seat]
        stats.set(terminated.sum(), episode_reward, (final_stacks - initial_stacks).sum())
        totals = (stats.all_reduce_async() if reduce_stats else stats).wait()
        host = totals.cpu()                                                           # one read-back per episode
        reward_scores.append(float(host[1]))
        scores.append(float(host[2]))
        total_steps += n_games * idx                                                  # :108

    torch.cuda.synchronize(device)
    end_time = time.time()
    elapsed = end_time - start_time
    summary = {"env": config.get("ENV_ID", "Pulse-Poker-GPU-v1"), "total_steps": total_steps, "start_time": start_time,
               "end_time": end_time, "total_training_seconds": elapsed, "sps": total_steps / elapsed if elapsed > 0 else 0.0,
               "episode_rewards": reward_scores, "episode_profits": scores, "config": dict(config), "env_step_calls": global_step}
    if plotter is not None and results_dir is not None:
        plotter.plot_learning_curve(scores=reward_scores, file_path=str(Path(results_dir) / "rewards_learning_curve"), window_size=10,
                                    title="Poker Q-Learning - Total Reward per Episode Batch")
        plotter.plot_learning_curve(scores=scores, file_path=str(Path(results_dir) / "total_chips_curve"), window_size=10,
                                    title="Poker Q-Learning - Total Chip Profit per Episode Batch")
    if benchmarker is not None:
        benchmarker.create_benchmark_file(env_name=summary["env"], episodes_return=reward_scores, start_time=start_time,
                                          end_time=end_time, total_steps=total_steps, config=config)
    return summary


class SimpleQNetwork(torch.nn.Module):
    """Minimal learner with the interface the driver needs (the reference's PokerQNetwork,
    environments/Poker/Player.py:178-298, is the real one and runs unchanged on PyTorch-ROCm)."""

    def __init__(self, device, gamma=0.95, state_dim=40, action_dim=13, lr=2e-4, epsilon=0.1):
        super().__init__()
        self.device, self.gamma, self.epsilon = device, gamma, epsilon
        self.network = torch.nn.Sequential(torch.nn.Linear(state_dim, 128), torch.nn.GELU(), torch.nn.Linear(128, 64), torch.nn.GELU(),
                                           torch.nn.Linear(64, action_dim)).to(device)
        self.optimizer = torch.optim.AdamW(self.parameters(), lr=lr)

    def get_actions(self, states):
        with torch.inference_mode():
            greedy = self.network(states).argmax(dim=1)
            explore = torch.rand(states.shape[0], device=states.device) < self.epsilon
            return torch.where(explore, torch.randint(0, 13, (states.shape[0],), device=states.device), greedy)

    def train_step(self, states, actions, rewards, next_states, dones):
        valid = (states[:, 12] == 0) | (states[:, 12] == 2)           # Player.py:261
        if not valid.any():
            return 0.0
        states, actions, rewards, next_states, dones = states[valid], actions[valid], rewards[valid], next_states[valid], dones[valid]
        q = self.network(states).gather(1, actions.unsqueeze(1)).squeeze(1)
        with torch.no_grad():
            target = rewards + self.gamma * self.network(next_states).max(dim=1).values * (~dones).float()
        loss = torch.nn.functional.mse_loss(q, target)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.parameters(), max_norm=1.0)
        self.optimizer.step()
        return loss


def main():
    import yaml
    from ..environments.Poker import PokerGPU, load_gpu_agents
    cfg = yaml.safe_load((Path(__file__).resolve().parent.parent / "config" / "pokerGPU.yaml").read_text())
    env_cfg, rew, learner = cfg["env"], cfg["reward"], cfg["learner"]
    device = torch.device("cuda", torch.cuda.current_device())
    n_games = int(env_cfg["tables"])
    agents, types = load_gpu_agents(device, env_cfg["opponents"], list(cfg["opponent_mix"]), env_cfg["starting_stack_bb"], env_cfg["actions"])
    agents.insert(0, SimpleQNetwork(device, gamma=learner["gamma"], state_dim=env_cfg["observation_size"], lr=float(learner["learning_rate"])))
    types.insert(0, PokerAgentType.QLEARNING)
    env = PokerGPU(device=device, agents=agents, n_players=env_cfg["opponents"] + 1, n_games=n_games,
                   starting_bbs=env_cfg["starting_stack_bb"], w1=rew["w1"], w2=rew["w2"], K=rew["K"], alpha=rew["alpha"],
                   seed=env_cfg.get("seed", 0))
    out = train_agent(env, agents, types, int(cfg["run"]["episodes"]), n_games, device, config={"ENV_ID": env_cfg["id"], **cfg["run"]},
                      max_episode_steps=env_cfg.get("max_episode_steps"))
    print({k: out[k] for k in ("total_steps", "total_training_seconds", "sps")})


if __name__ == "__main__":
    main()
