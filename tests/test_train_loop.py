"""Loop contract of the episode driver (pulselib_amd/scripts/trainGPU.py) pinned on a scripted environment,
as the reference pins its own trainer (tests/poker/test_train_gpu_inner_loop.py: ScriptedEnv + fake agents).
Runs on CPU tensors: the driver itself is device-agnostic glue around reset/step."""
import torch

from pulselib_amd.environments.Poker.utils import PokerAgentType
from pulselib_amd.scripts import trainGPU as trainer


class ScriptedEnv:
    def __init__(self, episodes):
        self.episodes, self.episode_idx, self.step_idx, self.step_calls, self.reset_options = episodes, -1, 0, 0, []

    def reset(self, options):
        self.reset_options.append(dict(options))
        self.episode_idx += 1
        self.step_idx = 0
        ep = self.episodes[self.episode_idx]
        return ep["state"].clone(), {k: v.clone() for k, v in ep["info"].items()}

    def step(self, actions):
        self.step_calls += 1
        st = self.episodes[self.episode_idx]["steps"][self.step_idx]
        self.step_idx += 1
        return (st["next_state"].clone(), st["rewards"].clone(), st["dones"].clone(), torch.zeros_like(st["dones"]),
                {k: v.clone() for k, v in st["info"].items()})


class RecordingQ:
    def __init__(self):
        self.calls = []

    def get_actions(self, states):
        return torch.zeros(states.shape[0], dtype=torch.long)

    def train_step(self, states, actions, rewards, next_states, dones):
        self.calls.append(dict(states=states.clone(), actions=actions.clone(), rewards=rewards.clone(),
                               next_states=next_states.clone(), dones=dones.clone()))


def _state(rows):
    return torch.tensor([[r] + [0.0] * 12 for r in rows], dtype=torch.float32)


def _step(rows, rewards, dones, seats, stacks):
    return {"next_state": _state(rows), "rewards": torch.tensor(rewards, dtype=torch.float32), "dones": torch.tensor(dones),
            "info": {"seat_idx": torch.tensor(seats), "stacks": torch.tensor(stacks, dtype=torch.float32)}}


def _fake_build_actions(state, actions, curr_players, agents, agent_types, device):
    actions.copy_(curr_players.long() + 1)


def test_masks_stop_rule_and_step_accounting():
    n = 5
    stacks0 = [[100.0, 100.0, 100.0]] * n
    steps = [
        # idx 0: Q seat (0) acts on games 0,1; game 1 finishes now -> still trained on (mask taken before |= dones)
        _step([10, 11, 12, 13, 14], [1.0, 2.0, 3.0, 4.0, 5.0], [False, True, False, False, False], [0, 0, 0, 1, 1], stacks0),
        # idx 1: Q seat acts on games 0,1,2 but game 1 is already terminated -> excluded
        _step([20, 21, 22, 23, 24], [0.5, 9.0, 1.5, 0.0, 0.0], [True, True, True, True, False], [1, 1, 1, 1, 1], stacks0),
    ] + [_step([30 + k] * n, [0.0] * n, [True] * n, [1] * n, [[90.0, 100.0, 110.0]] * n) for k in range(4)]
    episode = {"state": _state([0, 1, 2, 3, 4]), "info": {"seat_idx": torch.tensor([0, 0, 1, 1, 2]),
                                                        "stacks": torch.tensor(stacks0, dtype=torch.float32)}, "steps": steps}
    q = RecordingQ()
    agents = [q, object(), object()]
    types = [PokerAgentType.QLEARNING, PokerAgentType.HEURISTIC_HANDS, PokerAgentType.RANDOM]
    env = ScriptedEnv([episode])
    out = trainer.train_agent(env, agents, types, episodes=1, n_games=n, device=torch.device("cpu"),
                              build_actions_fn=_fake_build_actions, reduce_stats=False)
    # stop rule: checked at idx 0 (1/5 done: no) and idx 5 (all done: yes) -> six step() calls, idx ends at 5
    assert env.step_calls == 6
    assert out["total_steps"] == n * 5                                   # trainGPU.py:108
    assert env.reset_options[0] == {"rotation": 0, "active_players": True, "q_agent_seat": 0}
    # idx 0: games 0 and 1 (Q seat to act, not yet terminated)
    assert q.calls[0]["states"][:, 0].tolist() == [0.0, 1.0]
    assert q.calls[0]["actions"].tolist() == [1, 1]
    assert q.calls[0]["rewards"].tolist() == [1.0, 2.0]
    assert q.calls[0]["dones"].tolist() == [False, True]
    assert q.calls[0]["next_states"][:, 0].tolist() == [10.0, 11.0]
    # idx 1: seat_idx from step 0's info = [0,0,0,1,1]; game 1 terminated earlier -> games 0 and 2 only
    assert q.calls[1]["states"][:, 0].tolist() == [10.0, 12.0]
    assert q.calls[1]["rewards"].tolist() == [0.5, 1.5]
    assert len(q.calls) == 2                                             # later steps: Q seat never to act
    assert out["episode_rewards"] == [1.0 + 2.0 + 0.5 + 1.5]
    assert out["episode_profits"] == [n * (90.0 - 100.0)]


def test_seat_rotation_follows_episode_index():
    n = 2
    stacks = [[100.0, 100.0, 100.0]] * n
    ep = lambda: {"state": _state([0, 1]), "info": {"seat_idx": torch.tensor([0, 1]), "stacks": torch.tensor(stacks)},
                  "steps": [_step([5, 6], [0.0, 0.0], [True, True], [0, 1], stacks)]}
    env = ScriptedEnv([ep(), ep(), ep()])
    types = [PokerAgentType.QLEARNING, PokerAgentType.HEURISTIC_HANDS, PokerAgentType.RANDOM]
    trainer.train_agent(env, [RecordingQ(), object(), object()], types, episodes=3, n_games=n, device=torch.device("cpu"),
                        build_actions_fn=_fake_build_actions, reduce_stats=False)
    assert [o["q_agent_seat"] for o in env.reset_options] == [0, 1, 2]   # utils.py:173-183: Q seat = episode % n
    assert [o["rotation"] for o in env.reset_options] == [0, 1, 2]
