"""The learner of the Poker GPU trainer: `PokerQNetwork` with the reference's constructor, attribute and
method names (environments/Poker/Player.py:178-298), so checkpoints (`network.state_dict()`), the trainer
loop (scripts/Poker/trainGPU.py:57-108) and the tests' DummyPokerQNetwork subclasses carry over.

What is native here:
  * action selection -- `get_actions`, and the fused masked form `act_into` used by `build_actions` -- is one
    MFMA kernel (csrc/qnet.hip) reading the module's own weight tensors; no mask gather / scatter, no
    host sync, draws keyed by (seed, global table id, step) like the scripted opponents';
  * `train_step_native`: the whole update -- row filter, forward in train mode, TD target, backward, gradient
    clipping, AdamW, target sync -- as two launches of csrc/qnet.hip (three without act_into's row lists) with no host sync (the reference's boolean
    indexing costs a device->host sync per mask);
  * `train_step_masked`: the same sync-free contract on PyTorch-ROCm autograd (every row goes through with a 0/1
    weight), kept as the torch cross-check of the native path.
`train_step` (the reference's signature) runs on the native kernels too; `train_step_torch` keeps the reference's
filtering semantics and op sequence verbatim (Player.py:255-294) on PyTorch autograd."""
from __future__ import annotations

import copy
import ctypes as C
from pathlib import Path

import torch
import torch.nn as nn

from ... import _native

HIDDEN = (128, 128, 64, 32)      # Player.py:189-201
TRAIN_BLOCKS = 256               # persistent workgroups of the training kernel: one per CU of an MI355X


def build_network(state_dim: int, action_dim: int) -> nn.Sequential:
    """Same module indices as the reference's Sequential (0,2,5,8,10 Linear; 4,7 Dropout), so state_dicts interchange."""
    h1, h2, h3, h4 = HIDDEN
    return nn.Sequential(
        nn.Linear(state_dim, h1), nn.GELU(),
        nn.Linear(h1, h2), nn.GELU(), nn.Dropout(.1),
        nn.Linear(h2, h3), nn.GELU(), nn.Dropout(0.1),
        nn.Linear(h3, h4), nn.GELU(),
        nn.Linear(h4, action_dim),
    )


LINEAR_INDICES = (0, 2, 5, 8, 10)


class PokerQNetwork(nn.Module):
    def __init__(self, weights_path, device, gamma, update_freq: int, epsilon=.1, epsilon_end=.001, epsilon_decay=.99999,
                 state_dim=27, action_dim=13, hidden_dim=256, learning_rate=1e-3, weight_decay=1e-3, seed=0, table_id0=0):
        super().__init__()
        del hidden_dim                       # accepted and unused, as in the reference (:179)
        self.update_freq = update_freq
        self.device = device
        self.gamma = gamma
        self.epsilon = epsilon
        self.epsilon_decay = epsilon_decay
        self.epsilon_end = epsilon_end
        self.state_dim, self.action_dim = int(state_dim), int(action_dim)
        self.network = build_network(self.state_dim, self.action_dim)
        if weights_path and Path(weights_path).exists():
            self.network.load_state_dict(torch.load(weights_path, map_location=device, weights_only=True))
        self.target_network = copy.deepcopy(self.network)
        self.target_network.eval()
        self.lr = float(learning_rate)
        self.wd = float(weight_decay)
        self.step_count = 0
        self.criterion = nn.MSELoss()
        self.to(device)
        self._native = None
        self._flat = self._flat_target = None
        if torch.device(device).type == "cuda":
            self._flatten()                   # parameters become views of one flat buffer per network (pulse_env.h: PulseQNetTrain)
        self.optimizer = self.configure_optimizers()
        # Philox keys of the exploration draws (act_into takes the env step counter; get_actions counts its calls)
        self.seed, self.table_id0 = int(seed), int(table_id0)
        self._calls = 0
        self._struct_cache = {}
        self._lr_gate = None
        self.data_parallel = True            # under torch.distributed with > 1 rank: all-reduce the gradient every step
        # the reduce launch applies AdamW itself behind a meeting of its workgroups (qnet.hip); True = AdamW as its own launch
        self.separate_apply = False
        self.meet_wait_ticks = 0             # 0: the library's 5 s (ticks of the 100 MHz clock)
        self._meetings_called_off = int(_native.lib().pulse_qnet_called_off_meetings()) if torch.device(device).type == "cuda" else 0

    # ------------------------------------------------------------------ torch side
    def forward(self, states):
        return self.network(states)

    def configure_optimizers(self):
        on_gpu = torch.device(self.device).type == "cuda"
        return torch.optim.AdamW(self.parameters(), lr=self.lr, weight_decay=self.wd, fused=on_gpu)

    # ------------------------------------------------------------------ native action selection
    def _net_struct(self, net: nn.Sequential) -> _native.QNet:
        # the parameters are views of the flat buffers for the life of the module (cuda): their addresses are cached
        key = "target" if net is self.target_network else "net"
        cached = self._struct_cache.get(key)
        if cached is not None and cached[0] == net[LINEAR_INDICES[0]].weight.data_ptr():
            return cached[1]
        s = _native.QNet()
        s.state_dim, s.n_actions = self.state_dim, self.action_dim
        for i, li in enumerate(LINEAR_INDICES, start=1):
            lin = net[li]
            w, b = lin.weight, lin.bias
            if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and b.is_contiguous()):
                raise RuntimeError("PokerQNetwork: action selection runs on the MI355X only (fp32 weights on a cuda device); "
                                   "there is no CPU path")
            setattr(s, f"w{i}", w.data_ptr())
            setattr(s, f"b{i}", b.data_ptr())
        self._struct_cache[key] = (net[LINEAR_INDICES[0]].weight.data_ptr(), s)
        return s

    @staticmethod
    def _rows(states):
        if not (states.is_cuda and states.dtype == torch.float32 and states.dim() == 2 and states.stride(1) == 1):
            if not states.is_cuda:
                raise RuntimeError("PokerQNetwork: states must live on the GPU (no CPU path)")
            states = states.to(torch.float32).contiguous()
        return states

    def _decay_epsilon(self):
        self.epsilon = max(self.epsilon * self.epsilon_decay, self.epsilon_end)      # Player.py:243

    def q_values(self, states, target: bool = False):
        """network(states) in eval mode through the HIP kernel -> fp32[n, action_dim] (no autograd)."""
        states = self._rows(states)
        out = torch.empty((states.shape[0], self.action_dim), dtype=torch.float32, device=states.device)
        net = self._net_struct(self.target_network if target else self.network)
        _native.check(_native.lib().pulse_qnet_forward(C.byref(net), states.data_ptr(), states.stride(0), states.shape[0],
                                                       out.data_ptr(), _native.current_stream(states.device)), "pulse_qnet_forward")
        return out

    def get_actions(self, states):
        """Player.py:242-253: epsilon-greedy actions for every row of `states` -> int64[n]."""
        self._decay_epsilon()
        states = self._rows(states)
        actions = torch.empty(states.shape[0], dtype=torch.long, device=states.device)
        self._calls += 1
        net = self._net_struct(self.network)
        _native.check(_native.lib().pulse_qnet_act(C.byref(net), states.data_ptr(), states.stride(0), states.shape[0], None, 0,
                                                   float(self.epsilon), self.seed & (2**64 - 1), (1 << 40) + self._calls,
                                                   self.table_id0, actions.data_ptr(), None, None, None,
                                                   _native.current_stream(states.device)), "pulse_qnet_act")
        return actions

    def act_into(self, states, curr_players, q_seat: int, actions, step_counter=None, terminated=None, row_mask_out=None,
                 select_for_training=False):
        """`actions[mask] = self.get_actions(states[mask])` for mask = (curr_players == q_seat) (utils.py:113-119) as one
        launch: rows of other seats are not touched, nothing is gathered, nothing syncs.  `row_mask_out` (bool/uint8[n]):
        also receives `(curr_players == q_seat) & ~terminated` for every row, the trainer's mask (trainGPU.py:85).
        select_for_training: the next train_step_native on these `states` with this row_mask_out skips its row-selection
        launch (the lists are written here: pulse_qnet_act_select)."""
        self._decay_epsilon()
        states = self._rows(states)
        if actions.dtype != torch.int64 or not actions.is_contiguous():
            raise ValueError("actions must be a contiguous int64 tensor (it is written in place)")
        curr = curr_players if (curr_players.dtype == torch.int32 and curr_players.is_contiguous()) else curr_players.to(torch.int32).contiguous()
        self._calls += 1
        step = (1 << 40) + self._calls if step_counter is None else int(step_counter)
        net = self._net_struct(self.network)
        n = states.shape[0]
        self._act_selected = None
        if select_for_training and row_mask_out is not None and n > 0:
            self._native_state(n)
            scratch = self._native["select"]
            _native.check(_native.lib().pulse_qnet_act_select(
                C.byref(net), states.data_ptr(), states.stride(0), n, curr.data_ptr(), int(q_seat), float(self.epsilon),
                self.seed & (2**64 - 1), step, self.table_id0, actions.data_ptr(), None if terminated is None else terminated.data_ptr(),
                row_mask_out.data_ptr(), scratch.data_ptr(), scratch.numel(), _native.current_stream(states.device)), "pulse_qnet_act_select")
            self._act_selected = (states.data_ptr(), states.stride(0), row_mask_out.data_ptr(), n)
            return actions
        _native.check(_native.lib().pulse_qnet_act(C.byref(net), states.data_ptr(), states.stride(0), n,
                                                   curr.data_ptr(), int(q_seat), float(self.epsilon), self.seed & (2**64 - 1), step,
                                                   self.table_id0, actions.data_ptr(), None,
                                                   None if terminated is None else terminated.data_ptr(),
                                                   None if row_mask_out is None else row_mask_out.data_ptr(),
                                                   _native.current_stream(states.device)), "pulse_qnet_act")
        return actions

    def begin_fused_act(self, states, row_mask_out, n):
        """The host side of act_into(select_for_training=True) for a launch that does the acting itself
        (PokerGPU.act_policy_step): epsilon decay, call count, the row lists' scratch, and the note that the next
        train_step_native on these states / this mask finds its row lists written.  Returns (network struct, scratch)."""
        self._decay_epsilon()
        self._calls += 1
        self._native_state(n)
        scratch = self._native["select"]
        self._act_selected = (states.data_ptr(), states.stride(0), row_mask_out.data_ptr(), n)
        return self._net_struct(self.network), scratch

    # ------------------------------------------------------------------ native learning (csrc/qnet.hip)
    def _flatten(self):
        flats = []
        for net in (self.network, self.target_network):
            n = sum(p.numel() for li in LINEAR_INDICES for p in (net[li].weight, net[li].bias))
            flat = torch.empty(n, dtype=torch.float32, device=self.device)
            o = 0
            for li in LINEAR_INDICES:
                for p in (net[li].weight, net[li].bias):
                    view = flat[o:o + p.numel()].view_as(p)
                    view.copy_(p.data)
                    p.data = view
                    o += p.numel()
            flats.append(flat)
        self._flat, self._flat_target = flats

    def _native_state(self, n_rows=0):
        nat = self._native
        if nat is None:
            if getattr(self, "_flat", None) is None:
                raise RuntimeError("PokerQNetwork: native training runs on the MI355X only (construct it on a cuda device)")
            dev, n = self._flat.device, self._flat.numel()
            assert n == _native.lib().pulse_qnet_param_count(self.state_dim, self.action_dim)
            nat = self._native = {
                "grad": torch.zeros(n, dtype=torch.float32, device=dev),
                "m": torch.zeros(n, dtype=torch.float32, device=dev),
                "v": torch.zeros(n, dtype=torch.float32, device=dev),
                "step": torch.zeros(1, dtype=torch.int64, device=dev),
                "stats": torch.zeros(4, dtype=torch.float32, device=dev),
                "report": torch.zeros(4, dtype=torch.float32, device=dev),
                "partials": torch.empty(TRAIN_BLOCKS * _native.lib().pulse_qnet_slice_floats(), dtype=torch.float32,
                                        device=dev),                                            # 37 MB at 256 workgroups
            }
        words = 517 * ((max(int(n_rows), 1 << 16) + 255) // 256) + 512        # the training launch's row lists + the act launch's (pulse_env.h)
        if nat.get("select") is None or nat["select"].numel() < words:
            nat["select"] = torch.empty(words, dtype=torch.int32, device=self._flat.device)
            self._struct_cache.pop("train", None)
        t = self._struct_cache.get("train")
        if t is not None and t.params == self._flat.data_ptr():
            t.lr, t.weight_decay, t.gamma, t.update_freq = self.lr, self.wd, float(self.gamma), int(self.update_freq)
            t.dropout_p = float(self.network[4].p) if self.network.training else 0.0
            t.separate_apply, t.meet_wait_ticks = int(self.separate_apply), int(self.meet_wait_ticks)
            return t
        t = self._struct_cache["train"] = _native.QNetTrain()
        t.net, t.target = self._net_struct(self.network), self._net_struct(self.target_network)
        t.params, t.target_params = self._flat.data_ptr(), self._flat_target.data_ptr()
        t.grad, t.exp_avg, t.exp_avg_sq = nat["grad"].data_ptr(), nat["m"].data_ptr(), nat["v"].data_ptr()
        t.step, t.stats, t.report = nat["step"].data_ptr(), nat["stats"].data_ptr(), nat["report"].data_ptr()
        t.partials, t.max_blocks = nat["partials"].data_ptr(), TRAIN_BLOCKS
        t.select_scratch, t.select_words = nat["select"].data_ptr(), nat["select"].numel()
        t.lr, t.weight_decay, t.beta1, t.beta2, t.eps = self.lr, self.wd, 0.9, 0.999, 1e-8       # torch.optim.AdamW defaults (:296)
        t.max_grad_norm, t.gamma = 1.0, float(self.gamma)                                      # clip_grad_norm_ (:280)
        t.dropout_p = float(self.network[4].p) if self.network.training else 0.0
        t.update_freq = int(self.update_freq)
        t.separate_apply, t.meet_wait_ticks, t.debug_meet_extra = int(self.separate_apply), int(self.meet_wait_ticks), 0
        return t

    def train_step_native(self, states, actions, rewards, next_states, dones, row_mask=None, step_counter=None, terminated=None,
                          reward_sum=None):
        """train_step (Player.py:255-294) as two or three launches on the env's stream and no host sync: row filter
        (row_mask & seat status ACTIVE/ALLIN) + TD target + forward + backward on the matrix cores -> reduction of the
        workgroups' gradient slices -> gradient mean, clip_grad_norm_, AdamW, target sync every update_freq optimizer
        steps.  `terminated` (bool[n], |= dones) and `reward_sum` (float64 scalar, += rewards over row_mask) fold the
        trainer's per-step bookkeeping (scripts/Poker/trainGPU.py:86,96) into the same launches.  Returns the device tensor
        [rows trained on, MSE loss, gradient norm before clipping, 0] of this call (read it later, or never).
        Moments live in this path's own buffers (not in self.optimizer, which serves the torch train_step)."""
        states, next_states = self._rows(states), self._rows(next_states)
        n = states.shape[0]
        t = self._native_state(n)
        if n == 0:                            # nothing to learn from: the reference returns before the optimizer (:262)
            return self._native["report"]

        def u8(x):
            if x is None:
                return None
            if x.dtype == torch.bool:
                x = x.view(torch.uint8)
            if x.dtype != torch.uint8 or not x.is_contiguous():
                x = x.to(torch.uint8).contiguous()
            return x
        dones8, mask8 = u8(dones), u8(row_mask)
        term8 = None
        if terminated is not None:           # trainer bookkeeping folded into the launch: terminated |= dones, reward_sum += ...
            if terminated.dtype not in (torch.bool, torch.uint8) or not terminated.is_contiguous():
                raise ValueError("terminated must be a contiguous bool / uint8 tensor (it is updated in place)")
            term8 = terminated.data_ptr()
        if reward_sum is not None and (reward_sum.dtype != torch.float64 or reward_sum.numel() != 1):
            raise ValueError("reward_sum must be a float64 scalar tensor (it is accumulated in place)")
        if actions.dtype != torch.int64 or not actions.is_contiguous():
            actions = actions.to(torch.int64).contiguous()
        if rewards.dtype != torch.float32 or not rewards.is_contiguous():
            rewards = rewards.to(torch.float32).contiguous()
        self._calls += 1
        step = (1 << 41) + self._calls if step_counter is None else int(step_counter)
        lib, stream = _native.lib(), _native.current_stream(states.device)
        # the row lists act_into(select_for_training=True) wrote serve exactly the call on its states and its mask
        made = getattr(self, "_act_selected", None)
        t.select_from_act = int(made is not None and mask8 is not None
                                and made == (states.data_ptr(), states.stride(0), mask8.data_ptr(), n))
        self._act_selected = None
        args = (C.byref(t), states.data_ptr(), states.stride(0), actions.data_ptr(), rewards.data_ptr(), next_states.data_ptr(),
                next_states.stride(0), dones8.data_ptr(), None if mask8 is None else mask8.data_ptr(), n, self.seed & (2**64 - 1), step,
                self.table_id0, term8, None if reward_sum is None else reward_sum.data_ptr(), stream)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and self.data_parallel:
            # data-parallel learner: every rank trains on its own tables, the 130 KB gradient sum and the
            # row count are all-reduced over xGMI, every rank applies the identical AdamW step
            _native.check(lib.pulse_qnet_train_grads(*args), "pulse_qnet_train_grads")
            nat = self._native
            grad, stats = nat["grad"], nat["stats"]
            if dist.get_backend() == "gloo":          # one-GPU rehearsal / CPU-side reduction
                g, s2 = grad.cpu(), stats[1:3].cpu()
                dist.all_reduce(g); dist.all_reduce(s2)
                grad.copy_(g); stats[1:3].copy_(s2)
            else:
                dist.all_reduce(grad); dist.all_reduce(stats[1:3])
            stats[0:1].copy_(grad.square().sum().reshape(1))
            nat["step"] += (stats[1:2] > 0).to(torch.int64)
            _native.check(lib.pulse_qnet_train_apply(C.byref(t), stream), "pulse_qnet_train_apply")
        else:
            _native.check(lib.pulse_qnet_train_step(*args), "pulse_qnet_train_step")
        self.step_count += 1             # calls; the optimizer-step count (calls with at least one valid row) is native_steps()
        return self._native["report"]

    def native_steps(self) -> int:
        """Optimizer steps taken by train_step_native (device counter; reading it syncs)."""
        return 0 if self._native is None else int(self._native["step"].item())

    # ------------------------------------------------------------------ learning
    def _after_update(self, loss, q_taken, rewards):
        self.step_count += 1
        if self.step_count % 1000 == 0:                                            # Player.py:281-287
            print(f"Step {self.step_count} | Avg Loss: {float(loss):.2f} | Avg Q: {float(q_taken.mean()):.2f} | "
                  f"Avg Reward: {float(rewards.mean()):.2f} | Epsilon: {self.epsilon:.4f}")
        if self.step_count % self.update_freq == 0:                                # :289-290
            self.target_network.load_state_dict(self.network.state_dict())

    def train_step(self, states, actions, rewards, next_states, dones):
        """Player.py:255-294 -- the reference's signature.  On the GPU the update runs on the native kernels
        (`train_step_native` with every passed row a candidate; the filter on the seat status, :261, is the kernel's):
        no boolean indexing, no `.any()` sync, no autograd graph.  Returns the MSE loss as a 0-d DEVICE tensor
        (`float(loss)` works and only then waits for the GPU; 0 when no row was valid, as the reference's early return).
        Moments and step count of this path live in its own buffers (`train_step_native`), not in `self.optimizer`;
        dropout draws are Philox (a documented deviation).  `train_step_torch` is the same update on PyTorch autograd
        with the reference's own op sequence -- what the reference-fixture test compares (and the path of a CPU module)."""
        if self._flat is None or not (torch.is_tensor(states) and states.is_cuda):
            return self.train_step_torch(states, actions, rewards, next_states, dones)
        report = self.train_step_native(states, actions, rewards, next_states, dones, None)
        if self.step_count % 1000 == 0:                                            # Player.py:281-287 (the Q / reward means are not kept)
            print(f"Step {self.step_count} | Avg Loss: {float(report[1]):.2f} | Epsilon: {self.epsilon:.4f}")
        return report[1]

    def check_native_report(self, report_host=None, wait=True):
        """Raises if a native update was called off inside its launch (report[3] = -1, pulse_env.h: PulseQNetTrain).
        wait=True reads the last report (waits for the stream); wait=False only looks at the library's count of called-off
        meetings in pinned host memory (no wait: the trainer's per-episode check; it is behind by the launches in flight)."""
        if self._native is None:
            return
        if not wait and report_host is None:
            seen = int(_native.lib().pulse_qnet_called_off_meetings())
            if seen == self._meetings_called_off:
                return
            self._meetings_called_off = seen
            rep = [0.0, 0.0, 0.0, -1.0]
        else:
            rep = self._native["report"].cpu() if report_host is None else report_host
        if float(rep[3]) < 0.0:
            raise RuntimeError("PokerQNetwork: a reduce + AdamW launch could not gather its workgroups (is another process using this GPU?) "
                               "and applied no update; set `separate_apply = True` to run AdamW as a launch of its own")

    def train_step_torch(self, states, actions, rewards, next_states, dones):
        """Player.py:255-294 with the reference's filtering semantics and op sequence on PyTorch autograd (boolean indexing:
        syncs with the host)."""
        valid_mask = (states[:, 12] == 0) | (states[:, 12] == 2)                   # seat status ACTIVE or ALLIN (:261)
        if not valid_mask.any():
            return 0.0
        states, actions, rewards = states[valid_mask], actions[valid_mask], rewards[valid_mask]
        next_states, dones = next_states[valid_mask], dones[valid_mask]
        q_taken = self.forward(states).gather(1, actions.unsqueeze(1)).squeeze(1)
        with torch.no_grad():
            next_q = self.target_network(next_states).max(dim=1).values
            targets = rewards + self.gamma * next_q * (~dones).float()
        loss = self.criterion(q_taken, targets)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.parameters(), max_norm=1.0)
        if self._lr_gate is not None:
            self._lr_gate.fill_(self.lr)
        self.optimizer.step()
        self._after_update(loss, q_taken, rewards)
        return loss

    def train_step_masked(self, states, actions, rewards, next_states, dones, row_mask):
        """The same update with the row filter (`row_mask & status in {ACTIVE, ALLIN}`) applied as 0/1 weights:
        loss = sum w (q - target)^2 / max(sum w, 1).  No boolean indexing, no `.any()`: nothing syncs.  When no row
        is valid the gradient is zero and the learning rate of this step is gated to zero on the device, so the
        weights do not move (the reference returns before the optimizer; only AdamW's moment decay differs)."""
        w = (row_mask & ((states[:, 12] == 0) | (states[:, 12] == 2))).to(torch.float32)
        count = w.sum()
        q_taken = self.forward(states).gather(1, actions.clamp(0, self.action_dim - 1).unsqueeze(1)).squeeze(1)
        with torch.no_grad():
            next_q = self.q_values(next_states, target=True).max(dim=1).values
            targets = rewards + self.gamma * next_q * (~dones).float()
        loss = (w * (q_taken - targets) ** 2).sum() / count.clamp(min=1.0)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.parameters(), max_norm=1.0)
        if self._lr_gate is None:
            self._lr_gate = torch.zeros((), dtype=torch.float32, device=states.device)
            for g in self.optimizer.param_groups:
                g["lr"] = self._lr_gate                     # tensor lr: read on the device by the fused AdamW
        self._lr_gate.copy_((count > 0).to(torch.float32) * self.lr)
        self.optimizer.step()
        self._after_update_masked()
        return loss

    def _after_update_masked(self):
        self.step_count += 1
        if self.step_count % self.update_freq == 0:
            self.target_network.load_state_dict(self.network.state_dict())
