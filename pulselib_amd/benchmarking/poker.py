"""The seven cases of the reference's Poker benchmark harness on the MI355X classes.

Reference: benchmarking/Poker/presets.py:17-48 (presets quick / standard / stress: 256 / 1,024 / 4,096 tables, warm-up and
measure iteration counts), cases.py:272-336 (the case registry: name, category, description, unit, lower_is_better) and
:66-268 (what each case prepares and times), runtime.py:124-130 (a timed call = synchronize, clock, call, synchronize,
clock), runner.py:20-60 (report layout), reporting.py:31-56 (the LLM_BENCHMARK_SUMMARY block, printed here through
utils/benchmarking.emit_llm_summary).  One preset is ours: `mi355x` = BASELINE.json config 2's 65,536 tables, plus an
eighth case that times the fused trainer loop (train_agent_fused) next to the reference-semantics one.

Every case goes through the drop-in Python surface (PokerGPU / build_actions / PokerQNetwork / train_agent), i.e. it
measures the small-batch latency of one call -- interpreter + ctypes + one launch -- which bench.py's chunked roll-out
deliberately avoids."""
from __future__ import annotations

import argparse
import json
import statistics
import time
from copy import deepcopy
from dataclasses import dataclass
from datetime import datetime, timezone
from pathlib import Path
from typing import Callable

import torch

DEFAULT_CASES = ["env_reset", "env_calculate_equities", "env_execute_actions", "env_step", "trainer_build_actions",
                 "trainer_q_network_train_step", "trainer_short_run"]                      # presets.py:6-14


def _preset(n_games, episodes, warmup, measure, cases=None):
    return {"device": "auto", "cases": list(cases or DEFAULT_CASES), "warmup_iterations": warmup, "measure_iterations": measure,
            "env": {"n_games": n_games, "episodes": episodes}}


PRESETS = {
    "quick": _preset(256, 2, 1, 3),                                                         # presets.py:18-27
    "standard": _preset(1024, 3, 2, 5),                                                     # :28-37
    "stress": _preset(4096, 5, 2, 7),                                                       # :38-47
    "mi355x": _preset(65536, 3, 2, 7, DEFAULT_CASES + ["trainer_short_run_fused"]),         # BASELINE.json config 2's batch
}
RESET_OPTIONS = {"rotation": 0, "active_players": True, "q_agent_seat": 0}                  # what every case resets with (cases.py:71,...)
POKER_ACTION_SPACE_N = 13


def resolve_preset(name: str) -> dict:
    if name not in PRESETS:
        raise ValueError(f"Unknown preset '{name}'. Available presets: {', '.join(sorted(PRESETS))}")
    return deepcopy(PRESETS[name])


# ------------------------------------------------------------------------------------------------ context
@dataclass
class BenchmarkContext:
    config: dict
    benchmark_config: dict
    device: torch.device
    results_dir: Path
    weights_path: Path


def load_benchmark_context(preset: dict, results_root: Path, config_file="pokerGPU.yaml") -> BenchmarkContext:
    """runtime.py:50-79: the trainer's config with N_GAMES / EPISODES overridden by the preset."""
    from ..utils.config import get_config_file
    config = get_config_file(config_file)
    if config is None:
        raise FileNotFoundError(f"Could not load config/{config_file}")
    bench = dict(config)
    bench["N_GAMES"] = int(preset["env"]["n_games"])
    bench["EPISODES"] = int(preset["env"]["episodes"])
    bench["RESULTS_DIR"] = "PokerGPU"
    name = preset.get("device", "auto")
    device = torch.device("cuda", torch.cuda.current_device()) if name == "auto" and torch.cuda.is_available() else torch.device("cpu" if name == "auto" else name)
    if device.type != "cuda":
        raise RuntimeError("Poker GPU benchmarks require a CUDA device. This suite targets the live GPU poker environment and trainer only.")
    results_dir = Path(results_root) / "results" / "benchmarks" / "Poker"
    results_dir.mkdir(parents=True, exist_ok=True)
    return BenchmarkContext(config, bench, device, results_dir, results_dir / "benchmark_qnet_weights.pth")


def create_agents_and_types(ctx: BenchmarkContext):
    """runtime.py:82-104"""
    from ..environments.Poker import PokerAgentType, PokerQNetwork, load_gpu_agents
    cfg = ctx.benchmark_config
    agents, types = load_gpu_agents(ctx.device, cfg["NUM_PLAYERS"], cfg["AGENTS"], cfg["STARTING_BBS"], POKER_ACTION_SPACE_N)
    q_net = PokerQNetwork(weights_path=ctx.weights_path, device=ctx.device, gamma=cfg["GAMMA"], update_freq=cfg["UPDATE_FREQ"],
                          state_dim=cfg["STATE_SPACE"], action_dim=cfg["ACTION_SPACE"], learning_rate=cfg["LEARNING_RATE"],
                          weight_decay=cfg["WEIGHT_DECAY"], seed=int(cfg.get("SEED", 0)))
    agents.insert(0, q_net)
    types.insert(0, PokerAgentType.QLEARNING)
    return agents, types, q_net


def create_env(ctx: BenchmarkContext, agents):
    """runtime.py:107-121 (gym.make(ENV_ID, ...) there; the class directly here -- gymnasium is optional)"""
    from ..environments.Poker import PokerGPU
    cfg = ctx.benchmark_config
    return PokerGPU(device=ctx.device, agents=agents, n_players=cfg["NUM_PLAYERS"] + 1, n_games=cfg["N_GAMES"],
                    starting_bbs=cfg["STARTING_BBS"], w1=cfg["W1"], w2=cfg["W2"], K=cfg["K"], alpha=cfg["ALPHA"], seed=int(cfg.get("SEED", 0)))


def timed_call(fn, device):
    """runtime.py:124-130"""
    torch.cuda.synchronize(device)
    start = time.perf_counter()
    result = fn()
    torch.cuda.synchronize(device)
    return time.perf_counter() - start, result


# ------------------------------------------------------------------------------------------------ cases
@dataclass(frozen=True)
class BenchmarkCase:
    name: str
    category: str
    description: str
    primary_metric_name: str
    primary_metric_unit: str
    lower_is_better: bool
    runner: Callable


def _stats(values):
    return {"mean": statistics.fmean(values), "median": statistics.median(values), "min": min(values), "max": max(values),
            "stdev": statistics.stdev(values) if len(values) > 1 else 0.0}


def _result(case, timings, metadata, derived):
    """cases.py:32-52: the per-case block of the report"""
    summary = _stats(timings)
    return {"name": case.name, "category": case.category, "description": case.description,
            "primary_metric": {"name": case.primary_metric_name, "unit": case.primary_metric_unit, "value": summary["mean"],
                               "lower_is_better": case.lower_is_better},
            "timings": {"unit": case.primary_metric_unit, "trials": timings, **summary},
            "derived_metrics": derived, "metadata": metadata}


def _rate(name, count, seconds, unit="games_per_second"):
    return {"name": name, "value": count / seconds if seconds > 0 else 0.0, "unit": unit, "higher_is_better": True}


def _measure(ctx, call, warmup, measure, before_warm=None, before_timed=None):
    """warm-up calls, then `measure` timed calls; the optional hooks run untimed before each call"""
    for _ in range(warmup):
        arg = before_warm() if before_warm else None
        timed_call((lambda: call(arg)) if before_warm else call, ctx.device)
    timings = []
    for _ in range(measure):
        arg = before_timed() if before_timed else None
        timings.append(timed_call((lambda: call(arg)) if before_timed else call, ctx.device)[0])
    return timings


def run_env_reset(case, ctx, warmup, measure):                                            # cases.py:66-85
    env = create_env(ctx, create_agents_and_types(ctx)[0])
    timings = _measure(ctx, lambda: env.reset(options=dict(RESET_OPTIONS)), warmup, measure)
    env.close()
    n = ctx.benchmark_config["N_GAMES"]
    return _result(case, timings, {"n_games": n}, [_rate("games_reset_per_second", n, _stats(timings)["mean"])])


def run_env_calculate_equities(case, ctx, warmup, measure):                               # cases.py:88-120
    env = create_env(ctx, create_agents_and_types(ctx)[0])
    env.reset(options=dict(RESET_OPTIONS))
    pe = env.unwrapped if hasattr(env, "unwrapped") else env

    def river_then_equities():          # the reference times the preparation of the river state together with the call (:101-113)
        env.reset(options=dict(RESET_OPTIONS))
        pe.stages.fill_(3)
        pe.board[:, 0:3] = pe.deal_cards(pe.g, 3)
        pe.board[:, 3] = pe.deal_cards(pe.g, 1).squeeze(1)
        pe.board[:, 4] = pe.deal_cards(pe.g, 1).squeeze(1)
        pe.equities.fill_(0.5)
        pe.calculate_equities()

    timings = _measure(ctx, river_then_equities, warmup, measure)
    env.close()
    n = ctx.benchmark_config["N_GAMES"]
    return _result(case, timings, {"n_games": n, "street": "river"}, [_rate("equity_batches_per_second", n, _stats(timings)["mean"])])


def run_env_execute_actions(case, ctx, warmup, measure):                                  # cases.py:123-148
    env = create_env(ctx, create_agents_and_types(ctx)[0])
    env.reset(options=dict(RESET_OPTIONS))
    pe = env.unwrapped if hasattr(env, "unwrapped") else env
    n = ctx.benchmark_config["N_GAMES"]
    check_call = torch.ones(n, dtype=torch.long, device=ctx.device)
    timings = _measure(ctx, lambda _=None: pe.execute_actions(check_call), warmup, measure,
                       before_timed=lambda: env.reset(options=dict(RESET_OPTIONS)))
    env.close()
    return _result(case, timings, {"n_games": n, "action_profile": "check_call"}, [_rate("action_batches_per_second", n, _stats(timings)["mean"])])


def _default_actions(state, info, agents, types, device):                                 # runtime.py:133-136
    from ..environments.Poker import build_actions
    actions = torch.zeros(state.shape[0], dtype=torch.long, device=device)
    build_actions(state, actions, info["seat_idx"], agents, types, device)
    return actions


def run_env_step(case, ctx, warmup, measure):                                             # cases.py:151-173
    agents, types, _ = create_agents_and_types(ctx)
    env = create_env(ctx, agents)

    def prepare():
        state, info = env.reset(options=dict(RESET_OPTIONS))
        return _default_actions(state, info, agents, types, ctx.device)

    timings = _measure(ctx, lambda actions: env.step(actions), warmup, measure, before_warm=prepare, before_timed=prepare)
    env.close()
    n = ctx.benchmark_config["N_GAMES"]
    return _result(case, timings, {"n_games": n}, [_rate("env_steps_per_second", n, _stats(timings)["mean"])])


def run_trainer_build_actions(case, ctx, warmup, measure):                                # cases.py:176-196
    agents, types, _ = create_agents_and_types(ctx)
    env = create_env(ctx, agents)
    state, info = env.reset(options=dict(RESET_OPTIONS))
    timings = _measure(ctx, lambda: _default_actions(state, info, agents, types, ctx.device), warmup, measure)
    env.close()
    n = ctx.benchmark_config["N_GAMES"]
    return _result(case, timings, {"n_games": n}, [_rate("actions_built_per_second", n, _stats(timings)["mean"])])


def run_trainer_q_network_train_step(case, ctx, warmup, measure):                         # cases.py:199-225
    q_net = create_agents_and_types(ctx)[2]
    cfg = ctx.benchmark_config
    n, sd, ad = cfg["N_GAMES"], cfg["STATE_SPACE"], cfg["ACTION_SPACE"]
    states = torch.randn((n, sd), dtype=torch.float32, device=ctx.device)
    next_states = torch.randn((n, sd), dtype=torch.float32, device=ctx.device)
    actions = torch.randint(0, ad, (n,), dtype=torch.long, device=ctx.device)
    rewards = torch.randn((n,), dtype=torch.float32, device=ctx.device)
    dones = torch.zeros((n,), dtype=torch.bool, device=ctx.device)
    states[:, 12] = 0

    def update():
        loss = q_net.train_step(states, actions, rewards, next_states, dones)
        return float(loss.detach().item()) if isinstance(loss, torch.Tensor) else float(loss)

    timings = _measure(ctx, update, warmup, measure)
    return _result(case, timings, {"batch_size": n}, [_rate("q_updates_per_second", n, _stats(timings)["mean"], "samples_per_second")])


def _short_run(case, ctx, measure, fused):
    from ..scripts import trainGPU as train_gpu
    from ..utils.benchmarking import NullBenchmarker
    cfg = ctx.benchmark_config
    episodes, n = cfg["EPISODES"], cfg["N_GAMES"]
    timings, rates = [], []
    for _ in range(measure):
        agents, types, _ = create_agents_and_types(ctx)
        env = create_env(ctx, agents)
        kw = dict(env=env, agents=agents, agent_types=types, episodes=episodes, n_games=n, device=ctx.device,
                  results_dir=ctx.results_dir, config=cfg, plotter=None, benchmarker=NullBenchmarker(),
                  max_episode_steps=cfg.get("MAX_EPISODE_STEPS"))
        elapsed, _ = timed_call(lambda: (train_gpu.train_agent_fused if fused else train_gpu.train_agent)(**kw), ctx.device)
        timings.append(elapsed)
        rates.append(episodes * n / elapsed if elapsed > 0 else 0.0)          # cases.py:250 (episodes x games, as the reference defines it)
        env.close()
    derived = [{"name": "trainer_steps_per_second", "value": statistics.fmean(rates), "unit": "episode_games_per_second", "higher_is_better": True}]
    return _result(case, timings, {"episodes": episodes, "n_games": n}, derived)


def run_trainer_short_run(case, ctx, warmup, measure):                                    # cases.py:228-268
    return _short_run(case, ctx, measure, fused=False)


def run_trainer_short_run_fused(case, ctx, warmup, measure):
    return _short_run(case, ctx, measure, fused=True)


def _case(name, category, description, runner):
    return BenchmarkCase(name, category, description, "elapsed_seconds", "seconds", True, runner)


CASE_REGISTRY = {c.name: c for c in (                                                       # cases.py:272-336
    _case("env_reset", "environment", "Times live PokerGPU.reset() throughput for vectorized game batches.", run_env_reset),
    _case("env_calculate_equities", "environment", "Times live PokerGPU.calculate_equities() on prepared river-state batches.",
          run_env_calculate_equities),
    _case("env_execute_actions", "environment", "Times live PokerGPU.execute_actions() across the current vectorized game batch.",
          run_env_execute_actions),
    _case("env_step", "environment", "Times one live PokerGPU.step(...) call including reward and round progression work.", run_env_step),
    _case("trainer_build_actions", "trainer", "Times live trainer action routing through build_actions(...) for one batch.",
          run_trainer_build_actions),
    _case("trainer_q_network_train_step", "trainer", "Times live PokerQNetwork.train_step(...) update cost for one batch.",
          run_trainer_q_network_train_step),
    _case("trainer_short_run", "end_to_end", "Times a short live trainGPU.train_agent(...) run with no-op plotting and benchmark file output.",
          run_trainer_short_run),
    _case("trainer_short_run_fused", "end_to_end",
          "The same short run on train_agent_fused (learner and environment kernels back to back, no host sync per step); mi355x preset only.",
          run_trainer_short_run_fused),
)}


# ------------------------------------------------------------------------------------------------ runner / report
def run_benchmarks(*, preset_name="standard", selected_cases=None, output_dir=None, device_override=None, config_file="pokerGPU.yaml"):
    """runner.py:12-60"""
    from ..utils.benchmarking import emit_llm_summary
    preset = resolve_preset(preset_name)
    if device_override is not None:
        preset["device"] = device_override
    root = Path.cwd()
    ctx = load_benchmark_context(preset, root, config_file)
    names = selected_cases or preset["cases"]
    out_root = Path(output_dir) if output_dir else root / "results" / "benchmarks" / "Poker"
    out_root.mkdir(parents=True, exist_ok=True)
    results = []
    for name in names:
        if name not in CASE_REGISTRY:
            raise ValueError(f"Unknown case '{name}'. Available cases: {', '.join(sorted(CASE_REGISTRY))}")
        case = CASE_REGISTRY[name]
        results.append(case.runner(case, ctx, preset["warmup_iterations"], preset["measure_iterations"]))
    report = {
        "metadata": {"suite_name": "poker_gpu_benchmarking", "preset": preset_name, "device": str(ctx.device),
                     "generated_at_utc": datetime.now(timezone.utc).isoformat(), "warmup_iterations": preset["warmup_iterations"],
                     "measure_iterations": preset["measure_iterations"], "config_source": f"config/{config_file}",
                     "benchmark_overrides": {"N_GAMES": ctx.benchmark_config["N_GAMES"], "EPISODES": ctx.benchmark_config["EPISODES"]}},
        "cases": results,
    }
    path = out_root / f"poker_gpu_benchmark_{preset_name}_{datetime.now().strftime('%Y%m%d_%H%M%S')}.json"      # reporting.py:19-21
    path.write_text(json.dumps(report, indent=2, default=str), encoding="utf-8")
    report["output_path"] = str(path)
    emit_llm_summary(report)
    return report


def main(argv=None) -> int:
    """run.py:10-57"""
    ap = argparse.ArgumentParser(description="Run Poker GPU benchmarks against the live pulselib_amd codepaths.")
    ap.add_argument("--preset", default="standard", help="Benchmark preset to run (quick, standard, stress, mi355x). Default: standard")
    ap.add_argument("--case", dest="cases", action="append", help="Run only the named benchmark case. Repeat to select multiple cases.")
    ap.add_argument("--output-dir", type=Path, default=None, help="Optional output directory for the JSON report. Default: results/benchmarks/Poker")
    ap.add_argument("--device", default=None, help="Optional device override such as cuda or cuda:0.")
    ap.add_argument("--config", default="pokerGPU.yaml", help="configuration file (name under pulselib_amd/config/ or a path)")
    ap.add_argument("--list-cases", action="store_true", help="List available benchmark cases and exit.")
    args = ap.parse_args(argv)
    if args.list_cases:
        for name, case in CASE_REGISTRY.items():
            print(f"{name}: {case.description}")
        return 0
    run_benchmarks(preset_name=args.preset, selected_cases=args.cases, output_dir=args.output_dir, device_override=args.device,
                   config_file=args.config)
    return 0
