"""Run summaries an untouched Pulselib checkout can diff against its own.

`YamlBenchmarker.create_benchmark_file` writes `<results>/<folder>/runs/run_N.yaml` with the keys of the reference's
writer (utils/benchmarking/benchmarking.py:84-100: env, config, start_time, end_time, total_training_seconds,
total_steps, sps rounded to 4 places, episode_stats{count, mean, std, min, max, median}); N continues the numbering
of the files already there (utils/benchmarking/files.py:4-12).  `emit_llm_summary` prints the
LLM_BENCHMARK_SUMMARY block of benchmarking/Poker/reporting.py:31-56 for a report dict of the same shape."""
from __future__ import annotations

from pathlib import Path
from typing import Callable, Mapping

import numpy as np
import yaml

RESULT_FOLDERS = {"Pulse-Poker-GPU-v1": "PokerGPU", "Pulse-2048-v2": "2048"}      # utils/config.py:43-46


def default_results_root() -> Path:
    return Path.cwd() / "results"


def result_folder_for(env_name: str, root: Path | None = None) -> Path:
    if env_name not in RESULT_FOLDERS:
        raise ValueError(f"cannot get result folder for {env_name}")
    folder = (root or default_results_root()) / RESULT_FOLDERS[env_name]
    folder.mkdir(parents=True, exist_ok=True)
    return folder


def next_run_path(results_dir: Path) -> Path:
    runs = Path(results_dir) / "runs"
    runs.mkdir(parents=True, exist_ok=True)
    return runs / f"run_{sum(1 for f in runs.iterdir() if f.is_file()) + 1}.yaml"


def episode_statistics(episodes_return) -> dict:
    """utils/benchmarking/episodes.py:6-22 (population std, numpy median)."""
    values = np.asarray(episodes_return.detach().cpu().numpy() if hasattr(episodes_return, "detach") else episodes_return,
                        dtype=np.float64)
    return {"count": int(values.size), "mean": float(values.mean()), "std": float(values.std()), "min": float(values.min()),
            "max": float(values.max()), "median": float(np.median(values))}


def run_summary(env_name, episodes_return, start_time, end_time, total_steps, config) -> dict:
    """The dictionary the reference dumps (utils/benchmarking/benchmarking.py:84-100), key for key."""
    seconds = end_time - start_time
    sps = round(float(total_steps / seconds), 4) if seconds > 0 else 0.0
    return dict(env=env_name, config=config, start_time=start_time, end_time=end_time, total_training_seconds=seconds,
                total_steps=total_steps, sps=sps, episode_stats=episode_statistics(episodes_return))


class Benchmarker:
    """Base of the run-summary writers; same constructor, switches and hook as utils/benchmarking/benchmarking.py:19-57."""
    FEATURES = {"training_summary": True}

    def __init__(self, enabled: bool = True, feature_mask: Mapping[str, bool] | None = None,
                 results_dir_resolver: Callable[[str], Path] | None = None):
        self.enabled = bool(enabled)
        self.feature_mask = dict(self.FEATURES)
        self.feature_mask.update(feature_mask or {})
        self.results_dir_resolver = result_folder_for if results_dir_resolver is None else results_dir_resolver

    @classmethod
    def from_config(cls, config: Mapping[str, object] | None = None) -> "Benchmarker":
        cfg = dict(config or {})
        return cls(enabled=bool(cfg.get("enabled", True)), feature_mask=cfg.get("mask"))

    def is_enabled(self, feature_name: str) -> bool:
        return self.enabled and bool(self.feature_mask.get(feature_name, True))

    def create_benchmark_file(self, env_name, episodes_return, start_time, end_time, total_steps, config):
        raise NotImplementedError


class NullBenchmarker(Benchmarker):
    """Writes nothing."""

    def create_benchmark_file(self, *args, **kwargs):
        return None


class YamlBenchmarker(Benchmarker):
    """run_N.yaml under <results folder>/runs, N continuing the files already there."""

    def create_benchmark_file(self, env_name, episodes_return, start_time, end_time, total_steps, config):
        if not self.is_enabled("training_summary"):
            return None
        target = next_run_path(self.results_dir_resolver(env_name))
        print(target)
        target.write_text(yaml.dump(run_summary(env_name, episodes_return, start_time, end_time, total_steps, config),
                                    default_flow_style=False))
        return target


def emit_llm_summary(report: dict) -> None:
    """benchmarking/Poker/reporting.py:31-56, line for line the same fields."""
    meta = report["metadata"]
    print("LLM_BENCHMARK_SUMMARY_BEGIN")
    print(f"benchmark_suite={meta['suite_name']}")
    print(f"preset={meta['preset']}")
    print(f"device={meta['device']}")
    print(f"cases_run={len(report['cases'])}")
    print(f"output_path={report['output_path']}")
    for case in report["cases"]:
        pm = case["primary_metric"]
        print(f"case={case['name']} category={case['category']} unit={pm['unit']} value={pm['value']:.6f} "
              f"lower_is_better={str(pm['lower_is_better']).lower()}")
        for d in case.get("derived_metrics", []):
            print(f"derived={case['name']} {d['name']}={d['value']:.6f} unit={d['unit']} "
                  f"higher_is_better={str(d['higher_is_better']).lower()}")
    print("LLM_BENCHMARK_SUMMARY_END")
