"""The benchmark harness keeps the reference's registry (benchmarking/Poker/cases.py:272-336, presets.py:17-48,
reporting.py:31-56): case names, categories, units, lower_is_better, presets -- restated here as data -- and, on the
GPU, the quick preset runs end to end and prints the LLM_BENCHMARK_SUMMARY block."""
import json

import pytest

# name -> (category, unit, lower_is_better): the reference's CASE_REGISTRY, field by field
REFERENCE_CASES = {
    "env_reset": ("environment", "seconds", True),
    "env_calculate_equities": ("environment", "seconds", True),
    "env_execute_actions": ("environment", "seconds", True),
    "env_step": ("environment", "seconds", True),
    "trainer_build_actions": ("trainer", "seconds", True),
    "trainer_q_network_train_step": ("trainer", "seconds", True),
    "trainer_short_run": ("end_to_end", "seconds", True),
}
REFERENCE_PRESETS = {"quick": (256, 2, 1, 3), "standard": (1024, 3, 2, 5), "stress": (4096, 5, 2, 7)}      # n_games, episodes, warm-up, measure
REFERENCE_DERIVED = {"env_reset": ("games_reset_per_second", "games_per_second"),
                     "env_calculate_equities": ("equity_batches_per_second", "games_per_second"),
                     "env_execute_actions": ("action_batches_per_second", "games_per_second"),
                     "env_step": ("env_steps_per_second", "games_per_second"),
                     "trainer_build_actions": ("actions_built_per_second", "games_per_second"),
                     "trainer_q_network_train_step": ("q_updates_per_second", "samples_per_second"),
                     "trainer_short_run": ("trainer_steps_per_second", "episode_games_per_second")}


def test_registry_and_presets_equal_the_reference():
    from pulselib_amd.benchmarking import CASE_REGISTRY, DEFAULT_CASES, PRESETS, resolve_preset
    assert DEFAULT_CASES == list(REFERENCE_CASES)
    for name, (category, unit, lower) in REFERENCE_CASES.items():
        c = CASE_REGISTRY[name]
        assert (c.name, c.category, c.primary_metric_name, c.primary_metric_unit, c.lower_is_better) == (name, category, "elapsed_seconds", unit, lower)
    for name, (n, ep, warm, meas) in REFERENCE_PRESETS.items():
        p = resolve_preset(name)
        assert (p["env"]["n_games"], p["env"]["episodes"], p["warmup_iterations"], p["measure_iterations"]) == (n, ep, warm, meas)
        assert p["cases"] == list(REFERENCE_CASES) and p["device"] == "auto"
    assert PRESETS["mi355x"]["env"]["n_games"] == 65536 and PRESETS["mi355x"]["cases"][:7] == list(REFERENCE_CASES)
    with pytest.raises(ValueError, match="Unknown preset"):
        resolve_preset("nope")
    p = resolve_preset("quick"); p["cases"].append("x")
    assert "x" not in PRESETS["quick"]["cases"]                                   # a copy, as deepcopy gives (presets.py:51-55)


@pytest.mark.gpu
def test_quick_preset_runs_every_case_and_prints_the_summary_block(tmp_path, capsys, monkeypatch):
    from pulselib_amd.benchmarking import run_benchmarks
    monkeypatch.chdir(tmp_path)
    report = run_benchmarks(preset_name="quick")
    out = capsys.readouterr().out
    lines = out.strip().splitlines()
    begin, end = lines.index("LLM_BENCHMARK_SUMMARY_BEGIN"), lines.index("LLM_BENCHMARK_SUMMARY_END")
    block = lines[begin + 1:end]
    assert block[0] == "benchmark_suite=poker_gpu_benchmarking" and block[1] == "preset=quick" and block[3] == "cases_run=7"
    case_lines = [ln for ln in block if ln.startswith("case=")]
    derived_lines = [ln for ln in block if ln.startswith("derived=")]
    assert [ln.split()[0][5:] for ln in case_lines] == list(REFERENCE_CASES)
    for ln, (name, (category, unit, lower)) in zip(case_lines, REFERENCE_CASES.items()):
        f = dict(kv.split("=", 1) for kv in ln.split())
        assert (f["case"], f["category"], f["unit"], f["lower_is_better"]) == (name, category, unit, str(lower).lower())
        assert float(f["value"]) > 0
    for ln, (name, (dname, dunit)) in zip(derived_lines, REFERENCE_DERIVED.items()):
        parts = ln.split()
        assert parts[0] == f"derived={name}" and parts[1].startswith(dname + "=") and parts[2] == f"unit={dunit}" and parts[3] == "higher_is_better=true"
    saved = json.loads(open(report["output_path"]).read())
    assert saved["metadata"]["benchmark_overrides"] == {"N_GAMES": 256, "EPISODES": 2}
    assert [c["name"] for c in saved["cases"]] == list(REFERENCE_CASES)
    for c in saved["cases"]:
        assert len(c["timings"]["trials"]) == 3 and set(c["timings"]) >= {"unit", "trials", "mean", "median", "min", "max", "stdev"}
    step = next(c for c in saved["cases"] if c["name"] == "env_step")
    assert step["primary_metric"]["value"] < 9.83e-3          # the reference's published env_step at 256 tables: 9.83 ms/call
