"""Benchmark harness of the Poker GPU path: the reference's `python -m benchmarking.Poker.run` (benchmarking/Poker/)
on the HIP classes -- same presets, case names, units, report fields and LLM_BENCHMARK_SUMMARY block, so old and new
reports diff line by line.  `python -m pulselib_amd.benchmarking --preset quick|standard|stress|mi355x`."""
from .poker import CASE_REGISTRY, DEFAULT_CASES, PRESETS, resolve_preset, run_benchmarks  # noqa: F401
