"""Run-summary and metric helpers with the reference's names (utils/benchmarking/*, utils/performance.py)."""
