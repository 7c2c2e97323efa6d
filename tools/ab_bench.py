"""A/B of kernel builds on ONE box (box-to-box variation is +-2 %, more than most kernel changes): runs bench.py's env-only
leg with each given library in turn, twice, alternating, and prints kernel time / value per run.
    python tools/ab_bench.py pulselib_amd/libpulse_hip.so /path/to/other.so [--tables N] [--active-players sampled|10]
(a library is selected through the PULSE_LIB environment variable, which pulselib_amd/_native.py honours -- diagnostic)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
extra = [a for a in sys.argv[1:] if not a.endswith(".so")]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, PULSE_LIB=os.path.abspath(lib))
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--inproc", "--no-cpu-baseline", "--trainer-loop", "off", "--other-envs", "off",
                            "--census", "off", "--active-players", "sampled", "--min-timed-ms", "600", *extra], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            print(f"{lib}: FAILED rc={p.returncode}", flush=True)
            continue
        d = json.loads(lines[-1])
        r = d["roofline"]
        print(f"round {rnd} {os.path.basename(lib):40s} value {d['value']:.4g}  kernel {r['kernel_us']:.2f} us / {r['steps_per_launch']:.1f} steps  frac {r['frac']:.3f}", flush=True)
