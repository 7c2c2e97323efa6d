import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(sys.argv[1] if len(sys.argv) > 1 else "", f"{d['value']:.4g} steps/s", f"{d['ms_per_step']*1e3:.2f} us/step", f"kernel {r.get('kernel_us', 0):.2f} us", f"frac {r.get('frac', 0):.3f}")
