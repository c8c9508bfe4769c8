"""Reads bench.py's JSON line on stdin and prints the few numbers watched while tuning."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    k = d.get("kernel_ms_per_step", {})
    print(f"value={d['value']:.0f} {d['unit']}  ms/step={d['ms_per_step']:.4f}  "
          + "  ".join(f"{n}={v:.4f}" for n, v in k.items())
          + f"  frac={d.get('roofline', {}).get('frac')}  host={d.get('host_ms_per_step')}")
