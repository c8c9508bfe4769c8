#!/usr/bin/env python3
"""the sampled scan's head-row count H and window width W swept at LIBRARY scale on audio-like hashes
(tools/library_device.py: episodes x 45 min generated in HBM, full O(N^2) search), one process per shape
(NEEDLE_HIP_SCAN_SHAPE).  Per shape: scan kernel ms (HIP events), cell evaluations issued, diagonals that survived the
head rows, and the digest of the complete sorted run list -- which must be the same for every shape.

usage: python tools/scan_shape_sweep.py [episodes=1000] [minutes=45]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    n = sys.argv[1] if len(sys.argv) > 1 else "1000"
    minutes = sys.argv[2] if len(sys.argv) > 2 else "45"
    rows = []
    for w in (8, 4, 16):
        for h in (3, 2, 4):
            if h >= w:
                continue
            env = dict(os.environ, NEEDLE_HIP_SCAN_SHAPE=f"{w},{h}")
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "library_device.py"), n, "3", "2", minutes], env=env,
                                 capture_output=True, text=True, timeout=900)
            if out.returncode != 0:
                print(f"W={w} H={h}: failed\n{out.stderr[-600:]}", flush=True)
                continue
            d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
            rows.append({"W": w, "H": h, "scan_ms": d["kernel_ms"]["hamming_runs"], "job_ms": d["ms_per_job_two_in_flight"],
                         "issued_evaluations": d["scan_roofline"]["issued_cell_evaluations"], "head_survivors": d["head_survivors"],
                         "pruning_factor": d["scan_roofline"]["pruning_factor"], "frac_of_int_valu_roof": d["scan_roofline"]["frac"],
                         "runs": d["runs"], "run_list_digest": d["run_list_digest"]})
            print(json.dumps(rows[-1]), flush=True)
    same = len({r["run_list_digest"] for r in rows}) == 1
    base = next(r for r in rows if (r["W"], r["H"]) == (8, 3))
    best = min(rows, key=lambda r: r["scan_ms"])
    print(json.dumps({"episodes": int(n), "minutes": float(minutes), "same_run_list_for_every_shape": same,
                      "default_8_3_ms": base["scan_ms"], "best": best,
                      "gain_over_default": round(1.0 - best["scan_ms"] / base["scan_ms"], 4)}), flush=True)


if __name__ == "__main__":
    main()
