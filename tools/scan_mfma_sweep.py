#!/usr/bin/env python3
"""Round 5: the scan's matrix-pipe forms against each other at LIBRARY scale on audio-like hashes (tools/library_device.py:
episodes x minutes generated in HBM, full O(N^2) search), one process per variant.  A variant is a set of environment
switches (NEEDLE_HIP_MFMA_WAVES, NEEDLE_HIP_SCAN_MFMA=0 for the vector form).  Per variant: scan kernel
ms (HIP events), job ms, runs and the digest of the complete sorted run list -- which must be the same for all.

usage: python tools/scan_mfma_sweep.py [episodes=600] [minutes=45] [variants=all]
       variants: comma-separated names out of the table below"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = {
    "vector": {"NEEDLE_HIP_SCAN_MFMA": "0"},
    "w4": {"NEEDLE_HIP_MFMA_WAVES": "4"},
    "w12": {"NEEDLE_HIP_MFMA_WAVES": "12"},
    "w8": {"NEEDLE_HIP_MFMA_WAVES": "8"},
    "w16": {"NEEDLE_HIP_MFMA_WAVES": "16"},
    "w8forced": {"NEEDLE_HIP_MFMA_WAVES": "8", "NEEDLE_HIP_SCAN_MFMA": "1"},    # below the size the automatic choice takes it from
    "w8s2": {"NEEDLE_HIP_MFMA_WAVES": "8", "NEEDLE_HIP_SCAN_MFMA": "1", "NEEDLE_HIP_MFMA_SPLITS": "2"},   # workgroups per group
    "w8s3": {"NEEDLE_HIP_MFMA_WAVES": "8", "NEEDLE_HIP_SCAN_MFMA": "1", "NEEDLE_HIP_MFMA_SPLITS": "3"},
    "w8s4": {"NEEDLE_HIP_MFMA_WAVES": "8", "NEEDLE_HIP_SCAN_MFMA": "1", "NEEDLE_HIP_MFMA_SPLITS": "4"},
}
# laboratory builds (tools/build_variant.sh, wrong results, timing only): <lab>@<variant>, e.g. m2lab1@w12c2
LAB_DIR = os.path.join(ROOT, "needle_amd", "lib", "ab")


def main():
    n = sys.argv[1] if len(sys.argv) > 1 else "600"
    minutes = sys.argv[2] if len(sys.argv) > 2 else "45"
    names = sys.argv[3].split(",") if len(sys.argv) > 3 and sys.argv[3] != "all" else list(VARIANTS)
    rows = []
    for name in names:
        lab, _, base = name.rpartition("@")
        env = dict(os.environ, **VARIANTS[base])
        if lab:
            env["NEEDLE_CAPI_LIB"] = os.path.join(LAB_DIR, lab + ".so")
        env["NEEDLE_LIBRARY_DEVICE_NO_COUNT"] = "1"       # no counting launch of the vector form behind the timed jobs
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "library_device.py"), n, "3", "2", minutes], env=env,
                             capture_output=True, text=True, timeout=900)
        if out.returncode != 0:
            print(f"{name}: failed\n{out.stderr[-1500:]}", flush=True)
            continue
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        rows.append({"variant": name, "scan_ms": d["kernel_ms"]["hamming_runs"], "simhash_ms": d["kernel_ms"]["simhash_runs"],
                     "job_ms": d["ms_per_job_two_in_flight"], "scan_form": d.get("scan_form"), "runs": d["runs"],
                     "run_list_digest": d["run_list_digest"]})
        print(json.dumps(rows[-1]), flush=True)
    same = len({r["run_list_digest"] for r in rows if "@" not in r["variant"]}) == 1
    print(json.dumps({"episodes": int(n), "minutes": float(minutes), "same_run_list_for_every_variant": same,
                      "best": min(rows, key=lambda r: r["scan_ms"]) if rows else None}), flush=True)
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
