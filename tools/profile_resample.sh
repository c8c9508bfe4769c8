#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of tools/bench_resample.py (the resampler front-end:
# 48 kHz stereo / 44.1 kHz stereo / 48 kHz mono, 8 x 12-minute windows, three calls each), then separate PMC passes for
# the HBM traffic.  Usage: tools/profile_resample.sh <label>  ->  gpurun_out/prof_<label>/
LABEL=${1:-resample}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$LABEL
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/tools/bench_resample.py" > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/tools/bench_resample.py" > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/tools/bench_resample.py" > "$OUT/pmc_write.log" 2>&1
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
lines = ["# rocprofv3 summary: tools/bench_resample.py (resampler front-end)", ""]
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    lines += ["| kernel | calls | average us | min us | max us |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(f)):
        if "resample" in r["Name"]:
            lines.append("| `%s` | %s | %.1f | %.1f | %.1f |" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                               float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
for name, pat in (("FETCH_SIZE", "/pmc_fetch/**/*counter_collection.csv"), ("WRITE_SIZE", "/pmc_write/**/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + pat, recursive=True):
        for r in csv.DictReader(open(f)):
            if "resample" in r["Kernel_Name"] and r["Counter_Name"] == name:
                acc[r["Kernel_Name"][:100]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        # the guide's gfx950 rules: FETCH_SIZE counts 32-byte units... see tools/summarize_prof.py; raw values here
        lines.append("%s `%s`: %d launches, raw counter average %.0f" % (name, k, len(v), sum(v) / len(v)))
open(out + "/summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
cat "$OUT/stats.log" | grep "Hz x"
