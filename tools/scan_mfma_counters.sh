#!/bin/bash
# Runs on the GPU box (via gpurun): counters of the matrix-pipe form of the sampled scan (NEEDLE_HIP_SCAN_MFMA=1) on a
# library-scale job (tools/library_device.py, episodes x 45 min in HBM), separate rocprofv3 --pmc passes.
# Usage: tools/scan_mfma_counters.sh [episodes=400]   -> gpurun_out/scan_mfma_counters/
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
E=${1:-400}
OUT=$REPO/gpurun_out/scan_mfma_counters
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export NEEDLE_HIP_SCAN_MFMA=1
pass() { n=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/p$n" -- python3 "$REPO/tools/library_device.py" $E 2 1 > "$OUT/p$n.log" 2>&1; echo "pass $n rc=$?"; }
pass 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass 2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
python3 - "$OUT" "$E" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "mfma_scan" if "hamming_runs_mfma" in k else "valu_scan" if "hamming_runs_sampled" in k else None
        if name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
import json
summary = {}
for name, cs in acc.items():
    avg = {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())}
    print(name, avg, "launches", len(next(iter(cs.values()))))
    summary[name] = avg
m = summary.get("mfma_scan", {})
if m.get("SQ_VALU_MFMA_BUSY_CYCLES") and m.get("GRBM_GUI_ACTIVE"):
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES is cycles summed over the 1024 SIMDs
    summary["pipe_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0), 4)
    if m.get("SQ_ACTIVE_INST_VALU"):   # a vector wave-instruction holds its SIMD's vector ALU for 4 cycles
        summary["valu_issue_frac"] = round(4.0 * m["SQ_ACTIVE_INST_VALU"] / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0), 4)
    if m.get("SQ_INSTS_VALU") and m.get("SQ_INSTS_MFMA"):
        summary["vector_per_matrix_instruction"] = round(m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"], 2)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        summary["lds_bank_conflict_share"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"], 4)
    summary["source"] = "tools/scan_mfma_counters.sh %s: rocprofv3 --pmc, two passes, per launch" % (sys.argv[2] if len(sys.argv) > 2 else "")
json.dump(summary, open(sys.argv[1] + "/scan_mfma_counters.json", "w"), indent=1)
PY
find "$OUT" -name "*.csv" -size +2M -delete; find "$OUT" -name "*.db" -delete
