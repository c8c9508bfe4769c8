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
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "mfma_scan" if "hamming_runs_mfma" in k else "valu_scan" if "hamming_runs_sampled" in k else None
        if name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    print(name, {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())}, "launches", len(next(iter(cs.values()))))
PY
find "$OUT" -name "*.csv" -size +2M -delete; find "$OUT" -name "*.db" -delete
