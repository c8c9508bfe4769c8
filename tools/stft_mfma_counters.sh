#!/bin/bash
# Runs on the GPU box: SQ counters per variant of tools/stft_mfma_lab (every variant is a kernel of its own name), four passes.
# Usage: tools/stft_mfma_counters.sh [tag]   -> gpurun_out/mfma_lab_pmc_<tag>.txt
TAG=${1:-run}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/mfma_lab_pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() { n=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/p$n" -- "$REPO/tools/stft_mfma_lab" time > "$OUT/p$n.log" 2>&1; echo "pass $n rc=$?"; }
pass 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass 2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass 3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE
python3 - "$OUT" <<'PY' > "$REPO/gpurun_out/mfma_lab_pmc_$TAG.txt"
import csv, glob, sys, collections, re
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"stft_mfma_core_kernel<(\d+), (\d+)>", r["Kernel_Name"])
        if m:
            rows[(int(m.group(1)), int(m.group(2)))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(rows):
    c = {n: sum(v) / len(v) for n, v in rows[key].items()}
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    print("== workgroups per CU %d, LAB %d  (%d launches)" % (key[0], key[1], len(next(iter(rows[key].values())))))
    for n in sorted(c):
        print("   %-28s %.4g%s" % (n, c[n], ("  (%.1f%% of SQ_WAVE_CYCLES)" % (100 * c[n] / wc)) if wc and n.startswith("SQ_") else ""))
PY
cat "$REPO/gpurun_out/mfma_lab_pmc_$TAG.txt"
for n in 1 2 3; do grep -i "error\|invalid\|not supported\|unknown" "$OUT/p$n.log" | head -3; done
find "$OUT" -name "*.csv" -size +3M -delete; find "$OUT" -name "*.db" -delete
