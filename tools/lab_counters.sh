#!/bin/bash
# Runs on the GPU box: SQ counters per variant of tools/stft_lab (every variant is a kernel of its own name), one pass.
# Usage: tools/lab_counters.sh <binary> <tag>   -> gpurun_out/labpmc_<tag>.txt
BIN=${1:-tools/stft_lab}; TAG=${2:-run}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/labpmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE \
  --output-format csv -d "$OUT" -- "$REPO/$BIN" 6 > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY' > "$REPO/gpurun_out/labpmc_$TAG.txt"
import csv, glob, sys, collections, re
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"stft_chroma_kernel<(\d+), (\d+)>|stft_chroma_kernelILi(\d+)ELi(\d+)E", k)
        name = "LAB %s" % (m.group(2) or m.group(4)) if m else k[:40]
        key = (name, r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")))
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(rows):
    c = {n: sum(v) / len(v) for n, v in rows[key].items()}
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    print("%-10s lds=%-7s launches=%3d  " % (key[0], key[1], len(next(iter(rows[key].values())))) +
          "  ".join("%s=%.3g (%.1f%%)" % (n.replace("SQ_", ""), c[n], 100 * c[n] / wc) if n != "SQ_INSTS_VALU" else "%s=%.4g" % (n.replace("SQ_", ""), c[n]) for n in sorted(c)))
PY
cat "$REPO/gpurun_out/labpmc_$TAG.txt"
find "$OUT" -name "*.csv" -size +3M -delete; find "$OUT" -name "*.db" -delete
