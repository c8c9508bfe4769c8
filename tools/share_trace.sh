#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r04_share_trace
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$REPO/bench.py" --steps 30 --warmup 5 --preheat 0 --no-extras --no-cpu-baseline > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        short = "stft32" if "stft_chroma32" in n else "cert" if "cert_kernel" in n else "fallback" if "stft_chroma_kernel" in n else "fixup" if "fixup" in n else "scan" if "hamming" in n else "simhash" if "simhash" in n else None
        if short: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id","?")))
rows.sort()
t0=rows[len(rows)//2][0]
for s,e,n,q in rows[len(rows)//2: len(rows)//2+26]:
    print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} us  {n:9s} queue {q}  dur {(e-s)/1e3:7.1f}")
PY
find "$OUT" -name "*.csv" -size +2M -delete; find "$OUT" -name "*.db" -delete
