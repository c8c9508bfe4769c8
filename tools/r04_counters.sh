#!/bin/bash
# Counter survey of stft_chroma32_kernel (product, 2 waves/SIMD, no barriers) through tools/stft32_lab: several PMC passes.
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r04_counters
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/avail.txt" 2>&1
grep -o "SQ[C]*_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*" "$OUT/avail.txt" | sort -u > "$OUT/avail_names.txt"
wc -l "$OUT/avail_names.txt"
export LAB_ONLY="0,2,17"   # product, 2 waves per SIMD, no workgroup barrier at all (indices into stft32_lab's list)
pass() { n=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/p$n" -- "$REPO/tools/stft32_lab" 4 > "$OUT/p$n.log" 2>&1; echo "pass $n rc=$?"; }
pass 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD
pass 2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_EXP_GDS
pass 3 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_IFETCH SQ_INST_CYCLES_VMEM_RD
pass 4 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN
pass 5 SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_FMA_F32
pass 6 SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_BUSY_CU_CYCLES SQ_ACCUM_PREV
python3 - "$OUT" <<'PY' > "$OUT/summary.txt"
import csv, glob, sys, collections, re
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"stft_chroma32_kernel<(\d+), (\d+), (\d+)>", k)
        if not m: continue
        rows[(m.group(2), m.group(3))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(rows):
    c = {n: sum(v) / len(v) for n, v in rows[key].items()}
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    print("== waves/SIMD %s LAB %s  (%d launches)" % (key[0], key[1], len(next(iter(rows[key].values())))))
    for n in sorted(c):
        print("   %-28s %.4g%s" % (n, c[n], ("  (%.1f%% of SQ_WAVE_CYCLES)" % (100 * c[n] / wc)) if wc else ""))
PY
cat "$OUT/summary.txt"
for n in 1 2 3 4 5 6; do grep -i "error\|invalid\|not supported\|unknown" "$OUT/p$n.log" | head -3; done
find "$OUT" -name "*.csv" -size +3M -delete; find "$OUT" -name "*.db" -delete
