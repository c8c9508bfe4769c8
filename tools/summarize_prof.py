#!/usr/bin/env python3
"""Condenses a tools/profile_gpu.sh output directory into a Markdown summary (+ traffic.json):
per-kernel launch count / average duration from the rocprofv3 kernel-trace, and per-launch PMC values
(FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes for wide coalesced reads)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNELS = {"stft_chroma32": "stft_chroma32_kernel", "features_cert": "features_classify_cert_kernel", "fixup_items": "fixup_items_kernel",
           "stft_chroma": "stft_chroma_kernel", "fir_norm": "fir_norm_kernel", "features_classify": "features_classify_kernel",
           "classify": "classify_kernel",
           "hamming_runs_mfma": "hamming_runs_mfma2_kernel", "hamming_runs_sampled": "hamming_runs_sampled_kernel", "hamming_runs_band": "hamming_runs_band_kernel", "hamming_runs": "hamming_runs_kernel",
           "simhash_runs": "simhash_runs_kernel", "epilogue_entries": "pair_entries_kernel", "epilogue_best_match": "best_match_kernel"}


def short(name):
    for k, v in KERNELS.items():
        if v in name:
            if k == "stft_chroma" and (", true" in name or "Lb1E" in name):
                return "stft_fallback"                            # the LISTED instantiation: f64 recomputation of listed chunks
            return k
    return None


def rows(pattern):
    for path in glob.glob(pattern, recursive=True):
        with open(path, newline="") as f:
            yield from csv.DictReader(f)


def main():
    out = sys.argv[1]
    dur = defaultdict(list)
    for r in rows(os.path.join(out, "stats", "**", "*kernel_trace.csv")):
        k = short(r.get("Kernel_Name", ""))
        if k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("## rocprofv3 --kernel-trace --stats (bench.py)\n")
    print("| kernel | launches | avg us | min us | max us |\n|---|---|---|---|---|")
    for k, v in dur.items():
        print(f"| {k} | {len(v)} | {sum(v)/len(v):.1f} | {min(v):.1f} | {max(v):.1f} |")
    pmc = defaultdict(lambda: defaultdict(list))
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        for r in rows(os.path.join(out, sub, "**", "*counter_collection.csv")):
            k = short(r.get("Kernel_Name", ""))
            if k:
                pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("\n## PMC per launch (separate passes)\n")
    traffic, issue = {}, {}
    for k, counters in pmc.items():
        line = [f"**{k}**"]
        for c, vals in sorted(counters.items()):
            line.append(f"{c}={sum(vals)/len(vals):.4g}")
        fetch = counters.get("FETCH_SIZE")
        write = counters.get("WRITE_SIZE")
        if fetch and write:
            # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE reads exactly half of a wide coalesced stream
            fb = 2.0 * 1024.0 * sum(fetch) / len(fetch)
            wb = 1024.0 * sum(write) / len(write)
            traffic[k] = int(fb + wb)
            line.append(f"HBM bytes/launch (2*FETCH+WRITE) = {fb + wb:.4g} (fetch {fb:.4g}, write {wb:.4g})")
        act, gui = counters.get("SQ_ACTIVE_INST_VALU"), counters.get("GRBM_GUI_ACTIVE")
        if act and gui:
            # share of the SIMDs' issue cycles a vector instruction occupies: a wave-instruction holds its SIMD's vector ALU for 4
            # cycles; SQ_ACTIVE_INST_VALU is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
            issue[k] = round(4.0 * (sum(act) / len(act)) / (1024.0 * (sum(gui) / len(gui)) / 8.0), 4)
            line.append(f"vector-instruction issue = {100 * issue[k]:.1f} % of the SIMDs' cycles")
        print("- " + ", ".join(line))
    if issue:
        traffic["valu_issue_frac"] = issue
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"))
    for name in ("stats.log", "plain.json"):
        p = os.path.join(out, name)
        if os.path.exists(p):
            for ln in open(p):
                if ln.startswith("{\"metric\"") or ln.startswith("{\"episodes\""):
                    print("\n## bench line under the profiler\n\n```\n" + ln.strip() + "\n```")


if __name__ == "__main__":
    main()
