# round 5, final tree: r5_final_a.sh (GPU suite, bench lines, kernel trace + PMC passes of the driver's form), then
# r5_final_b.sh (BASELINE.json configs[4] at full size: bench line, kernel trace; the scan's counters at 79 800 pairs)
bash tools/r5_final_a.sh && bash tools/r5_final_b.sh
