export TMPDIR=/tmp
mkdir -p gpurun_out/r5
( while sleep 50; do echo "[r5_gpu47] $(date +%T) still running"; done ) &
HB=$!
timeout -k 10 300 python bench.py > gpurun_out/r5/bench_final_default.json 2> gpurun_out/r5/bench_final.err; echo "bench default rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench_final_steps20_warmup5.json 2>> gpurun_out/r5/bench_final.err; echo "bench rc=$?"
NEEDLE_HIP_SCAN_MFMA=1 timeout -k 10 420 python tools/fuzz_search.py 600 251 2>&1 | tail -2 | tee gpurun_out/r5/fuzz_search_mfma2.log
timeout -k 10 420 python tools/fuzz_cert_adversarial.py 360 5 gpurun_out/r5/cert_adversarial_seed5.json 2>&1 | tee gpurun_out/r5/cert_adversarial_seed5.log | tail -10
kill $HB
