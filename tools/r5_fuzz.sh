# round 5, end of round: the fuzzers on the final tree (GPU against the oracle), then ten more minutes of the adversarial search
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
( while sleep 50; do echo "[r5_fuzz] $(date +%T) still running"; done ) &
HB=$!
NEEDLE_HIP_SCAN_MFMA=1 timeout -k 10 500 python tools/fuzz_search.py 400 151 2>&1 | tail -3 | tee gpurun_out/r5/fuzz_search_mfma.log
timeout -k 10 300 python tools/fuzz_search.py 200 152 2>&1 | tail -2 | tee gpurun_out/r5/fuzz_search.log
NEEDLE_HIP_SCAN_MFMA=1 timeout -k 10 400 python tools/fuzz_pipeline.py 200 153 2>&1 | tail -2 | tee gpurun_out/r5/fuzz_pipeline_mfma.log
timeout -k 10 400 python tools/fuzz_fingerprint.py 1000 154 2>&1 | tail -2 | tee gpurun_out/r5/fuzz_fingerprint.log
NEEDLE_HIP_DEVICE_EPILOGUE=1 timeout -k 10 300 python tools/fuzz_epilogue.py 400 155 2>&1 | tail -2 | tee gpurun_out/r5/fuzz_epilogue.log
timeout -k 10 700 python tools/fuzz_cert_adversarial.py 600 4 gpurun_out/r5/cert_adversarial_seed4.json 2>&1 | tee gpurun_out/r5/cert_adversarial_seed4.log | tail -12
kill $HB
