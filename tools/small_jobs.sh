#!/bin/bash
# What one rank of an N-rank run of BASELINE.json configs[3] has to do, measured on ONE GPU: the 28-episode job's
# per-rank share is 28/N episodes of fingerprinting, so bench.py with --episodes 28/N (fewer pairs, same kernels)
# shows how close a short job stays to the sum of its kernels (launch gaps, host enqueue, download, epilogue).
cd "$(dirname "$0")/.."
for e in 28 14 7 4; do
  python bench.py --episodes $e --steps 100 --warmup 10 --no-extras --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['kernel_ms_per_step']
print('episodes %2d pairs %3d: %.4f ms/job, kernels %.4f (stft %.4f feat %.4f scan %.4f simhash %.4f), host enqueue %.4f wait+epilogue %.4f' % (d['config']['episodes'], d['config']['pairs'], d['ms_per_step'], sum(k.values()), k['stft_chroma'], k['features_classify'], k['hamming_runs'], k['simhash_runs'], d['host_ms_per_step']['enqueue'], d['host_ms_per_step']['wait_and_epilogue']))"
done
