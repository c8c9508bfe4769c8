export TMPDIR=/tmp
mkdir -p gpurun_out/r5
for i in 1 2; do
timeout -k 10 400 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); so=d['search_only']
print('bench', d['ms_per_step'], 'search_only', so['wall_ms'], so['wall_ms_each'], so['scan_kernel_ms'])"
done
timeout -k 10 300 python tools/search_only_trace.py 2>&1 | tail -1
