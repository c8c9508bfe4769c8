#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 ./tools/issue_rates > gpurun_out/r04_issue_rates2.log 2>&1; echo rc=$?
grep -a "v_xor\|v_bcnt\|v_cmp\|v_max3\|scan cell\|v_cndmask\|v_add_f32_e32" gpurun_out/r04_issue_rates2.log
