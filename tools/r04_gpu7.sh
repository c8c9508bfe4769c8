#!/bin/bash
mkdir -p gpurun_out
LAB_ONLY=0,5,13 timeout -k 10 200 ./tools/stft32_lab 30 | tail -4
echo "--- packed twiddle products"
LAB_ONLY=0,5,13 timeout -k 10 200 ./tools/stft32_lab_pkc 30 | tail -4
