#!/usr/bin/env bash
# Runs ON the GPU box: the streamed attention backward of two builds, interleaved (ABAB), on one box.   usage: attn4_ab.sh <variant suffix> [rounds]
v=$1; n=${2:-3}
for i in $(seq $n); do
  timeout -k 10 120 python tools/attn4_time.py 5 2>&1 | grep key7 | sed 's/^/new  /'
  LPI_LIB=lpi_amd/csrc/variants/liblpi_hip_$v.so timeout -k 10 120 python tools/attn4_time.py 5 2>&1 | grep key7 | sed 's/^/prev /'
done
