#!/bin/bash
# Span size (LDPC_HIP_PASS_MB: megabytes of edge messages decoded to the end before the next span starts) against
# throughput of the two-pass engines.   bash tools/probes/span_sweep.sh > gpurun_out/span_sweep.txt
set -e
export LDPC_HIP_FUSED=0 LDPC_HIP_CHAIN=0
run() {  # code alg frames mb...
  code=$1; alg=$2; f=$3; shift 3
  for mb in "$@"; do
    echo -n "PASS_MB=$mb  "
    LDPC_HIP_PASS_MB=$mb BENCH_CODES_FRAMES=$f timeout -k 10 120 python tools/bench_codes.py $code $alg 2>/dev/null | tail -n +2
  done
}
run peg_16000 MINSUM_Q88 20480 0 120 160 200 240 280 320
run peg_16000 MINSUM_F64 5120 0 160 200 240 280 320
run 32000 MINSUM_F32 2560 0 160 200 240 280 320
run peg_16000 SPA_F32 5120 0 240
run dvb SPA_F32 1024 0 240
run dvb SPA_TABLE 1024 0 240
run peg_16000 SPA_TABLE 5120 0 240
