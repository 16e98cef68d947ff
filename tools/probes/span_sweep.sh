#!/bin/bash
# Span size of the two-pass min-sum engines (LDPC_HIP_PASS_SLABS: slabs of 64 x V frames decoded to the end before the
# next span starts; 0 = the whole call at once) against throughput.   bash tools/probes/span_sweep.sh > gpurun_out/span_sweep.txt
set -e
export LDPC_HIP_FUSED=0 LDPC_HIP_CHAIN=0
run() {  # code alg frames slabs...
  code=$1; alg=$2; f=$3; shift 3
  for k in "$@"; do
    echo -n "slabs=$k  "
    LDPC_HIP_PASS_SLABS=$k BENCH_CODES_FRAMES=$f timeout -k 10 120 python tools/bench_codes.py $code $alg 2>/dev/null | tail -n +2
  done
}
run peg_16000 MINSUM_Q88 20480 0 2 3 4 5 6 8 10
run peg_16000 MINSUM_F32 10240 0 2 3 4 5 6 8 10
run peg_16000 MINSUM_F64 5120 0 2 3 4 5 6 8 10
run 32000 MINSUM_F64 2560 0 2 3 4 5 6 8
run 4095 MINSUM_Q88 65536 0 8 12 16 20 24 32
run dvb MINSUM_Q88 4096 0 1 2 3 4
run dvb MINSUM_F32 2048 0 1 2 3 4
run dvb MINSUM_F64 2176 0 1 2 3 4 5 6 8
