#!/bin/bash
# Throughput of the two-pass engines against the batch size of one call: does a message store that fits the
# 256 MB Infinity Cache run faster than one that streams through HBM?   bash tools/probes/batch_sweep.sh > gpurun_out/batch_sweep.txt
set -e
run() {  # code alg frames...
  code=$1; alg=$2; shift 2
  for f in "$@"; do
    BENCH_CODES_FRAMES=$f timeout -k 10 120 python tools/bench_codes.py $code $alg 2>/dev/null | tail -n +2
  done
}
export LDPC_HIP_FUSED=0 LDPC_HIP_CHAIN=0
run peg_16000 MINSUM_Q88 1280 1536 1792 2048 2304 2560 3072 8192
run peg_16000 MINSUM_F32 512 768 1024 1280 1536 4096
run peg_16000 MINSUM_F64 256 384 512 640 768 2048
run 32000 MINSUM_F64 128 256 384 512 2048
run dvb MINSUM_F64 64 128 192 1088
run dvb MINSUM_Q88 256 384 512 640 768 1088 4352
run dvb MINSUM_F32 128 256 384 1088
run 4095 MINSUM_Q88 4096 8192 12288 65536
