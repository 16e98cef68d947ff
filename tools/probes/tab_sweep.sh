# cn_tab_lds_kernel: threads per workgroup and rows per chunk on the bench codes
set -e
for cfg in "512 8" "512 4" "1024 4"; do
  set -- $cfg
  echo "NT=$1 CH=$2"
  for code in peg_16000 32000 dvb 1057 Frolov; do
    LDPC_HIP_TAB_NT=$1 LDPC_HIP_TAB_CH=$2 timeout -k 10 120 python tools/bench_codes.py $code SPA_TABLE 2>&1 | grep SPA_TABLE
  done
done
