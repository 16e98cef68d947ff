# per-wavefront phase cycles of the timing builds named in TIMING_LIST (tools/probes/build_ablate.sh T P1 ...)
for N in ${TIMING_LIST:-T}; do
  echo "== build $N"
  LDPC_HIP_LIB=$GRAFT_REPO_ROOT/build_ablate/libldpc_hip_abl$N.so LDPC_HIP_FUSED_TIMING=1 \
    timeout -k 10 120 python3 bench.py --no-cpu --no-secondary --no-pmc --no-configs --no-verify --steps 1 --warmup 1 2>&1 | grep -A17 "fused kernel cycles" | tail -18 | cut -c1-110 || exit 1
done
