# bench step time of the ablation builds named in ABL_LIST (tools/probes/build_ablate.sh; 0 = the product library)
for N in ${ABL_LIST:-0 1 2 3 4 7}; do
  if [ $N = 0 ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$GRAFT_REPO_ROOT/build_ablate/libldpc_hip_abl$N.so; fi
  timeout -k 10 120 python3 bench.py --no-cpu --no-secondary --no-pmc --no-configs --no-verify --steps 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate', '$N', d['ms_per_step'], 'avg_it', d['avg_iterations'])"
done
