# Round profile of the headline benchmark, run on the GPU box from the repo root:
#   bash tools/profile_round.sh r03
# -> gpurun_out/<tag>/: the bench line (with live counters, also written to profiles/<tag>_counters.json by
#    --save-counters), rocprofv3 --kernel-trace --stats of the same command, and the float legs.
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
timeout -k 10 400 python3 bench.py --save-counters $O/counters.json > $O/bench_q88.json 2> $O/bench_q88.err
timeout -k 10 300 python3 bench.py --alg f32 --no-cpu --no-configs --save-counters $O/counters.json > $O/bench_f32.json 2> $O/bench_f32.err
timeout -k 10 300 python3 bench.py --alg f64 --frames 16384 --no-cpu --no-configs --save-counters $O/counters.json > $O/bench_f64.json 2> $O/bench_f64.err
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/kt_q88 -o p --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-secondary --no-pmc --no-configs --no-verify > $O/kt_q88.log 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/kt_f32 -o p --output-format csv -- python3 bench.py --alg f32 --steps 4 --warmup 1 --no-cpu --no-secondary --no-pmc --no-configs --no-verify > $O/kt_f32.log 2>&1
cp $(find $O/kt_q88 -name "*kernel_stats.csv") $O/fused_kernel_stats.csv
cp $(find $O/kt_f32 -name "*kernel_stats.csv") $O/fused_f32_kernel_stats.csv
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/kt_f64 -o p --output-format csv -- python3 bench.py --alg f64 --frames 16384 --steps 4 --warmup 1 --no-cpu --no-secondary --no-pmc --no-configs --no-verify > $O/kt_f64.log 2>&1
cp $(find $O/kt_f64 -name "*kernel_stats.csv") $O/fused_f64_kernel_stats.csv
cut -c1-300 $O/bench_q88.json; head -3 $O/fused_kernel_stats.csv | cut -c1-200
