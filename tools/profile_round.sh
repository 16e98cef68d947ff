set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01b
mkdir -p $O
timeout -k 10 200 python3 bench.py > $O/bench_q88.json 2> $O/bench_q88.err
timeout -k 10 200 python3 bench.py --alg f32 > $O/bench_f32.json 2> $O/bench_f32.err
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $O/kt_q88 -o p --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-secondary > $O/kt_q88.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $O/kt_f32 -o p --output-format csv -- python3 bench.py --alg f32 --steps 4 --warmup 1 --no-cpu --no-secondary > $O/kt_f32.log 2>&1
for alg in q88 f32; do
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE -d $O/pf_$alg -o p --output-format csv -- python3 bench.py --alg $alg --steps 2 --warmup 0 --no-cpu --no-profile --no-secondary > $O/pf_$alg.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE -d $O/pw_$alg -o p --output-format csv -- python3 bench.py --alg $alg --steps 2 --warmup 0 --no-cpu --no-profile --no-secondary > $O/pw_$alg.log 2>&1
python3 tools/pmc_sum.py $O/pf_$alg/p_counter_collection.csv fused > $O/pmc_hbm_$alg.txt
python3 tools/pmc_sum.py $O/pw_$alg/p_counter_collection.csv fused >> $O/pmc_hbm_$alg.txt
done
cat $O/bench_q88.json | cut -c1-200; cat $O/bench_f32.json | cut -c1-200; cat $O/pmc_hbm_*.txt
