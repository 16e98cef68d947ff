# kernel time and SQ / HBM counters of the staircase binary32 decoder on DVB-S2 0.5 (tools/bench_codes.py, 2 048 frames, 10 iterations)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_stair
rm -rf $O; mkdir -p $O
export BENCH_CODES_FRAMES=2048
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python3 $R/tools/bench_codes.py dvb MINSUM_F32 > $O/kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $O/a -o p --output-format csv -- python3 $R/tools/bench_codes.py dvb MINSUM_F32 > $O/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/b -o p --output-format csv -- python3 $R/tools/bench_codes.py dvb MINSUM_F32 > $O/b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/c -o p --output-format csv -- python3 $R/tools/bench_codes.py dvb MINSUM_F32 > $O/c.log 2>&1
head -2 $O/kt/p_kernel_stats.csv | cut -c1-200
for x in a b c; do python3 $R/tools/pmc_sum.py $(find $O/$x -name "*counter_collection.csv") stair; done
