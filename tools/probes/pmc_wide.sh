# SQ / LDS counters of the wide-row binary32 kernel on 32000.2240.3.105 (tools/bench_codes.py: 2 560 frames, 10 iterations)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_wide
rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $O/a -o p --output-format csv -- python3 $R/tools/bench_codes.py 32000 MINSUM_F32 > $O/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $O/b -o p --output-format csv -- python3 $R/tools/bench_codes.py 32000 MINSUM_F32 > $O/b.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python3 $R/tools/bench_codes.py 32000 MINSUM_F32 > $O/kt.log 2>&1
for x in a b; do python3 $R/tools/pmc_sum.py $(find $O/$x -name "*counter_collection.csv") wide; done
head -3 $O/kt/p_kernel_stats.csv | cut -c1-200
