# rocprofv3 passes over the non-binary engines (tools/bench_gf.py <case>): kernel times, HBM bytes,
# L2 / vector-L1 hit counters.  Usage on the GPU box: bash tools/profile_gf.sh <tag> <case-substring>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; CASE=$2
O=$R/gpurun_out/gf_$TAG
rm -rf $O; mkdir -p $O
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python3 $R/tools/bench_gf.py $CASE > $O/kt.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/pf -o p --output-format csv -- python3 $R/tools/bench_gf.py $CASE > $O/pf.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $O/pw -o p --output-format csv -- python3 $R/tools/bench_gf.py $CASE > $O/pw.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/l2 -o p --output-format csv -- python3 $R/tools/bench_gf.py $CASE > $O/l2.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum -d $O/l1 -o p --output-format csv -- python3 $R/tools/bench_gf.py $CASE > $O/l1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD -d $O/sq -o p --output-format csv -- python3 $R/tools/bench_gf.py $CASE > $O/sq.log 2>&1
head -12 $(find $O/kt -name "*kernel_stats.csv") > $O/summary.txt
for x in pf pw l2 l1 sq; do f=$(find $O/$x -name "*counter_collection.csv"); [ -n "$f" ] && python3 $R/tools/pmc_sum.py $f gf >> $O/summary.txt || tail -3 $O/$x.log >> $O/summary.txt; done
cat $O/summary.txt
