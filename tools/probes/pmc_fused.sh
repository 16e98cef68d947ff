# two SQ passes over one launch of the bench workload (fused Q8.8 kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_fused
rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS -d $O/a -o p --output-format csv -- python3 $R/bench.py --pmc-child --steps 1 --warmup 0 > $O/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS_ATOMIC SQ_INSTS_SALU SQ_INST_CYCLES_SALU -d $O/b -o p --output-format csv -- python3 $R/bench.py --pmc-child --steps 1 --warmup 0 > $O/b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH -d $O/c -o p --output-format csv -- python3 $R/bench.py --pmc-child --steps 1 --warmup 0 > $O/c.log 2>&1
for x in a b c; do python3 $R/tools/pmc_sum.py $(find $O/$x -name "*counter_collection.csv") fused; done
