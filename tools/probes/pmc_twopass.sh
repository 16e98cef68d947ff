# SQ counters of the two-pass Q8.8 passes on the bench workload (PEG-16000, 32 768 frames, one step)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_twopass
rm -rf $O; mkdir -p $O
export LDPC_HIP_FUSED=0 LDPC_HIP_CHAIN=0
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/a -o p --output-format csv -- python3 $R/bench.py --pmc-child --frames 32768 --steps 1 --warmup 0 > $O/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES -d $O/b -o p --output-format csv -- python3 $R/bench.py --pmc-child --frames 32768 --steps 1 --warmup 0 > $O/b.log 2>&1
for x in a b; do python3 $R/tools/pmc_sum.py $(find $O/$x -name "*counter_collection.csv") kernel; done
