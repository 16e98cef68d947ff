# Staircase Q8.8 decoder (chain_q88.hip) on DVB-S2 0.5, run on the GPU box from the repo root:
#   bash tools/profile_dvb_chain.sh [tag]
# -> gpurun_out/<tag>_dvb_chain/: throughput line, rocprofv3 --kernel-trace --stats, and separate
#    PMC passes (FETCH_SIZE; WRITE_SIZE; SQ counters) of the same command.
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG}_dvb_chain
mkdir -p $O
export BENCH_CODES_FRAMES=8192
timeout -k 10 200 python3 tools/bench_codes.py dvb MINSUM_Q88 > $O/bench.txt 2> $O/bench.err
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python3 tools/bench_codes.py dvb MINSUM_Q88 > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv") $O/kernel_stats.csv
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/pf -o p --output-format csv -- python3 tools/bench_codes.py dvb MINSUM_Q88 > $O/pf.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $O/pw -o p --output-format csv -- python3 tools/bench_codes.py dvb MINSUM_Q88 > $O/pw.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $O/ps -o p --output-format csv -- python3 tools/bench_codes.py dvb MINSUM_Q88 > $O/ps.log 2>&1
python3 tools/pmc_sum.py $O/pf/p_counter_collection.csv chain > $O/pmc.txt
python3 tools/pmc_sum.py $O/pw/p_counter_collection.csv chain >> $O/pmc.txt
python3 tools/pmc_sum.py $O/ps/p_counter_collection.csv chain >> $O/pmc.txt
cat $O/bench.txt; head -3 $O/kernel_stats.csv | cut -c1-200; cat $O/pmc.txt
