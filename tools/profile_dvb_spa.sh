# HBM traffic of the two-pass engine on DVB-S2 0.5, SPA_F32 (2176 frames per pass): separate
# FETCH_SIZE / WRITE_SIZE passes, as tools/profile_round.sh does for the headline kernels.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/dvb_spa
mkdir -p $O
export BENCH_CODES_BUDGET=4e9 BENCH_CODES_LLR=awgn
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/pf -o p --output-format csv -- python3 tools/bench_codes.py dvb SPA_F32 > $O/pf.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/pw -o p --output-format csv -- python3 tools/bench_codes.py dvb SPA_F32 > $O/pw.log 2>&1
python3 tools/pmc_sum.py $O/pf/p_counter_collection.csv ldpc > $O/pmc_hbm.txt
python3 tools/pmc_sum.py $O/pw/p_counter_collection.csv ldpc >> $O/pmc_hbm.txt
cat $O/pmc_hbm.txt
