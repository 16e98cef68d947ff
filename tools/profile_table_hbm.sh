# HBM traffic of the table sum-product's kernels on PEG-16000 (separate FETCH_SIZE / WRITE_SIZE passes)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/tabspa
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/pf -o p --output-format csv -- python3 tools/bench_codes.py peg_16000 SPA_TABLE > $O/pf.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/pw -o p --output-format csv -- python3 tools/bench_codes.py peg_16000 SPA_TABLE > $O/pw.log 2>&1
python3 tools/pmc_sum.py $O/pf/p_counter_collection.csv ldpc > $O/pmc_hbm.txt
python3 tools/pmc_sum.py $O/pw/p_counter_collection.csv ldpc >> $O/pmc_hbm.txt
cat $O/pmc_hbm.txt
