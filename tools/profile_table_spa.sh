# Sum-product on the two-pass engine: kernel times and SQ / LDS counters, each counter pass a run of its own.
#   bash tools/profile_table_spa.sh [code-substring [ALG [kernel-substring [out-dir]]]]
# defaults: PEG-16000, SPA_TABLE, cn_tab (cn_tab_lds_kernel), gpurun_out/tabspa/
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ALG=${2:-SPA_TABLE}
KERN=${3:-cn_tab}
O=${4:-gpurun_out/tabspa}
mkdir -p $O
CODE=${1:-peg_16000}
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $O/kt -o k --output-format csv -- python3 tools/bench_codes.py $CODE $ALG > $O/kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d $O/p1 -o p --output-format csv -- python3 tools/bench_codes.py $CODE $ALG > $O/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES -d $O/p2 -o p --output-format csv -- python3 tools/bench_codes.py $CODE $ALG > $O/p2.log 2>&1
python3 - <<PY > $O/summary.txt
import csv, collections
rows = list(csv.DictReader(open("$O/kt/k_kernel_stats.csv")))
for r in rows[:6]:
    print(r["Name"][:70], "calls", r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3, "pct", r["Percentage"])
PY
python3 tools/pmc_sum.py $O/p1/p_counter_collection.csv $KERN >> $O/summary.txt
python3 tools/pmc_sum.py $O/p2/p_counter_collection.csv $KERN >> $O/summary.txt
cat $O/summary.txt
