set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/gfsynd
mkdir -p $O
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $O/kt -o k --output-format csv -- python3 tools/bench_gf.py q4.sp.9000.6000.4500.1GF2P > $O/kt.log 2>&1
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$O/kt/k_kernel_stats.csv")))[:7]:
    print(r["Name"][:60], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
