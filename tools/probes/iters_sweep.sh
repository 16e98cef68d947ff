# ms per step of the bench workload against the number of iterations executed (fixed cost = the 0-iteration line)
for it in 0 1 2 10; do timeout -k 10 200 python3 bench.py --no-cpu --no-secondary --no-pmc --no-configs --no-verify --iters $it 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('iters', $it, d['ms_per_step'])"; done
