for alg in q88 f32; do for it in 0 1 5 10; do
python bench.py --alg $alg --steps 8 --no-cpu --no-secondary --iters $it 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$alg', $it, d['ms_per_step'])"
done; done
