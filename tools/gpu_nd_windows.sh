cd "$GRAFT_REPO_ROOT"
for nd in 1 0; do for n in 1 8 64; do
echo -n "ND=$nd windows $n: "; SE2GPU_BA_ND=$nd timeout 120 python bench.py --steps 20 --warmup 10 --no-orb --no-cpu-baseline --ba-windows $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ba_windows']['best']['iters_per_s']))"
done; done
