cd "$GRAFT_REPO_ROOT"
for n in 4 8 12; do for g in 1 2; do
echo -n "windows $n groups $g: "; SE2GPU_BA_BATCH_GROUPS=$g timeout 120 python bench.py --steps 20 --warmup 10 --no-orb --no-cpu-baseline --ba-windows $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ba_windows']['best']['iters_per_s']))"
done; done
