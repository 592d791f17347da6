#!/bin/bash
# usage (GPU box): tools/soak_fresh.sh <runs> [tag]   -> gpurun_out/<tag>.log : <runs> FRESH python processes of tools/soak_case.py,
# cycling through {verify, plain} x {lock-step, per-stream batches}; every failure is kept in full.
cd "$GRAFT_REPO_ROOT" || exit 1
N=${1:-100}; TAG=${2:-soak}
mkdir -p gpurun_out; LOG=gpurun_out/$TAG.log; : > $LOG
echo "soak_fresh: $N fresh processes, commit ${GRAFT_COMMIT:-?}, $(date -u +%FT%TZ)" >> $LOG
ok=0; bad=0; t0=$(date +%s)
for i in $(seq 0 $((N-1))); do
  v=$(( i % 2 )); ls=$(( (i / 2) % 4 == 3 ? 0 : 1 ))
  out=$(SE2GPU_BA_CHOL_VERIFY=$v SE2GPU_BA_LOCKSTEP=$ls timeout 300 python tools/soak_case.py $i 2>&1); rc=$?
  if [ $rc -eq 0 ]; then ok=$((ok+1)); echo "$out" | tail -1 >> $LOG; else bad=$((bad+1)); { echo "run $i rc=$rc verify=$v lockstep=$ls"; echo "$out" | tail -40; } >> $LOG; fi
done
echo "soak_fresh: $ok ok, $bad failed, $(( $(date +%s) - t0 )) s" | tee -a $LOG
grep -c "SOAK ok" $LOG; grep "SOAK FAIL" -A12 $LOG | head -60
awk '/handoffs_checked/ {s+=$(NF-2); m+=$NF} END {print "hand-offs checked:", s, "mismatches:", m}' $LOG | tee -a $LOG
