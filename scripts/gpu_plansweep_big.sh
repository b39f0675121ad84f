#!/bin/bash
# planner check for the MFMA-bound stage-2 layers (tuning aid): every tile x K-group choice with the planner's own split count,
# one layer at a time through RY_PLAN=layer:tile:splits:kgroups; FRAMES (default 300) picks the window size
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
F=${FRAMES:-300}
OUT=gpurun_out/plansweep_big_$F.txt; : > $OUT
fmt='{printf "%s %s %sus | ", $2, $3, $4; t += $4} END {printf "sum %.2f us", t}'
python bench.py --frames $F --profile-only --profile-reps 10 --layers-out /tmp/p.txt >/dev/null 2>&1
names=(x encoder/c1 encoder/c2 encoder/c3 encoder/c4 x x x x x x decoder/c3 decoder/c4 decoder/c5 decoder/c6)
for i in 1 2 3 4 11 12 13 14; do
  l=${names[$i]}
  echo "planner $l $(grep "$l " /tmp/p.txt | awk "$fmt")" >> $OUT
  if [ $i = 14 ]; then tiles="5 2"; else tiles="1 6 3"; fi
  for t in $tiles; do for kg in 1 2; do
    [ $t = 2 ] && [ $kg = 2 ] && continue
    RY_PLAN="$i:$t:0:$kg" python bench.py --frames $F --profile-only --profile-reps 10 --layers-out /tmp/l.txt >/dev/null 2>&1
    echo "  $l tile$t kg$kg $(grep "$l " /tmp/l.txt | awk "$fmt")" >> $OUT
  done; done
done
cat $OUT
