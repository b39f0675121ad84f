#!/bin/bash
# planner check for the small stage-2 layers (tuning aid): per-layer time (implicit GEMM + reduce launch) with the plan of one
# layer fixed through RY_PLAN=layer:tile:splits:kgroups (tile codes of include/ry355.h) against the planner's own choice
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/plansweep.txt; : > $OUT
run() {  # label, layer name, RY_PLAN value
  RY_PLAN="$3" python bench.py --profile-only --profile-reps 10 --layers-out /tmp/l.txt >/dev/null 2>&1
  echo "$1 $(grep "$2 " /tmp/l.txt | awk '{printf "%s %s %sus | ", $2, $3, $4; t += $4} END {printf "sum %.2f us", t}')" >> $OUT
}
python bench.py --profile-only --profile-reps 10 --layers-out /tmp/p.txt >/dev/null 2>&1
for l in encoder/c5 encoder/c6 encoder/c7 decoder/c0 decoder/c1 decoder/c2; do
  echo "planner $l $(grep "$l " /tmp/p.txt | awk '{printf "%s %s %sus | ", $2, $3, $4; t += $4} END {printf "sum %.2f us", t}')" >> $OUT
done
for s in 8 16 32 64; do run "enc5 64x128 k1 s$s" "encoder/c5" "5:3:$s:1"; done
for s in 4 8 16; do run "enc5 64x128 k2 s$s" "encoder/c5" "5:3:$s:2"; run "enc5 96x128 k2 s$s" "encoder/c5" "5:6:$s:2"; done
for s in 32 64 256; do run "enc6 64x128 k1 s$s" "encoder/c6" "6:3:$s:1"; done
for s in 32 64; do run "enc6 32x128 k1 s$s" "encoder/c6" "6:4:$s:1"; done
for s in 32 64 256; do run "enc7 32x128 k1 s$s" "encoder/c7" "7:4:$s:1"; done
for s in 8 16 64; do run "dec0 32x128 k1 s$s" "decoder/c0" "8:4:$s:1"; done
for s in 8 16 32; do run "dec1 64x128 k1 s$s" "decoder/c1" "9:3:$s:1"; done
for s in 8 16 32; do run "dec1 32x128 k1 s$s" "decoder/c1" "9:4:$s:1"; done
for s in 2 4 16; do run "dec2 96x128 k2 s$s" "decoder/c2" "10:6:$s:2"; done
for s in 4 8 16; do run "dec2 64x128 k1 s$s" "decoder/c2" "10:3:$s:1"; run "dec2 128x128 k1 s$s" "decoder/c2" "10:1:$s:1"; done
cat $OUT
