#!/bin/bash
# round 6, call j: when do the DMA requests of an iteration go out?  RY_WINO_ESTEPS = 1 / 3 (first steps) against the product (spread over the nine position steps), fixed plans
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_j; mkdir -p $O
A='12:1:2:3,13:1:2:3,14:1:2:1,1:1:2:1,3:1:2:5'; B='12:3:2:3,13:4:4:3,3:3:2:5,11:3:2:5'; C='3:3:1:1,12:1:2:1'
for s in - _es1 _es3 - _es1 _es3; do timeout 300 python scripts/gpu_r6_var.py $s 300 $A $B $C 2>>$O/err.txt | grep "^var" >> $O/var.txt; done
cat $O/var.txt
