#!/bin/bash
# Round 5 (verdict item 7): what do the bf16 implicit-GEMM launches of config #5 (bf16 stage 2, 400 frames) wait for?  The layer table under the
# RY_IGEMM_DBG ablations (wrong results, timing only): 4 no output stores, 128 no operand loads inside the K loop, 8 no K loop at all.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_l; mkdir -p $O
for dbg in 0 4 128 8; do
  RY_IGEMM_DBG=$dbg timeout 300 python bench.py --no-cpu-baseline --no-extras --frames 400 --dtype bf16 --steps 20 --repeats 1 --layers-out $O/layers_bf16_dbg$dbg.txt --details-out $O/d.json > $O/b.json 2> $O/err.txt; echo "dbg $dbg exit $?"
done
python - <<'PY'
import re
O = 'gpurun_out/r5_l'
tabs = {}
for dbg in (0, 4, 128, 8):
    t = {}
    for ln in open('%s/layers_bf16_dbg%d.txt' % (O, dbg)):
        c = ln.split()
        if len(c) > 5 and c[0] == 'stage2' and c[2].startswith('ry_igemm'):
            t[(c[1], c[2])] = (float(c[4]), c[3])
    tabs[dbg] = t
out = ['# round 5: config #5 (bf16 stage 2, 400 frames, SYN-64), implicit-GEMM launches under RY_IGEMM_DBG (us per launch, HIP events, eager; wrong results, timing only)',
       '# %-11s %-42s %-10s %9s %12s %14s %10s' % ('layer', 'kernel', 'grid', 'as it is', 'no stores', 'no K-loop loads', 'no K loop')]
for k in tabs[0]:
    out.append('  %-11s %-42s %-10s %9.2f %12.2f %14.2f %10.2f' % (k[0], k[1], tabs[0][k][1], tabs[0][k][0], tabs[4].get(k, (0,))[0], tabs[128].get(k, (0,))[0], tabs[8].get(k, (0,))[0]))
open(O + '/bf16_ablate_n400.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
