# lanes of the window call, interleaved on one box (bench.py headline only)
for r in 1 2; do
for l in 1 2 3; do python bench.py --lanes $l --steps 120 --warmup 12 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench lanes $l:', d['ms_per_step'], d['device_ms_per_step_rank0'], d['graph_replay_ms'])"; done
done
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default lanes, 20 steps:', d['ms_per_step'], d['value'], d['host_path'], d['chained_batch8']['ms_per_window'], d['split_bf16']['ms_per_step'])"
