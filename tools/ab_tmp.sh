P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["whole_step"]["ms_per_step_by_kernel"])'
for r in 1 2; do
echo "== pair"; python bench.py --impl bf16 --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "$P"
echo "== two launches"; python bench.py --impl bf16 --no-fuse-pair --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "$P"
done
