P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["whole_step"]["ms_per_step_by_kernel"])'
for r in 1 2; do
echo "== cur"; python bench.py --impl bf16 --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "$P"
echo "== var"; TSR_ALLOW_VARIANT=1 TSR_LIB_OVERRIDE=tactilesr_amd/lib/exp/$1/libtactilesr_hip.so python bench.py --impl bf16 --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "$P"
done
