P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["whole_step"].get("ms_per_step_by_kernel"))'
for v in ${ABL_LIST:-cur abl_nobar abl_now abl_nownobar abl_nobrd cur}; do
  echo "== $v"
  if [ $v = cur ]; then python bench.py --no-legs --no-cpu-baseline --steps 3 --warmup 1 | python -c "$P";
  else TSR_LIB_OVERRIDE=tactilesr_amd/lib/exp/$v/libtactilesr_hip.so python bench.py --no-legs --no-cpu-baseline --steps 3 --warmup 1 | python -c "$P"; fi
done
if [ -n "$ABL_ENV" ]; then echo "== env $ABL_ENV"; env $ABL_ENV python bench.py --no-legs --no-cpu-baseline --steps 3 --warmup 1 | python -c "$P"; fi
