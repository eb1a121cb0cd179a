for r in 1 2; do
 for v in cur nodg; do
  if [ $v = cur ]; then unset TSR_LIB_OVERRIDE TSR_ALLOW_VARIANT; else export TSR_ALLOW_VARIANT=1 TSR_LIB_OVERRIDE=tactilesr_amd/lib/exp/$v/libtactilesr_hip.so; fi
  echo "== $v $r"; python bench.py --mode train --impl bf16 --batch 2048 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v for k,v in (d['whole_step'].get('ms_per_step_by_family') or {}).items() if 'dgrad' in k or 'conv' in k})"
 done
done
