for r in 1 2; do
 for v in cur prev; do
  if [ $v = cur ]; then unset TSR_LIB_OVERRIDE TSR_ALLOW_VARIANT; else export TSR_ALLOW_VARIANT=1 TSR_LIB_OVERRIDE=tactilesr_amd/lib/exp/prev/libtactilesr_hip.so; fi
  echo "== $v $r"; python bench.py --no-cpu-baseline --no-legs --impl bf16 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['whole_step'].get('ms_per_step_by_kernel'))"
 done
done
