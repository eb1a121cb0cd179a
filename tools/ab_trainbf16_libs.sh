# same-box A/B of library variants on the bf16-storage train step: AB_VARIANTS="cur name ..."
for r in 1 2; do
 for v in ${AB_VARIANTS:-cur}; do
  if [ $v = cur ]; then unset TSR_LIB_OVERRIDE TSR_ALLOW_VARIANT; else export TSR_ALLOW_VARIANT=1 TSR_LIB_OVERRIDE=tactilesr_amd/lib/exp/$v/libtactilesr_hip.so; fi
  echo -n "$v $r: "; python bench.py --mode train --impl bf16 --batch 2048 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['whole_step'].get('ms_per_step_by_family'), d['whole_step']['ms_per_launch'].get('fwd_k1_co64_ci256'))"
 done
done
