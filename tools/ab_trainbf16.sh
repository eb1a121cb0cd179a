for r in 1 2; do
 for v in all masked off; do
  echo "== $v $r"; python tools/ab_dgrad.py $v --mode train --impl bf16 --batch 2048 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['whole_step'].get('ms_per_step_by_family'))"
 done
done
