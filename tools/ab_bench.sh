#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab_bench.sh <other .so> [bench args...]  (alternating runs, 2 rounds)
OTHER=$1; shift
for r in 1 2; do
  echo "== cur $r"; python bench.py --no-cpu-baseline "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['whole_step'].get('ms_per_step_by_kernel') or d['whole_step'].get('ms_per_step_by_family'))"
  echo "== other $r"; TSR_ALLOW_VARIANT=1 TSR_LIB_OVERRIDE=$OTHER python bench.py --no-cpu-baseline "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['whole_step'].get('ms_per_step_by_kernel') or d['whole_step'].get('ms_per_step_by_family'))"
done
