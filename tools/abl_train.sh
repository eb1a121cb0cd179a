#!/bin/bash
# Same-box A/B of library variants (tools/build_variant.py NAME -D...) on the train step:
#   ABL_LIST="cur NAME cur NAME" tools/abl_train.sh [extra bench args, e.g. --impl bf16]
# prints ms/step, the loss (a variant that changes results shows here) and the per-launch times by layer shape.
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "loss", d.get("loss"), d["whole_step"]["ms_per_launch"])'
for v in ${ABL_LIST:-cur}; do
  echo "== $v"
  if [ $v = cur ]; then python bench.py --mode train --batch 2048 --no-legs --no-cpu-baseline --steps 3 --warmup 1 "$@" | python -c "$P";
  else TSR_ALLOW_VARIANT=1 TSR_LIB_OVERRIDE=tactilesr_amd/lib/exp/$v/libtactilesr_hip.so python bench.py --mode train --batch 2048 --no-legs --no-cpu-baseline --steps 3 --warmup 1 "$@" | python -c "$P"; fi
done
