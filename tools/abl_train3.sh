#!/bin/bash
# Same-box A/B of library variants on the train step, per-launch conv times: ABL_LIST="cur x ..." tools/abl_train3.sh [bench args]
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], {k: v for k, v in d["whole_step"]["ms_per_launch"].items() if not k.startswith("wgrad")})'
for v in ${ABL_LIST:-cur}; do
  echo "== $v"
  if [ $v = cur ]; then python bench.py --mode train --batch 2048 --no-legs --no-cpu-baseline --steps 3 --warmup 1 "$@" | python -c "$P";
  else TSR_LIB_OVERRIDE=tactilesr_amd/lib/exp/$v/libtactilesr_hip.so python bench.py --mode train --batch 2048 --no-legs --no-cpu-baseline --steps 3 --warmup 1 "$@" | python -c "$P"; fi
done
