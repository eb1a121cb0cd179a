#!/bin/bash
# Same-box comparison of library builds on the single-launch timings of tools/b16k_stamps.py:
#   tools/ab_kernels.sh cur prev head ...   (names under tactilesr_amd/lib/exp/, `cur` = the in-tree library), 2 rounds
for r in 1 2; do
 for v in "$@"; do
  if [ $v = cur ]; then unset TSR_LIB_OVERRIDE TSR_ALLOW_VARIANT; else export TSR_ALLOW_VARIANT=1 TSR_LIB_OVERRIDE=tactilesr_amd/lib/exp/$v/libtactilesr_hip.so; fi
  echo "== $v $r"; python tools/b16k_stamps.py 2>/dev/null | grep "ms/launch" | cut -c1-60
 done
done
