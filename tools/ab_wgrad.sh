# same-box A/B of library variants on single bf16-tensor wgrad launches (no input transform): AB_VARIANTS="cur ring4 ..."
for sh in "5 128 128" "3 128 128" "5 64 64" "3 64 64"; do
 for v in ${AB_VARIANTS:-cur}; do
  if [ $v = cur ]; then unset TSR_LIB_OVERRIDE TSR_ALLOW_VARIANT; else export TSR_ALLOW_VARIANT=1 TSR_LIB_OVERRIDE=tactilesr_amd/lib/exp/$v/libtactilesr_hip.so; fi
  echo -n "$v: "; WG_RAW=${AB_RAW:-1} python tools/wgrad_microbench.py $sh ${AB_BATCH:-2048} ${AB_PLANES:--1} 2>/dev/null | tail -1
 done
done
