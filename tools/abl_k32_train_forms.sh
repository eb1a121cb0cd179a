P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline_conv"]["avg_launch_ms"])'
B="python bench.py --mode train --batch 2048 --no-legs --no-cpu-baseline --steps 3 --warmup 1"
for r in 1 2; do
echo "== cur"; $B | python -c "$P"
echo "== 256"; TSR_CONV_K32_256=1 $B | python -c "$P"
echo "== extdma"; TSR_LIB_OVERRIDE=tactilesr_amd/lib/exp/extdma/libtactilesr_hip.so $B | python -c "$P"
echo "== extdma256"; TSR_CONV_K32_256=1 TSR_LIB_OVERRIDE=tactilesr_amd/lib/exp/extdma/libtactilesr_hip.so $B | python -c "$P"
done
