for r in 1 2; do
 for v in ${AB_LIBS:-prev new}; do
  export TSR_LIB_OVERRIDE=tactilesr_amd/lib/exp/$v/libtactilesr_hip.so
  echo "== $v $r"
  python bench.py --no-cpu-baseline --no-legs --impl bf16 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16', d['value'], d['ms_per_step'])"
  python bench.py --no-cpu-baseline --no-legs --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp16x3', d['value'], d['ms_per_step'])"
  python bench.py --no-cpu-baseline --no-legs --seqs --impl bf16 --batch 512 --steps 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('seqs bf16', d['value'], d['ms_per_step'])"
 done
done
