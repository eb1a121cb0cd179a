#!/bin/bash
# Profile recipe of a round (run on the GPU box through gpurun; outputs under gpurun_out/prof_$TAG, condensed into
# profiles/ by tools/summarize_prof.py).  Kernel timing and PMC counters are collected in SEPARATE runs; FETCH_SIZE and
# WRITE_SIZE in separate passes (TCC slot limit); the program itself follows `--` (no env/bash hop under rocprofv3).
#   usage: tools/collect_profiles.sh r02 [groups]     groups: any of "eval train bf16 trainbf16 tpsf" (default all; a second
#   call with other groups adds to the same directory -- one gpurun call is limited to 20 minutes)
set -e -o pipefail
TAG=${1:-rXX}
GROUPS_=${2:-eval train bf16 trainbf16 tpsf}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
# what the counters were taken on: bench.py nulls roofline.traffic when the kernel sources no longer match
python3 -c "import json,sys; sys.path.insert(0,'.'); from tactilesr_amd import build as b; json.dump({'csrc_sha16': b.source_hash(), 'files': b.source_hashes()}, open('$OUT/meta.json','w'))"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # name, rocprof args..., -- bench args
  local name=$1; shift
  case " $GROUPS_ " in *" ${name%%_*} "*) ;; *) return 0;; esac
  echo "== $name"
  rm -rf $OUT/$name
  rocprofv3 "$@" > $OUT/$name.log 2>&1
}
EVAL="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs"
TRAIN="python3 bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline"
BF16="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs --impl bf16"
TRAINBF16="python3 bench.py --mode train --impl bf16 --steps 3 --warmup 1 --no-cpu-baseline"
TPSF="python3 bench.py --mode tpsf --steps 5 --warmup 1 --no-cpu-baseline"
run eval_stats   --kernel-trace --stats --output-format csv -d $OUT/eval_stats  -- $EVAL
run eval_fetch   --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/eval_fetch -- $EVAL
run eval_write   --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/eval_write -- $EVAL
run eval_sq      --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/eval_sq -- $EVAL
run eval_clk     --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/eval_clk -- $EVAL
run train_stats  --kernel-trace --stats --output-format csv -d $OUT/train_stats -- $TRAIN
run train_fetch  --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/train_fetch -- $TRAIN
run train_write  --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/train_write -- $TRAIN
run train_sq     --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/train_sq -- $TRAIN
run train_clk    --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/train_clk -- $TRAIN
run bf16_stats   --kernel-trace --stats --output-format csv -d $OUT/bf16_stats  -- $BF16
run bf16_fetch   --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/bf16_fetch -- $BF16
run bf16_write   --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/bf16_write -- $BF16
run bf16_sq      --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/bf16_sq -- $BF16
run bf16_clk     --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/bf16_clk -- $BF16
run trainbf16_stats --kernel-trace --stats --output-format csv -d $OUT/trainbf16_stats -- $TRAINBF16
run trainbf16_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/trainbf16_fetch -- $TRAINBF16
run trainbf16_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/trainbf16_write -- $TRAINBF16
run tpsf_stats   --kernel-trace --stats --output-format csv -d $OUT/tpsf_stats  -- $TPSF
run tpsf_fetch   --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/tpsf_fetch -- $TPSF
run tpsf_write   --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/tpsf_write -- $TPSF
# the raw per-dispatch traces are large: keep the stats and counter files only
find $OUT -name '*_kernel_trace.csv' -size +2M -delete || true
find $OUT -name '*_agent_info.csv' -delete || true
echo done
