#!/bin/bash
# Ordered per-dispatch durations of the LAST step of a bench command: tools/prof_trace.sh NAME <bench args...>
#   -> gpurun_out/prof/NAME_last_step.txt
set -e -o pipefail
NAME=$1; shift
OUT=gpurun_out/prof/$NAME
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --no-legs "$@" > $OUT/run.log 2>&1
f=$(find $OUT -name '*_kernel_trace.csv' | head -1)
python3 - "$f" > gpurun_out/prof/${NAME}_last_step.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step ends with the optimizer kernel (train) -- else print the last 400 dispatches
ends = [i for i, r in enumerate(rows) if 'adam_l2_multi' in r['Kernel_Name']]
lo, hi = (ends[-2] + 1, ends[-1] + 1) if len(ends) >= 2 else (max(0, len(rows) - 400), len(rows))
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:hi]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%9.1f us  +%8.1f  %s  grid %s wg %s" % ((s - t0) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:90], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', ''))))
PY
rm -rf $OUT
