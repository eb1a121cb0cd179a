set -e
for sh in "5 128 128 1024" "3 64 64 1024" "3 128 64 1024"; do
python tools/wgrad_microbench.py $sh
done
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline
