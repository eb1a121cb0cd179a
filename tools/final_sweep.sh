set -e
O=gpurun_out/r02_final
rm -rf $O; mkdir -p $O
python bench.py > $O/eval.json 2> $O/eval.err
python bench.py --impl bf16 --steps 20 --no-cpu-baseline > $O/eval_bf16.json 2>> $O/eval.err
python bench.py --impl bf16x6 --steps 10 --no-cpu-baseline > $O/eval_bf16x6.json 2>> $O/eval.err
python bench.py --mode train > $O/train.json 2> $O/train.err
python bench.py --mode train --impl bf16 --steps 10 --no-cpu-baseline > $O/train_bf16.json 2>> $O/train.err
python bench.py --mode train --batch 8192 --steps 3 --warmup 1 --no-cpu-baseline > $O/train_8192.json 2>> $O/train.err
python bench.py --mode tpsf > $O/tpsf.json 2> $O/tpsf.err
python bench.py --seqs --steps 10 --no-cpu-baseline > $O/seqs_eval.json 2> $O/seqs.err
python bench.py --seqs --mode train --steps 5 --no-cpu-baseline > $O/seqs_train.json 2>> $O/seqs.err
echo sweep done
bash tools/collect_profiles.sh r02
