for sh in "5 128 128" "3 128 128" "5 64 64" "3 64 64"; do
 for raw in 0 1; do WG_RAW=$raw python tools/wgrad_microbench.py $sh 2048 -1 2>&1 | tail -1; done
done
