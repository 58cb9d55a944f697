set -o pipefail
for cfg in "8 8 48" "16 4 24" "16 6 24" "24 3 16" "16 4 48"; do set -- $cfg
python tools/bench_configs.py --hash-n 64 --hash-chunk 64 --sign-n 64 --agg-n 64 --agg-steps 3 --multi-n 2 --multi-engine-n $1 --multi-engine-coalesce $2 --multi-engine-steps $3 2>gpurun_out/r03_me.err | tail -1 | cut -c40-420 | tee -a gpurun_out/r03_multi_engine_sweep.txt
done
