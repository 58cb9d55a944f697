run() { # name lib variant
  export BLSW_EXPAND_VARIANT=$3
  if [ "$2" = default ]; then unset BLSW_LIB; else export BLSW_LIB=$PWD/build/$2; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --allgather-steps 0 $BENCH_ARGS > gpurun_out/ab_$1.json 2>gpurun_out/ab_$1.err && echo "$1 $(python -c "import json;d=json.load(open('gpurun_out/ab_$1.json'));print(round(d['value']),round(d['ms_per_step'],2),round(d['roofline']['avg_launch_ms'],2),d['config']['results_ok'])")"
}
