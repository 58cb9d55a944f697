for cfg in "team 3 16" "lane 3 16" "team 3 16" "lane 3 16"; do
  set -- $cfg
  BLSW_PAIRING=$1 timeout -k 10 200 python bench.py --steps 512 --warmup 48 --buffers $2 --coalesce $3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$cfg', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), d['config']['groups_in_flight'], d['config']['results_ok'])" >> gpurun_out/sweep.txt
done
