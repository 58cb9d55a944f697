#!/bin/bash
# engine / memory clocks and power while the default bench runs and while a plain fill runs (is the stream's rate tied to a clock that drops under the chains?)
mkdir -p gpurun_out
out=gpurun_out/r04_clock_probe.txt
: > $out
ls /sys/class/drm/ | head -20 >> $out
probe() {
  for i in $(seq 1 $1); do
    line="$(date +%s.%N | cut -c1-14)"
    for f in /sys/class/drm/card*/device/pp_dpm_sclk /sys/class/drm/card*/device/pp_dpm_mclk /sys/class/drm/card*/device/pp_dpm_fclk; do
      [ -r $f ] && line="$line $(basename $f):$(grep '\*' $f | tr -d '\n')"
    done
    for f in /sys/class/drm/card*/device/hwmon/hwmon*/power1_average /sys/class/drm/card*/device/hwmon/hwmon*/power1_input /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input; do
      [ -r $f ] && line="$line $(basename $f):$(cat $f)"
    done
    echo "$line" >> $out
    sleep 0.1
  done
}
echo "== idle" >> $out; probe 5
echo "== rocm-smi" >> $out; (rocm-smi --showclocks --showpower 2>&1 | head -30) >> $out
echo "== fill lab (plain fills, all CUs)" >> $out
build/cu_mask_lab > /dev/null 2>&1 &
probe 12; wait
echo "== bench default (1024 steps)" >> $out
python bench.py --no-cpu-baseline --consumer-shard 0 > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
probe 400 &
wait %2 2>/dev/null
wait
(rocm-smi --showclocks --showpower 2>&1 | head -30) >> $out
tail -c 300 gpurun_out/clock_bench.json | head -c 300
