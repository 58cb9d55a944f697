set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest7.log 2>&1; tail -3 gpurun_out/r03_gputest7.log
bash tools/prof_round3.sh > gpurun_out/r03prof.log 2>&1; tail -3 gpurun_out/r03prof.log
python - <<'PY'
import json
for f in ("bench_default","bench_20_5"):
    d=json.load(open("gpurun_out/r03prof/%s.json"%f))
    print(f, {k:d.get(k) for k in ("value","ms_per_step","witness_ok","value_consumer_mode","value_consumer_mode_steady")}, d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
PY
