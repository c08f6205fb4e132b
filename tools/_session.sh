cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r4k/bench.json 2> gpurun_out/r4k/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r4k/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4k/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","blocking_value","resident_value","value_survey_8d","value_survey_8d_pipelined","kernel_ms_per_step","within_1pct_of_gt")}, d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
PY
