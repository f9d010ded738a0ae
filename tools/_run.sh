timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 900 python bench.py > gpurun_out/r4_bench_a.json 2> gpurun_out/r4_bench_a.err; tail -c 600 gpurun_out/r4_bench_a.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_bench_a.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["minsnap"])
PY
