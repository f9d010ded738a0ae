#!/bin/bash
# Everything profiles/ holds for a round, collected on the GPU box into gpurun_out/profiles_<tag>/ (copy what is to be
# judged into profiles/ afterwards).   bash tools/collect_profiles.sh r06
set -u
TAG=${1:-r06}
export UAVAC_PROFILE_TAG=$TAG
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/profiles_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_mode.sh 2>&1 | grep -E "Unique|sampler" | head -2 > "$OUT/box.txt"
# 1. kernel trace + stats of the bench command (HIP-event average in the JSON line must agree with the trace average)
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-config4 > "$OUT/${TAG}_bench_profiled.json" 2> "$OUT/trace.err"
python3 tools/summarize_trace.py "$(ls $OUT/trace/*kernel_trace.csv | head -1)" "$OUT/${TAG}_bench" > "$OUT/per_dispatch_summary.txt"
# 2. PMC traffic (three passes) -> hbm_traffic.json + pmc summary
timeout -s KILL 900 python3 tools/pmc_traffic.py > "$OUT/pmc_traffic.log" 2>&1
cp gpurun_out/hbm_traffic.json "$OUT/hbm_traffic.json"; cp gpurun_out/${TAG}_pmc_summary.csv "$OUT/${TAG}_pmc_summary.csv"
cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json      # so that the bench of step 3 quotes it
# 3. the plain bench line (what the driver runs)
python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err"
# 4. other BASELINE configurations
python3 tools/config_sweep.py > "$OUT/${TAG}_config_sweep.jsonl" 2>/dev/null
# 5. small batch: where the cycles go
python3 tools/small_batch_profile.py 2>/dev/null | tail -1 > "$OUT/${TAG}_small_batch.json"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/pmc_small" -o pmc -- python3 tools/small_batch_profile.py > /dev/null 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, collections, json, sys
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/pmc_small/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "control_rollout" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("<")[1].split(">")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
json.dump({k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in acc.items()}, open(f"{out}/{tag}_small_batch_pmc.json", "w"), indent=1)
PY
# 6. issue / exchange micro-probes and the host-facing paths
[ -x tools/dpp_exchange_probe.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/dpp_exchange_probe.hip -o tools/dpp_exchange_probe.bin
./tools/dpp_exchange_probe.bin > "$OUT/${TAG}_dpp_exchange_probe.txt" 2>&1
python3 tools/single_uav_loop.py 2>/dev/null | tail -4 > "$OUT/${TAG}_single_uav_loop.txt"
python3 tools/host_path_rate.py 2>/dev/null | tail -2 > "$OUT/${TAG}_host_path_rate.txt"
python3 tools/first_launch_bisect.py 2>/dev/null | grep "^|" > "$OUT/${TAG}_first_launch_bisect.md"
# 7. round 3: effective clock of the rollout by batch size, obstacle-loop rate
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_rclock" -o pmc -- python3 tools/rollout_clock.py > /dev/null 2>&1
python3 tools/rollout_clock.py --report "$OUT/pmc_rclock" > "$OUT/${TAG}_rollout_clock.jsonl" 2>/dev/null
python3 tools/replan_rate.py 4096 2>/dev/null | grep -v amdgpu > "$OUT/${TAG}_replan_rate.txt"
# 8. round 4: the two sampler kernels on six row buffers side by side; plan-fed against row-fed rollout by batch size
python3 tools/sampler_stream_ab.py 8 65536 12 1x1,4x1,2x1,8x1,4x2 2>/dev/null | grep -v amdgpu > "$OUT/${TAG}_sampler_stream_ab.jsonl"
python3 tools/plan_vs_rows.py 2>/dev/null | grep "B=" > "$OUT/${TAG}_plan_vs_rows.txt"
python3 tools/solve_time.py 2>/dev/null | grep "B=" > "$OUT/${TAG}_solve_park.txt"
# 9. round 5: the tick below a full chip from the un-instrumented build (time + in-kernel clock per batch size, three counter passes),
#    the launcher's options over batch size, config 4's 5 000 ticks as five launches and as one
bash tools/collect_tick_budget.sh "$TAG" > /dev/null 2>&1
cp gpurun_out/tick_budget_$TAG/*.jsonl "$OUT/" 2>/dev/null
SWEEP=launcher python3 tools/half_chip_options.py 8 16384 24576 32768 35237 49152 2>/dev/null | grep "^{" > "$OUT/${TAG}_launcher_options_m8.jsonl"
python3 tools/config4_chunking.py 32768 2>/dev/null | grep "^{" > "$OUT/${TAG}_config4_chunking.jsonl"
rm -rf "$OUT/trace/"*results.db "$OUT/pmc_small" "$OUT/pmc_rclock"
ls -la "$OUT"
# 10. round 6: config 4 with the rows sampled once -- a peer's chain with rows / rows-free, the root's sampling beside its flight
python3 tools/config4_rows_free.py 2>/dev/null | grep "^{" > "$OUT/${TAG}_config4_rows_free.jsonl"
ls -la "$OUT"
