#!/bin/bash
# Round 5: the half-full chip's tick from an un-instrumented build -- time + clock per batch size, then three counter passes.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:-r05}
OUT=gpurun_out/tick_budget_$TAG
rm -rf $OUT; mkdir -p $OUT
python3 tools/tick_budget.py 8 2>/dev/null | grep '^{' > $OUT/${TAG}_tick_budget_m8.jsonl
python3 tools/tick_budget.py 12 16384 32768 49152 65536 2>/dev/null | grep '^{' > $OUT/${TAG}_tick_budget_m12.jsonl
for pass in 1 2 3; do
  timeout -s KILL 300 rocprofv3 --pmc $(python3 tools/tick_budget.py --counters $pass) --kernel-trace --output-format csv -d $OUT/pmc_$pass -o pmc -- python3 tools/tick_budget.py --counters-run > /dev/null 2>&1
done
python3 tools/tick_budget.py --report $OUT/pmc_1 $OUT/pmc_2 $OUT/pmc_3 > $OUT/${TAG}_tick_budget_counters.jsonl
rm -rf $OUT/pmc_*
cat $OUT/*.jsonl
