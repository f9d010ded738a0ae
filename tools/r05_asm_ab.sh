#!/bin/bash
# Round 5, VERDICT item 1: the well-defined inline asm (m0 saved / restored in coeffs_dma, "+v" row prefetch) against round 4's
# library (tools/ab/libuavac_r04.so = `git archive ac2e051 | make`), same process, same buffers, alternating.
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05_asm_ab.jsonl
: > $OUT
if [ $# -eq 0 ]; then set -- "4096 rows" "8192 rows" "16384 rows" "4096 plan" "16384 plan" "32768 plan" "49152 plan" "65536 plan"; fi
for spec in "$@"; do
  python3 tools/rollout_ab.py tools/ab/libuavac_r04.so $spec 2>/dev/null | grep '^{' >> $OUT
done
cat $OUT
