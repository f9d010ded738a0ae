#!/bin/bash
# Round 5: what couples the two workgroups of a CU?  Timing-only builds of the rollout (wrong results; /tmp/probe, not in the tree):
# probe1 = the store wave issues no global store, probe2 = ... and reads nothing from LDS, probe3 = ... and the compute wave writes no
# slab (the per-tick barrier is all that is left of the hand-over).  Against the product build, same process, alternating.
cd "${GRAFT_REPO_ROOT:-.}"
export AB_LATE_HANDOVER=${AB_LATE_HANDOVER:-1} AB_COEFF_DMA=${AB_COEFF_DMA:-1}
for n in 1 2 3; do for B in 16384 32768; do
  python3 tools/rollout_ab.py tools/ab/libuavac_probe$n.so $B plan 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'probe': $n, 'B': d['B'], 'product_ms': d['median_tree'], 'probe_ms': d['median_other']}))"
done; done
