cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 16384 32768 65536; do
  rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_if_a_$B -o pmc -- python3 tools/small_batch_profile.py $B > /dev/null 2>&1
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB --kernel-trace --output-format csv -d gpurun_out/pmc_if_b_$B -o pmc -- python3 tools/small_batch_profile.py $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_if_c_$B -o pmc -- python3 tools/small_batch_profile.py $B > /dev/null 2>&1
done
ls gpurun_out | grep pmc_if
