export TMPDIR=/tmp TLB_OUT=gpurun_out/tlb
rm -rf gpurun_out/tlb; mkdir -p gpurun_out/tlb
# NOTE every pass is its own process: buffers (and their kinds) differ from pass to pass, so each pass writes its own times
i=0
for c in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_BUSY_sum"; do
  i=$((i+1)); export TLB_OUT=gpurun_out/tlb/p$i; mkdir -p $TLB_OUT
  timeout -s KILL 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $TLB_OUT/pass -o pmc -- python3 tools/scratch/tlb_by_buffer.py > $TLB_OUT/log.txt 2>&1
  mkdir -p $TLB_OUT/pass1; mv $TLB_OUT/pass/* $TLB_OUT/pass1/ 2>/dev/null
  python3 tools/scratch/tlb_by_buffer.py --report $TLB_OUT 2>&1 | cut -c1-200
  rm -rf $TLB_OUT/pass $TLB_OUT/pass1
done
