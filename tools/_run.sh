timeout 600 ./tools/vmm_va_probe.bin > gpurun_out/vmm_va.txt 2>&1; cat gpurun_out/vmm_va.txt
