bash tools/collect_profiles.sh r04 > gpurun_out/collect_r04.log 2>&1; tail -1 gpurun_out/collect_r04.log
