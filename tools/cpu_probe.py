import sys
sys.path.insert(0, '.')
from oracle import c_oracle as co
from oracle.cpu_baseline import effective_cpus
from oracle.minsnap_oracle import synthetic_missions
print(effective_cpus())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/proc/loadavg"):
    try: print(f, open(f).read().strip())
    except OSError as e: print(f, "absent")
wps = synthetic_missions(2048, 12)
for n in (1, 4, 16, 64, 128, 256):
    done, el = co.bench_threads(wps, 3.0, 0.01, 10000, n, 2.0)
    print(n, "threads:", done * 10000 / el / 1e6, "M steps/s")
