"""Where do the waves of rollout-shaped workgroups (1 compute + 1 store wave, 25.6 KB LDS) land when the launch does NOT fill
the chip (B = 4 096 ... 32 768 -> 64 ... 512 workgroups for 1 024 SIMDs)?  Per grid size and LDS request: CUs used, workgroups
per CU, and how many SIMDs hold which mix of compute / store waves.  (tools/wave_census.hip)
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/wave_census.hip -o tools/libwave_census.so
    python3 tools/half_chip_census.py
"""
import ctypes as C, json, os, sys
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
lib = C.CDLL(os.path.join(ROOT, "tools", "libwave_census.so"))
lib.wave_census.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
lib.noop.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
dev = "cuda:0"
out = torch.zeros((4096, 2), dtype=torch.int32, device=dev)
arrived = torch.zeros((1,), dtype=torch.int32, device=dev)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)   # noqa: E731
for wgs in (64, 128, 256, 512, 768, 1024):
    for lds in (25600, 33792, 57344, 86016):
        for align in (0, 1):
            if align:
                lib.noop(st(), wgs, 128, 0)
            assert lib.wave_census(st(), C.c_void_p(out.data_ptr()), C.c_void_p(arrived.data_ptr()), wgs, 128, lds) == 0
            torch.cuda.synchronize()
            h = out.cpu().numpy().astype(np.uint32)[:2 * wgs]
            hw, xcc = h[:, 0], h[:, 1] & 0xF
            simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
            cu_key = (xcc.astype(np.int64) << 16) | (se << 12) | (sh << 11) | (cu << 4)
            key = cu_key | simd
            role = np.arange(len(key)) % 2
            per_simd = {}
            for k, r in zip(key.tolist(), role.tolist()):
                per_simd.setdefault(k, [0, 0])[r] += 1
            wg_per_cu = Counter(Counter(cu_key[::2].tolist()).values())
            mix = Counter(f"{c}+{s}" for c, s in per_simd.values())
            print(json.dumps({"workgroups": wgs, "lds": lds, "after_2wave_noop": align, "arrived": int(arrived.item()),
                              "cus_used": len(set(cu_key.tolist())), "xcds_used": len(set(xcc.tolist())),
                              "workgroups_per_cu": dict(sorted(wg_per_cu.items())),
                              "simds_by_compute+store": dict(sorted(mix.items()))}), flush=True)
