"""Which SIMD does each wave of a rollout-shaped workgroup get?  Raw view of a few CUs (tools/wave_census.hip), for
workgroups of 2, 3 and 4 waves at half-chip grids.  python3 tools/census_detail.py"""
import ctypes as C, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
lib = C.CDLL(os.path.join(ROOT, "tools", "libwave_census.so"))
lib.wave_census.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
dev = "cuda:0"
out = torch.zeros((8192, 2), dtype=torch.int32, device=dev)
arrived = torch.zeros((1,), dtype=torch.int32, device=dev)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)   # noqa: E731
for wgs, wpw in ((512, 2), (512, 3), (512, 4), (256, 4), (768, 2), (1024, 2), (341, 3), (682, 3)):
    assert lib.wave_census(st(), C.c_void_p(out.data_ptr()), C.c_void_p(arrived.data_ptr()), wgs, 64 * wpw, 25600) == 0
    torch.cuda.synchronize()
    h = out.cpu().numpy().astype(np.uint32)[:wpw * wgs]
    hw, xcc = h[:, 0], h[:, 1] & 0xF
    wave_slot, simd, cu, sh, se = hw & 0xF, (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    cu_key = (xcc.astype(np.int64) << 16) | (se << 12) | (sh << 11) | (cu << 4)
    by_cu = defaultdict(list)
    for i in range(len(hw)):
        by_cu[int(cu_key[i])].append((i // wpw, i % wpw, int(simd[i]), int(wave_slot[i])))
    print(f"--- {wgs} workgroups x {wpw} waves: CUs used {len(by_cu)}")
    pat = defaultdict(int)
    for k, v in by_cu.items():
        v.sort()
        pat[tuple((w, s) for _, w, s, _ in v)] += 1
    for p, n in sorted(pat.items(), key=lambda x: -x[1])[:6]:
        print(f"   {n:4d} CUs: (wave-in-wg, simd) in workgroup order: {p}")
