"""Is a physically contiguous buffer slower for ANY write stream, or only for the sampler's?  hipMemsetD32Async (the
runtime's fill kernel) over 7.5 GB in a plain hipMalloc buffer and in a hipDeviceMallocContiguous one, alternating."""
import ctypes as C, sys
import torch
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemsetD32Async.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
torch.zeros(1, device="cuda:0")
nbytes = 7_530_905_096 // 4096 * 4096
st = torch.cuda.current_stream().cuda_stream
def timed(p, n=10):
    for _ in range(3): hip.hipMemsetD32Async(p, 1, nbytes // 4, C.c_void_p(st))
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): hip.hipMemsetD32Async(p, 1, nbytes // 4, C.c_void_p(st))
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
bufs = []
for i in range(8):
    p = C.c_void_p()
    contiguous = i % 2 == 1
    rc = hip.hipExtMallocWithFlags(C.byref(p), nbytes, 0x4) if contiguous else hip.hipMalloc(C.byref(p), nbytes)
    if rc: print("alloc error", rc); continue
    bufs.append(p)
    ms = timed(p)
    print(f"{'contiguous' if contiguous else 'plain     '} {p.value:#x}: fill {ms:.3f} ms = {nbytes / ms / 1e9:.2f} TB/s")
for p in bufs: hip.hipFree(p)
