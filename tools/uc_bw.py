"""Developer tool: bandwidth of the local GPU reading / writing the kinds of device memory a peer-exchange receive
buffer can live in (hipDeviceMallocUncached as tgs_peer_alloc uses, hipDeviceMallocFinegrained, plain hipMalloc).
   python tools/uc_bw.py"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from touch_gs_amd.parallel import _RawDeviceArray
hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
n = 64 * 1024 * 1024           # floats (256 MB)
plain = torch.empty(n, device=dev)
flags = {"uncached": 0x3, "finegrained": 0x1, "default": 0x0}
for name, fl in flags.items():
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(4 * n), C.c_uint(fl))
    if rc != 0:
        print(name, "hipExtMallocWithFlags failed", rc); continue
    t = torch.as_tensor(_RawDeviceArray(p.value, n, "<f4"), device=dev)
    res = {}
    for what, fn in (("write", lambda: t.copy_(plain)), ("read", lambda: plain.copy_(t)), ("fill", lambda: t.fill_(1.0)),
                     ("sum", lambda: t.sum())):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        byt = 4 * n * (2 if what in ("write", "read") else 1)
        res[what] = f"{byt / ms / 1e6:.0f} GB/s"
    print(name, res, flush=True)
    hip.hipFree(p)
