"""Developer tool: device-to-device copy bandwidth on this box (read + write bytes per second)."""
import torch
dev = torch.device('cuda:0')
for mb in (256, 1024, 4096):
    a = torch.empty(mb * 1024 * 1024 // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(5): b.copy_(a)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): b.copy_(a)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e-3
    print(f"{mb} MB copy: {2 * a.numel() * 4 / t / 1e12:.2f} TB/s (read+write)")
