"""which box is this, and does it compute the same thing twice: torch fp16 / bf16 / fp32 matrix products repeated and compared bit for bit
(independent of this library), then the library's half-precision network kernels repeated (developer scratch)"""
import os, socket, subprocess, sys
import torch
print("host", socket.gethostname(), flush=True)
try:
    print(subprocess.run(["rocm-smi", "--showuniqueid", "--showserial"], capture_output=True, text=True, timeout=60).stdout.strip()[-400:], flush=True)
except Exception as e:
    print("rocm-smi:", e)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for dt in (torch.float16, torch.bfloat16, torch.float32):
    a = torch.randn(4096, 4096, device=dev, generator=g).to(dt)
    b = torch.randn(4096, 4096, device=dev, generator=g).to(dt)
    ref = a @ b
    bad = 0
    for r in range(int(os.environ.get("MM_REPS", "300"))):
        c = a @ b
        if not torch.equal(c, ref):
            bad += 1
            d = (c != ref)
            print("   %s repeat %d: %d entries differ, rows %s" % (dt, r, int(d.sum()), torch.nonzero(d.any(dim=1)).flatten()[:6].tolist()), flush=True)
    torch.cuda.synchronize()
    print("%s 4096^3 product: %d of 300 repeats differ from the first" % (dt, bad), flush=True)
