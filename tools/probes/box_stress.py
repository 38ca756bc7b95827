"""sustained matrix-core load, every product compared with the first on the device (no host round trip between launches): does this box
compute the same thing every time when it is hot?  Independent of this library (torch / hipBLASLt).  (developer scratch)"""
import os, socket, subprocess, sys, time
import torch
try:
    print(subprocess.run(["rocm-smi", "--showserial", "--showtemp", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=60).stdout.strip()[-900:], flush=True)
except Exception as e:
    print("rocm-smi:", e)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
reps = int(os.environ.get("MM_REPS", "20000"))
for dt in (torch.float16, torch.bfloat16):
    a = torch.randn(4096, 4096, device=dev, generator=g).to(dt)
    b = torch.randn(4096, 4096, device=dev, generator=g).to(dt)
    ref = a @ b
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    worst = torch.zeros((), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    for r in range(reps):
        c = a @ b
        n = (c != ref).sum()
        bad += (n > 0).to(torch.int64)
        worst = torch.maximum(worst, n)
    torch.cuda.synchronize()
    dt_s = time.time() - t0
    print("%s: %d of %d products differ from the first (most entries in one: %d), %.1f s, %.0f TFLOP/s" % (dt, int(bad), reps, int(worst), dt_s,
                                                                                                           reps * 2 * 4096 ** 3 / dt_s / 1e12), flush=True)
try:
    print(subprocess.run(["rocm-smi", "--showtemp", "--showpower"], capture_output=True, text=True, timeout=60).stdout.strip()[-500:], flush=True)
except Exception as e:
    print("rocm-smi:", e)
