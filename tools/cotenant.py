"""A second process that holds K hardware queues on the same GPU and keeps them idle (or lightly busy) for T seconds:
the co-tenant experiment behind DESIGN.md section 5 ("what a neighbour does to the 16-slot executor").
    GPU_MAX_HW_QUEUES=K python tools/cotenant.py K T [busy]"""
import os
import sys
import time

k, t = int(sys.argv[1]), float(sys.argv[2])
busy = len(sys.argv) > 3 and sys.argv[3] == "busy"
os.environ["GPU_MAX_HW_QUEUES"] = str(max(k, 1))
import torch  # noqa: E402

dev = torch.device("cuda:0")
streams = [torch.cuda.Stream(device=dev) for _ in range(k)]
x = [torch.zeros(1 << 16, device=dev) for _ in range(k)]
for s, v in zip(streams, x):
    with torch.cuda.stream(s):
        v.add_(1.0)
torch.cuda.synchronize()
print("cotenant: %d streams live" % k, flush=True)
t0 = time.time()
while time.time() - t0 < t:
    if busy:
        for s, v in zip(streams, x):
            with torch.cuda.stream(s):
                v.add_(1.0)
        torch.cuda.synchronize()
    time.sleep(0.05)
