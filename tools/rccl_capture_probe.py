"""Which RCCL collectives can a 1-rank process group run eagerly / under hipGraph capture on this box?  Each case in its own subprocess
(a crash must not take the others down).   python tools/rccl_capture_probe.py"""
import os
import subprocess
import sys

CASES = ["eager_allgather", "eager_a2a", "capture_allgather", "capture_a2a", "capture_allgather_branch", "capture_a2a_branch"]
BODY = r'''
import os, sys, time, torch, torch.distributed as dist
case = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
a = torch.arange(64, device="cuda", dtype=torch.float32).view(1, 64); b = torch.zeros_like(a)
def coll():
    if "a2a" in case:
        dist.all_to_all_single(b, a)
    else:
        dist.all_gather_into_tensor(b, a)
coll(); torch.cuda.synchronize()
assert torch.equal(a, b)
if case.startswith("capture"):
    time.sleep(0.3)
    s = torch.cuda.Stream(); side = torch.cuda.Stream()
    b.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        if case.endswith("branch"):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                coll()
            main.wait_stream(side)
        else:
            coll()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(a, b)
dist.destroy_process_group()
print("CASE", case, "ok")
'''
for c in CASES:
    r = subprocess.run([sys.executable, "-c", BODY, c], capture_output=True, text=True, timeout=300)
    tail = (r.stdout + r.stderr).strip().splitlines()[-3:]
    print("== %s rc=%d :: %s" % (c, r.returncode, " | ".join(tail)), flush=True)
