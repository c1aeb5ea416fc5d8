"""Time the phases of GraphedPredictor.predict (staging copy, H2D, replay) per model.   usage: python tools/graph_probe.py"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
import bench
from context_attentive_ir_amd.graph_runner import GraphedPredictor
class A: pass
for model, batch, cands, dlen, qlen in (("cars", 16, 10, 64, 4), ("duet", 64, 50, 290, 8), ("match_tensor", 32, 10, 64, 4)):
    sys.argv = ["bench.py", "--model", model, "--batch", str(batch), "--cands", str(cands), "--dlen", str(dlen), "--qlen", str(qlen)]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    m = bench.build(args, dev)
    batches = bench.make_batches(args, 0, dev)
    host = [{k: v.cpu().pin_memory() for k, v in b.items()} for b in batches]
    gp = GraphedPredictor(m, batches[0])
    def T(fn, n=30):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
    print(model, "bytes", gp.dev_buf.numel(), {k: (tuple(v.shape), str(v.dtype)) for k, v in batches[0].items()})
    print("  replay only        %.3f ms" % T(lambda: gp.graph.replay()))
    print("  predict(device ex) %.3f ms" % T(lambda: gp.predict(batches[1], clone=False)))
    print("  predict(host ex)   %.3f ms" % T(lambda: gp.predict(host[1], clone=False)))
    def hostcopy():
        for k, hv in gp._host_views.items(): hv.copy_(host[1][k])
    print("  host staging copy  %.3f ms" % T(hostcopy))
    print("  H2D only           %.3f ms" % T(lambda: gp.dev_buf.copy_(gp.host_buf, non_blocking=True)))
    # phase-by-phase wall clock of one host-input predict
    ph = [0.0] * 5
    for it in range(20):
        with torch.cuda.stream(gp.stream):
            t0 = time.perf_counter(); gp.stream.synchronize()
            t1 = time.perf_counter(); hostcopy()
            t2 = time.perf_counter(); gp.dev_buf.copy_(gp.host_buf, non_blocking=True)
            t3 = time.perf_counter(); gp.graph.replay()
            t4 = time.perf_counter(); gp.stream.synchronize()
            t5 = time.perf_counter()
        for i, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))):
            ph[i] += (b - a) * 1e3 / 20
    print("  phases ms: sync %.3f | staging %.3f | h2d enqueue %.3f | replay enqueue %.3f | wait %.3f" % tuple(ph))
    torch.set_num_threads(1)
    print("  predict(host ex), 1 torch thread %.3f ms" % T(lambda: gp.predict(host[1], clone=False)))
    torch.set_num_threads(64)
