import sys, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from helpers import build_model
from context_attentive_ir_amd import lib
torch.manual_seed(0)
for (B, N, DL) in [(1, 1, 66), (1, 1, 290), (1, 4, 290), (2, 3, 130)]:
    V = 500
    m = build_model("DUET", vocab=V, device="cuda", max_query_len=4, max_doc_len=DL)
    g = torch.Generator().manual_seed(DL)
    q = torch.randint(4, V, (B, 4), generator=g).cuda(); d = torch.randint(4, V, (B, N, DL), generator=g).cuda()
    ql = torch.full((B,), 4).cuda(); dl = torch.full((B, N), DL).cuda()
    outs = [m(q, ql, d, dl, return_parts=True)[2].clone() for _ in range(3)]
    with lib.tunable("duet_unfused", 1, 0):
        ref = m(q, ql, d, dl, return_parts=True)[2].clone()
    print(B, N, DL, "run-to-run max diff", float((outs[0] - outs[1]).abs().max()), float((outs[0] - outs[2]).abs().max()),
          "vs unfused", (outs[0] - ref).abs().flatten().cpu().numpy().round(6))
