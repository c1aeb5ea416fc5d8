"""Per-kernel HIP-event times of one full Multitask.predict (ranking + greedy decode) at the C3 shape.  python tools/decode_profile.py"""
import os
os.environ.setdefault("NIR_DEBUG_TUNABLES", "1")
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from context_attentive_ir_amd import lib, synth  # noqa: E402
from context_attentive_ir_amd.config import default_args  # noqa: E402
from context_attentive_ir_amd.detinit import fill_module_  # noqa: E402
from context_attentive_ir_amd.wrappers import Multitask  # noqa: E402

V = 100000
mt = Multitask(default_args("CARS", src_vocab_size=V))
fill_module_(mt.network, 1013)
mt.cuda()
MACRO = int(sys.argv[1]) if len(sys.argv) > 1 else 1          # python tools/decode_profile.py 8: the bench's macro-batch of 8 (predict_many)
exs = [{k: v.cuda() for k, v in synth.session_batch(16, 7, 10, 4, 64, V, seed=1 + i).items()} for i in range(MACRO)]
ex = exs[0]
mt.id_check_interval = 0
run = (lambda: mt.predict(ex)) if MACRO == 1 else (lambda: mt.predict_many(exs, suggest=True))
L = lib.load()
for _ in range(3):
    run()
torch.cuda.synchronize()
L.nir_debug_set_tunable(b"no_fork", 1)
L.nir_profile_enable(1)
N = 5
for _ in range(N):
    run()
torch.cuda.synchronize()
L.nir_profile_enable(0)
buf = ctypes.create_string_buffer(1 << 17)
L.nir_profile_report(buf, len(buf))
rows = []
for line in buf.value.decode().strip().splitlines():
    k, c, ms = line.rsplit(",", 2)
    rows.append((float(ms) / N * 1e3, int(c) / N, k))
tot = sum(r[0] for r in rows)
for us, cnt, k in sorted(rows, reverse=True):
    print("%9.1f us/predict  %5.1f launches  %s" % (us, cnt, k))
print("total %.1f us of kernels per predict" % tot)
