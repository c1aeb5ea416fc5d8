import json,sys
d=json.load(open("bench_detail.json"))
for n,s in d["sub"].items():
    if n.startswith("train"):
        print(n,{k:s.get(k) for k in ("ms_per_step","eager_ms_per_step","hip_kernel_ms_per_step")}); print(s.get("top_kernels_ms_per_step")); print(s.get("roofline"))
