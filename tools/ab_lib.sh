set -u
cd $GRAFT_REPO_ROOT
P=context_attentive_ir_amd
cp $P/libneuroir_hip.so /tmp/new.so
run() { timeout 300 python bench.py --steps 20 --warmup 5 --sub NS_cars_50 --no-cpu-baseline > /tmp/b.log 2>&1; python - <<PY
import json
l=[x for x in open("/tmp/b.log") if x.startswith("{")][-1]
d=json.loads(l); print("$1", d["value"], d["ms_per_step"], d["power"]["package_w"], [(s["name"], s["pairs_per_s"], s.get("w")) for s in d["sub"]])
PY
}
run new1
cp $P/libneuroir_hip_g3old.so $P/libneuroir_hip.so; run old1
cp /tmp/new.so $P/libneuroir_hip.so; run new2
cp $P/libneuroir_hip_g3old.so $P/libneuroir_hip.so; run old2
cp /tmp/new.so $P/libneuroir_hip.so
