import sys, json
d = json.loads(sys.stdin.read()); r = d["roofline"]
print(d["ms_per_step"], d["value"], d["dtype"]); print({k: r[k] for k in r if k != "by_kernel"})
