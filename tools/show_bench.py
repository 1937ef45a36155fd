"""dev: the headline numbers of a bench line.  python tools/show_bench.py FILE   (or the line on stdin)"""
import json
import sys

d = json.loads(open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read())
r = d["roofline"]
print(d["ms_per_step"], d["value"], d["dtype"])
print({k: r[k] for k in r if k != "by_kernel"})
