import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
