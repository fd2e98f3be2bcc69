import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith("{")][-1]
j=json.loads(l); t=j["train"]
print(j["ms_per_step"], t["ms_per_step"]); print({k:(v["avg_us"],v["ms_per_step"]) for k,v in t["kernels"].items()})
