import json,sys
d=json.load(open(sys.argv[1]))["tv"]
print(sys.argv[1], round(d["train_sessions_per_s"]), round(d["ivectors_per_s"]))
print("  train", {k:v for k,v in d["train_kernels_ms"].items() if v>1.5})
