import json
d=json.loads(open("gpurun_out/r6A/default.json").read().strip().split("\n")[-1])
print("value", d["value"]); r=d["roofline"]; print({k:r[k] for k in ("kernel","frac")}, {k:r["binding"][k] for k in ("achieved_B_per_clk_per_cu","frac","lds_idx_active_over_sq_busy")})
for k,v in d["kernels"].items():
    if "binding" in v: print(k, round(v["ms_per_step"],4), round(v["tflops"],1), round(v["mfma_frac"],3), {a:(round(b,3) if b is not None else None) for a,b in v["binding"].items()})
t=json.loads(open("gpurun_out/r6A/teacher.json").read().strip().split("\n")[-1])
print("teacher", t["value"])
for k,v in t["roofline"]["kernels"].items(): print(k, round(v["ms_per_step"],3), round(v["tflops"],1), round(v["mfma_frac"],3), {a:(round(b,3) if b is not None else None) for a,b in v.get("binding",{}).items()})
