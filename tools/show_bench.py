"""Prints the numbers of a bench.py JSON line that are looked at after every run: python tools/show_bench.py <bench.json>"""
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["frame_pairs_per_s"], d["value"], d["roofline"]["frac"], d["timed_seconds"])
w=d["workloads"]
for k,v in w.items():
    if "error" in v: print(k, "ERROR", v["error"])
print("single", w["dvo_single_pair_vga"]["ms_per_call"], {k:(round(x,4) if isinstance(x,float) else x) for k,x in w["dvo_single_pair_vga"]["breakdown_ms"].items() if k!="note"})
print("720p", w["dvo_720p_x64"]["ms_per_step"], w["dvo_720p_x64"]["roofline"]["frac"])
print("stream", {k:v for k,v in w["dvo_stream_x256"].items() if ("ms" in k or "pairs" in k) and not isinstance(v,dict)})
sd=w["semi_dense_vga"]; print("sd", sd["frames_per_s"], sd["roofline"]["frac"], sd["roofline"]["kernel_ms"], sd["roofline_warp"]["frac"], sd["roofline_warp"]["kernel_ms"])
print("dropin", {k:v for k,v in w["semi_dense_dropin_vga"].items() if "ms" in k and not isinstance(v,dict)})
ba=w["ba_8x50k"]; print("ba", ba["lm_ms_per_iteration"], ba["latency_budget"]["sum_of_kernels_us_per_trial"])
