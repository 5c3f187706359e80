import json
b = json.loads(open("gpurun_out/r06/bench_c2_final.json").read().strip().splitlines()[-1])
print(round(b["value"]), b["ms_per_step"], b["roofline"]["frac"])
rp = b["secondary"]["round6_pairs"]
if isinstance(rp, dict):
    print(rp)
else:
    for r in rp:
        print(r["pair"], "| single", r["us_per_frame_single"], r["frac_single"], "| lists", r["us_per_frame_lists_of_8"], r["frac_lists"], r["list_launches"])
