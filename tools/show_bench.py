import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
print({k:v for k,v in d["roofline"].items() if not isinstance(v,(dict,list))})
print("sched", d.get("sched"))
print("cpu", {k:v for k,v in d.get("cpu_baseline",{}).items() if k!="other_lines"})
ex=d.get("extra",{})
for k in ("config3_multi_restart_end_to_end","config5_multi_restart_end_to_end","config3_end_to_end_sched_example_seed42","config5_end_to_end_sched10k_window65536_seed42","sched_verified","sched10k_exhaustive_320k_host_round_trip_generated_on_device","sched10k_random_stream_window_65536","sched10k_exhaustive_320k","sched_example_window100_x64_restarts"):
    print(k, ex.get(k))
