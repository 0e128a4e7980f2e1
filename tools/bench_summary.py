import json,sys
d=json.load(open(sys.argv[1]))
print("value",d["value"],"ms/step",d["ms_per_step"],"cpus_busy",d["config"].get("host_cpus_busy"))
print("tile",d["roofline"]["avg_launch_ms"],"serial",{k:round(v,3) for k,v in d["gpu_ms_per_step_by_kernel_group_serial"].items()})
for k in ("nms_ties_leg","pcie_inclusive","pcie_inclusive_nv12","latency_1frame"):
    v=d.get(k,{}); print(k, {a:b for a,b in v.items() if a not in ("note","sample","unit","h2d_bytes_per_step","host_cores")})
