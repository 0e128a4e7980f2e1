# Developer aid (GPU box): workgroups per plane in k_resolve / k_reduce (STR_ER_NODE_BLOCKS)
for nb in 6 12 24 48; do
echo -n "STR_ER_NODE_BLOCKS=$nb text: "; STR_ER_NODE_BLOCKS=$nb python bench.py --no-cpu-baseline --no-latency --no-host-frames --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); g=d['gpu_ms_per_step_by_kernel_group_serial']; print(d['value'], d['ms_per_step'], g['resolve'], g['accumulate'], g['select'])"
done
for nb in 12 32; do
echo -n "STR_ER_NODE_BLOCKS=$nb noise: "; STR_ER_NODE_BLOCKS=$nb python bench.py --kind noise --workload native6 --no-cpu-baseline --no-latency --no-host-frames --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); g=d['gpu_ms_per_step_by_kernel_group_serial']; print(d['value'], d['ms_per_step'], g['resolve'], g['accumulate'], g['select'])"
done
