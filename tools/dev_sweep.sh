export GPU_MAX_HW_QUEUES=16
for so in 2 0; do
echo -n "noise native6 sibling $so: "; STR_ER_DEBUG_STATS=0 python bench.py --kind noise --workload native6 --no-cpu-baseline --no-latency --no-host-frames --sibling-order $so --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gpu_ms_per_step_by_kernel_group_serial'])"
done
STR_ER_DEBUG_STATS=1 python bench.py --kind noise --workload native6 --no-cpu-baseline --no-latency --no-host-frames --steps 2 --warmup 1 --pipelines 1 2>&1 | grep "str_er\]" | grep -v "tie plane" | tail -12
