# Developer aid (GPU box): S-noise native6 over pipelines x batch size
for f in 16 48; do for p in 1 2 3 6; do
echo -n "noise native6 frames $f pipelines $p: "; python bench.py --kind noise --workload native6 --no-cpu-baseline --no-latency --no-host-frames --frames-per-gpu $f --pipelines $p --steps 12 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
