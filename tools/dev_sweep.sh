# Developer aid (GPU box): frames per batch x batches in flight
for f in 32 48 64 96; do for p in 4 6; do
echo -n "frames $f pipelines $p: "; python bench.py --no-cpu-baseline --no-latency --no-host-frames --frames-per-gpu $f --pipelines $p --steps 24 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
