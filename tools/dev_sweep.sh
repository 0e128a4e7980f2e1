# Developer aid (GPU box): the secondary stage sets of bench.py
for extra in "--group" "--ocr" "--group --ocr"; do
echo -n "bench.py $extra: "; python bench.py --no-cpu-baseline --no-latency --no-host-frames --steps 20 $extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
