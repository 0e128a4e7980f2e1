# Developer aid (GPU box): the PCIe-inclusive leg with and without SDMA copies
for sd in 1 0; do
echo -n "HSA_ENABLE_SDMA=$sd: "; HSA_ENABLE_SDMA=$sd python bench.py --no-cpu-baseline --no-latency --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['pcie_inclusive']['value'], d['pcie_inclusive']['ms_per_step'])"
done
