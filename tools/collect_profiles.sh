#!/bin/bash
# Run on a GPU box (via gpurun) from the repo root: bench lines + rocprofv3 summaries for profiles/.
# Usage: tools/collect_profiles.sh r02
set -u
TAG=${1:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (the profiled runs keep ONE batch in flight: a kernel's average duration is then its isolated launch duration, the quantity roofline.avg_launch_ms reports)
B="python $ROOT/bench.py --steps 6 --warmup 2 --pipelines 1 --min-region-s 0 --no-ocr-legs --no-4k-leg --no-cpu-baseline --no-latency --no-host-frames --no-ties-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $B > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $B > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq1 -o p -- $B > $OUT/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA -d $OUT/pmc_sq2 -o p -- $B > $OUT/pmc_sq2.log 2>&1
# bench.py reads roofline.traffic from profiles/pmc_tile_tree.json: write it from the counter passes above before the bench lines are taken
python $ROOT/tools/make_profile_summary.py $TAG --pmc-json-only > $OUT/pmc_json.log 2>&1
# (round 5) the legs that have no headline of their own: config 3 (the OCR scorer), the reference's real call pattern (lines, then the scorer), config 5 (4K)
P1="--steps 6 --warmup 2 --repeats 1 --pipelines 1 --min-region-s 0 --no-cpu-baseline --no-latency --no-host-frames --no-ties-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_config3 -o s -- python $ROOT/bench.py --ocr $P1 > $OUT/stats_config3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_group_ocr -o s -- python $ROOT/bench.py --group --ocr $P1 > $OUT/stats_group_ocr.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_4k -o s -- python $ROOT/bench.py --size 4k $P1 > $OUT/stats_4k.log 2>&1
bash $ROOT/tools/dev_ocr_pmc.sh profiles_$TAG/pmc_config3 > /dev/null 2>&1
bash $ROOT/tools/dev_lat.sh profiles_$TAG/latency > /dev/null 2>&1
cd /tmp
python $ROOT/bench.py > $OUT/bench_${TAG}_pyr3x8_text.json 2> $OUT/bench_err.log
python $ROOT/bench.py --workload native6 > $OUT/bench_${TAG}_native6_text.json 2>> $OUT/bench_err.log
python $ROOT/bench.py --workload native6 --kind noise --no-cpu-baseline > $OUT/bench_${TAG}_native6_noise.json 2>> $OUT/bench_err.log
rm -f $OUT/*/*kernel_trace.csv      # large; the stats and counter files are what we keep
ls -R $OUT | head -40
