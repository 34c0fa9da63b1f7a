#!/bin/bash
# One GPU-box visit for ONE codec: its parity tests, bench line, rocprofv3 kernel stats, sizes next to the reference.
# usage: tools/gpu_codec.sh <tag> <zstd|flzma2|brotli> [levels for the ratio run]
TAG=${1:-c}; C=${2:-flzma2}; LV=$3; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
case $C in zstd) T="tests/test_gpu_parity.py";; flzma2) T="tests/test_flzma2.py";; brotli) T="tests/test_brotli.py";; esac
timeout 900 python -m pytest $T tests/test_plugin.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
EXTRA=""; [ $C == brotli ] && EXTRA="--bytes 500000000"
timeout 900 python bench.py --codec $C $EXTRA > $OUT/bench_$C.json 2> $OUT/bench_$C.err; cat $OUT/bench_$C.json; tail -2 $OUT/bench_$C.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$C -- python bench.py --codec $C $EXTRA --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_prof_$C.json 2> $OUT/prof_$C.err
DB=$(find $OUT/prof_$C -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_stats_$C.md && cat $OUT/kernel_stats_$C.md
rm -rf $OUT/prof_$C
LVA=""; [ -n "$LV" ] && LVA="--levels $LV"
python tools/gpu_ratio.py --codecs $C $LVA > $OUT/ratio.jsonl 2> $OUT/ratio.err; cat $OUT/ratio.jsonl
