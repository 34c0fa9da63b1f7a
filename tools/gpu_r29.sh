export TMPDIR=/tmp
mkdir -p gpurun_out/r29
timeout 900 python -m pytest tests/test_price_parse.py tests/test_brotli.py tests/test_flzma2.py -m gpu -x -q > gpurun_out/r29/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r29/pytest.log; tail -4 gpurun_out/r29/pytest.log
for lv in 9 12; do for f in 0 1; do echo "== zstd L$lv far $f"; GC_FAR_PASS=$f python tools/gpu_ratio.py --codecs zstd --levels $lv --corpora text-zipf,silesia-like --bytes 33554432 2>>gpurun_out/r29/err.log; done; done > gpurun_out/r29/zstd_far.log; cat gpurun_out/r29/zstd_far.log
bash tools/gpu_exp.sh r29 zstd "GC_FAR_PASS=0" "GC_FAR_PASS=1" 2>&1 | tail -4
