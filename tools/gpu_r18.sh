export TMPDIR=/tmp
mkdir -p gpurun_out/r18
timeout 600 python -m pytest tests/test_flzma2.py tests/test_price_parse.py -m gpu -x -q > gpurun_out/r18/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r18/pytest.log; tail -4 gpurun_out/r18/pytest.log
bash tools/gpu_exp.sh r18 flzma2 "GC_FAR_PASS=0" "GC_FAR_PASS=1" "GC_FAR_PASS=1 GC_SEARCH_DEPTH=0" "GC_FAR_PASS=1 GC_SEARCH_DEPTH=2" "GC_FAR_PASS=1 GC_SEG_LOG=16"
python tools/gpu_ratio.py --codecs flzma2 > gpurun_out/r18/ratio_far1.jsonl 2>gpurun_out/r18/ratio.err; cat gpurun_out/r18/ratio_far1.jsonl
GC_SEARCH_DEPTH=0 python tools/gpu_ratio.py --codecs flzma2 > gpurun_out/r18/ratio_far1_d0.jsonl 2>>gpurun_out/r18/ratio.err; cat gpurun_out/r18/ratio_far1_d0.jsonl
