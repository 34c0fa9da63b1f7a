export TMPDIR=/tmp
mkdir -p gpurun_out/r19
timeout 600 python -m pytest tests/test_flzma2.py tests/test_price_parse.py -m gpu -x -q > gpurun_out/r19/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r19/pytest.log; tail -4 gpurun_out/r19/pytest.log
bash tools/gpu_exp.sh r19 flzma2 "GC_SHORT_PASS=1" "GC_SHORT_PASS=0" "GC_SEARCH_DEPTH=2" "GC_SEARCH_DEPTH=0" "GC_SEG_LOG=16 GC_SEARCH_DEPTH=2"
python tools/gpu_ratio.py --codecs flzma2 > gpurun_out/r19/ratio_fl2.jsonl 2>gpurun_out/r19/ratio.err; cat gpurun_out/r19/ratio_fl2.jsonl
GC_SEARCH_DEPTH=2 python tools/gpu_ratio.py --codecs flzma2 > gpurun_out/r19/ratio_fl2_d2.jsonl 2>>gpurun_out/r19/ratio.err; cat gpurun_out/r19/ratio_fl2_d2.jsonl
python tools/gpu_ratio.py --codecs zstd --levels 19 --bytes 33554432 --corpora text-zipf,lz-7zip > gpurun_out/r19/ratio_zstd19.jsonl 2>>gpurun_out/r19/ratio.err; cat gpurun_out/r19/ratio_zstd19.jsonl
