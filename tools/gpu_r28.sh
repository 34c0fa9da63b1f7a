export TMPDIR=/tmp
mkdir -p gpurun_out/r28
echo "== brotli q6 greedy + far" > gpurun_out/r28/log.txt
GC_PRICE_PARSE=0 GC_FAR_PASS=1 python tools/gpu_ratio.py --codecs brotli >> gpurun_out/r28/log.txt 2>gpurun_out/r28/err.log
echo "== brotli q6 price + far" >> gpurun_out/r28/log.txt
GC_FAR_PASS=1 python tools/gpu_ratio.py --codecs brotli >> gpurun_out/r28/log.txt 2>>gpurun_out/r28/err.log
bash tools/gpu_exp.sh r28 brotli "GC_PRICE_PARSE=0 GC_FAR_PASS=1" "GC_FAR_PASS=1" "GC_PRICE_PARSE=0" >> gpurun_out/r28/log.txt 2>&1
cat gpurun_out/r28/log.txt
