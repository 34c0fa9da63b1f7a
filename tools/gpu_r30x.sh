export TMPDIR=/tmp
mkdir -p gpurun_out/r30x
{ for lv in 1 3; do for f in 0 1; do echo "== flzma2 L$lv far $f"; GC_FAR_PASS=$f python tools/gpu_ratio.py --codecs flzma2 --levels $lv --corpora text-zipf,silesia-like --bytes 33554432; done; done
for lv in 1 4; do for f in 0 1; do echo "== brotli q$lv far $f"; GC_FAR_PASS=$f python tools/gpu_ratio.py --codecs brotli --levels $lv --corpora web-text,silesia-like --bytes 33554432; done; done
echo "== zstd L7 default"; python tools/gpu_ratio.py --codecs zstd --levels 7 --corpora text-zipf,silesia-like --bytes 33554432; } > gpurun_out/r30x/log.txt 2> gpurun_out/r30x/err.log
cat gpurun_out/r30x/log.txt
