export TMPDIR=/tmp
mkdir -p gpurun_out/r21
python tools/gpu_host_rate.py > gpurun_out/r21/host_rate.jsonl 2> gpurun_out/r21/host_rate.err; cat gpurun_out/r21/host_rate.jsonl
for s in 16 17; do GC_SEG_LOG=$s python tools/gpu_ratio.py --codecs flzma2 --corpora silesia-like,text-zipf > gpurun_out/r21/ratio_seg$s.jsonl 2>> gpurun_out/r21/ratio.err; cat gpurun_out/r21/ratio_seg$s.jsonl; done
python tools/gpu_ratio.py --codecs flzma2 --levels 1,3,7,9 --corpora silesia-like > gpurun_out/r21/ratio_levels.jsonl 2>> gpurun_out/r21/ratio.err; cat gpurun_out/r21/ratio_levels.jsonl
python tools/gpu_ratio.py --codecs zstd --levels 1,6,12,16 --corpora text-zipf --bytes 33554432 > gpurun_out/r21/ratio_zstd_levels.jsonl 2>> gpurun_out/r21/ratio.err; cat gpurun_out/r21/ratio_zstd_levels.jsonl
python tools/gpu_ratio.py --codecs brotli --levels 1,4,9 --corpora web-text --bytes 33554432 > gpurun_out/r21/ratio_brotli_levels.jsonl 2>> gpurun_out/r21/ratio.err; cat gpurun_out/r21/ratio_brotli_levels.jsonl
