export TMPDIR=/tmp
mkdir -p gpurun_out/r22
for c in silesia-like text-zipf; do python tools/gpu_profile.py --codec flzma2 --bytes 211900000 --corpus $c >> gpurun_out/r22/l2_phase.jsonl 2>> gpurun_out/r22/err.log; done
GC_SEG_LOG=17 python tools/gpu_profile.py --codec flzma2 --bytes 211900000 --corpus text-zipf >> gpurun_out/r22/l2_phase.jsonl 2>> gpurun_out/r22/err.log
cat gpurun_out/r22/l2_phase.jsonl; tail -3 gpurun_out/r22/err.log
