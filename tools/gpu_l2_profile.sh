export TMPDIR=/tmp
# FLZMA2 parity tests + in-kernel phase profile of the model kernel (L2).  usage: tools/gpu_l2_profile.sh <tag>
T=${1:-l2prof}
mkdir -p gpurun_out/$T
timeout 600 python -m pytest tests/test_flzma2.py -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/$T/pytest.log; tail -3 gpurun_out/$T/pytest.log
for c in silesia-like text-zipf; do python tools/gpu_profile.py --codec flzma2 --bytes 211900000 --corpus $c 2>> gpurun_out/$T/err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms']
print(json.dumps({'corpus': d['corpus'], 'comp': d['compressed'], 'total': k['total'], 'model': k['model'], 'rc': k['rc'], 'phase': d['phase_cycles_per_block']}))" >> gpurun_out/$T/l2_phase.jsonl; done
cat gpurun_out/$T/l2_phase.jsonl
