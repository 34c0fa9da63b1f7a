export TMPDIR=/tmp
mkdir -p gpurun_out/r31
timeout 900 python -m pytest tests/test_brotli.py tests/test_price_parse.py -m gpu -x -q > gpurun_out/r31/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r31/pytest.log; tail -3 gpurun_out/r31/pytest.log
python bench.py --codec brotli --bytes 500000000 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'MBps': d['value'], 'ms': d['ms_per_step'], 'comp': d['compressed_bytes'], 'kernel_ms': d['roofline']['kernel_ms']}))" | tee gpurun_out/r31/bench.log
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" bash tools/gpu_pmc.sh r31/pmc_brotli --codec brotli --bytes 500000000 > /dev/null 2>&1
cat $(find gpurun_out/r31/pmc_brotli -name '*.md' | head -1) | grep -i "block\|kernel" | head -5
