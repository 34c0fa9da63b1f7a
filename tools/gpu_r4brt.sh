#!/bin/bash
# round 4: brotli GPU tests after quality 6 took one match link (W5b), then the C5-share bench line
TAG=${1:-r4brt}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 170 python -m pytest tests/test_brotli.py tests/test_gpu_real_data.py tests/test_gpu_ratio_bars.py tests/test_bare_streams.py tests/test_emu_pipeline.py -m gpu -q -k "brotli or bare or gpu_edge or gpu_corpora or emulator_bytes or web_text" > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
timeout 60 python bench.py --codec brotli --no-cpu-baseline --steps 5 > $OUT/bench_br.log 2>$OUT/bench_br.err; tail -c 1500 $OUT/bench_br.log
