#!/bin/bash
# round 4, last GPU visit: the plugin after the read-ahead change under the reference's own 7z (tests/test_real_host.py, tests/test_plugin*.py, -m gpu), then the
# product-level wall times of `7z a` / `7z x` on 1 GB (tools/gpu_7z_product_rate.py)
TAG=${1:-r4plugin}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 280 python -m pytest tests/test_real_host.py tests/test_plugin.py tests/test_plugin_filters.py tests/test_abi.py -m gpu -q > $OUT/tests_plugin.log 2>&1; tail -4 $OUT/tests_plugin.log
timeout 200 python tools/gpu_7z_product_rate.py 1e9 > $OUT/product_7z.log 2>&1; tail -12 $OUT/product_7z.log
