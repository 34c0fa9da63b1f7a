#!/bin/bash
# round 4, last GPU action: HBM traffic counters (FETCH_SIZE, WRITE_SIZE in separate passes) of the two legs of the metric at the round's last code
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" bash tools/gpu_pmc.sh r4pmc_zstd --codec zstd > /dev/null 2>&1; tail -45 gpurun_out/r4pmc_zstd/pmc.md | cut -c1-120
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" bash tools/gpu_pmc.sh r4pmc_fl2 --codec flzma2 > /dev/null 2>&1; grep -c gc_ gpurun_out/r4pmc_fl2/pmc.md
