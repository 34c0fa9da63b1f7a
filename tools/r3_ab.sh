#!/bin/bash
# round 3: A/B of a test hook on the 1 GB zstd-L3 workload.  usage: tools/r3_ab.sh <tag> <ENVVAR> <values...>
TAG=$1; VAR=$2; shift 2; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for v in "$@"; do
  env $VAR=$v timeout 600 python tools/gpu_profile.py --bytes 1000000000 --reps 3 > $OUT/phase_${VAR}_$v.json 2> $OUT/phase_${VAR}_$v.err; cat $OUT/phase_${VAR}_$v.json
done
