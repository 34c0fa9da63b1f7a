#!/bin/bash
# Assemble the judged profile files of one GPU visit (tools/gpu_full.sh <tag>) from gpurun_out/<tag>/ into profiles/.
# usage: tools/collect_profiles.sh <tag> <run-number>
TAG=$1; RUN=$2; IN=gpurun_out/$TAG
for c in zstd flzma2 brotli; do
  EXTRA=""; [ $c == brotli ] && EXTRA=" --bytes 500000000"
  O=profiles/r01_run${RUN}_${c}_kernel_stats.md
  { echo "# run $RUN ($c): rocprofv3 --kernel-trace --stats of \`python bench.py --codec $c$EXTRA --steps 5 --warmup 1 --no-cpu-baseline\` on one MI355X"; echo
    cat $IN/kernel_stats_$c.md; echo
    echo "## bench line of the same workload (20 steps, with the reference codec on the host cores)"; echo '```'; cat $IN/bench_$c.json; echo '```'; echo
    PMC=$(find $IN/pmc_$c -name '*.md' | head -1)
    if [ -n "$PMC" ]; then echo "## HBM traffic counters (separate rocprofv3 --pmc passes, KB per dispatch; FETCH_SIZE counts 64 B per 128 B request: see profiles/pmc_traffic.json)"; cat $PMC; fi
  } > $O
done
{ echo "# run $RUN: sizes of the three GPU codecs next to the reference codecs on the same bytes (64 MiB per corpus; zstd level 3, flzma2 level 5, brotli quality 6)"; echo '```'; cat $IN/ratio.jsonl; echo '```'; tail -3 $IN/pytest.log; } > profiles/r01_run${RUN}_ratio.md
python tools/pmc_to_json.py zstd=$(find $IN/pmc_zstd -name '*.md' | head -1) flzma2=$(find $IN/pmc_flzma2 -name '*.md' | head -1) brotli=$(find $IN/pmc_brotli -name '*.md' | head -1) > profiles/pmc_traffic.json
