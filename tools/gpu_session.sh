#!/bin/bash
# One GPU visit, assembled from steps (replaces the per-run scripts of rounds 1-4).   usage: tools/gpu_session.sh <tag> <step> [<step> ...]
# Outputs under gpurun_out/<tag>/; every step is bounded by its own timeout.  Steps:
#   smoke                                   __graft_entry__.smoke()
#   tests[:<pytest args>]                   python -m pytest -m gpu -q <args>      (default: the whole suite; "," separates args)
#   bench[:<bench.py args>]                 one bench line -> bench<k>.json        (default: the metric)
#   shard[:<Ns>]                            bench.py --shard-of <Ns>               (default 1,2,4,8; both legs of the metric)
#   ratio:<codec>:<levels>:<corpora>:<MiB>[:<ENV=V,...>]   tools/gpu_ratio.py (sizes against the reference, decode check); env => hooks library
#   prof[:<bench.py args>]                  rocprofv3 --kernel-trace --stats of the bench command -> kernel_stats<k>.md
#   pmc[:<bench.py args>]                   FETCH_SIZE and WRITE_SIZE passes (separate runs, kernel trace only) -> pmc<k>.md
#   py:<script>[:<args>]                    python tools/<script> <args>
TAG=${1:-session}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
k=0
for step in "$@"; do
  k=$((k+1)); kind=${step%%:*}; rest=""; [ "$step" != "$kind" ] && rest=${step#*:}
  args=$(echo "$rest" | tr ',' ' ')
  echo "== [$k] $step"
  case $kind in
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log ;;
    tests) timeout 1500 python -m pytest ${args:-tests} -m gpu -q > $OUT/tests$k.log 2>&1; tail -5 $OUT/tests$k.log ;;
    bench) timeout 1200 python bench.py $args > $OUT/bench$k.json 2> $OUT/bench$k.err; tail -2 $OUT/bench$k.err
           python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench$k.json').read().strip().splitlines()[-1])
    print(d['metric'][:60], d['value'], 'MB/s', d['ms_per_step'], 'ms', 'ratio', d.get('ratio'), (d.get('ratio_vs_ref') or {}).get('ours_over_ref'), d['roofline']['kernel_ms'])
    f = d.get('flzma2_l5_silesia')
    if f: print('flzma2', f['value'], f['ms_per_step'], (f.get('ratio_vs_ref') or {}).get('ours_over_ref'), f['roofline']['kernel_ms'])
    if d.get('real_data'): print('real', json.dumps(d['real_data']['corpora'])[:1500])
    if d.get('gpu_decode'): print('decode', d['gpu_decode']['value'], d['gpu_decode'].get('kernels_ms'))
except Exception as e: print('bench line unreadable', e)
PY
           ;;
    shard) timeout 900 python bench.py --steps 3 --warmup 1 --shard-of ${rest:-1,2,4,8} > $OUT/shard$k.json 2> $OUT/shard$k.err; tail -2 $OUT/shard$k.err
           python - <<PY
import json
try:
    d = json.loads(open('$OUT/shard$k.json').read().strip().splitlines()[-1])
    for leg in d['legs']:
        for r in leg['rows']: print(leg['codec'], leg['level'], 'N', r['n_gpus'], 'slowest', r['ms_slowest_rank'], 'ms', r['predicted_MBps'], 'MB/s', 'x', r['speedup_vs_first'], 'eff', r['efficiency'])
except Exception as e: print('unreadable', e)
PY
           ;;
    ratio) IFS=':' read -r codec levels corpora mib envs <<< "$rest"
           lib=""; [ -n "$envs" ] && lib="--lib 7-zip-zstd_amd/csrc/libgpucodec_hooks.so"
           env $(echo "$envs" | tr ',' ' ') timeout 900 python tools/gpu_ratio.py $lib --bytes $(( ${mib:-64} * 1024 * 1024 )) --codecs $codec --levels $levels --corpora $corpora > $OUT/ratio$k.jsonl 2> $OUT/ratio$k.err
           cut -c1-200 $OUT/ratio$k.jsonl; tail -2 $OUT/ratio$k.err ;;
    prof)  R=$PWD; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof$k -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-check $args > $R/$OUT/bench_prof$k.json 2> $R/$OUT/prof$k.err)
           DB=$(find $OUT/prof$k -name '*.db' | head -1)
           [ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_stats$k.md && head -45 $OUT/kernel_stats$k.md
           rm -rf $OUT/prof$k ;;
    pmc)   rm -f $OUT/pmc$k.md
           for ctr in FETCH_SIZE WRITE_SIZE; do
             timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/pmc_$ctr -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-check $args > $OUT/pmc_$ctr.json 2> $OUT/pmc_$ctr.err
             DB=$(find $OUT/pmc_$ctr -name '*.db' | head -1)
             if [ -n "$DB" ]; then echo "## $ctr" >> $OUT/pmc$k.md; python tools/rocpd_pmc.py $DB >> $OUT/pmc$k.md; echo >> $OUT/pmc$k.md; else echo "## $ctr: no result" >> $OUT/pmc$k.md; tail -3 $OUT/pmc_$ctr.err >> $OUT/pmc$k.md; fi
             rm -rf $OUT/pmc_$ctr
           done
           head -60 $OUT/pmc$k.md ;;
    py)    s=${rest%%:*}; a=""; [ "$rest" != "$s" ] && { a="${rest#*:}"; case "$a" in *\;*) a=$(echo "$a" | tr ';' ' ');; *) a=$(echo "$a" | tr ',' ' ');; esac; }      # (";" separates the arguments when they hold commas themselves)
           timeout 900 python tools/$s $a > $OUT/py$k.log 2>&1; tail -40 $OUT/py$k.log ;;
    *) echo "unknown step $step" ;;
  esac
done
