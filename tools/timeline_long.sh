out=$GRAFT_REPO_ROOT/gpurun_out/tl_long
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace -d $out -o trace -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-rooflines "$@" > $out/run.log 2>&1
python3 tools/rocpd_timeline.py $out/trace_results.db 3600 > $out/timeline.tsv 2> $out/cols.txt
rm -f $out/trace_results.db
tail -1 $out/run.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step under the profiler', d['ms_per_step'], 'host enqueue', d.get('host_enqueue_ms_per_step'))"
n=$(python3 tools/timeline_step.py $out/timeline.tsv 1 | head -1 | cut -d" " -f4)
python3 tools/timeline_gaps.py $out/timeline.tsv $n
