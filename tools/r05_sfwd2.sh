#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $1 python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 120 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"; }
for i in 1 2; do
run CDRL_STEM_FWD_BAND=0
run "CDRL_STEM_FWD_NP=4 CDRL_STEM_FWD_WGS=768"
run "CDRL_STEM_FWD_NP=2 CDRL_STEM_FWD_WGS=512"
run "CDRL_STEM_FWD_NP=2 CDRL_STEM_FWD_WGS=1024"
run "CDRL_STEM_FWD_NP=3 CDRL_STEM_FWD_WGS=512"
run "CDRL_STEM_FWD_NP=3 CDRL_STEM_FWD_WGS=1024"
run "CDRL_STEM_FWD_NP=2 CDRL_STEM_FWD_WGS=512 CDRL_STEM_FWD_R=4"
run "CDRL_STEM_FWD_NP=3 CDRL_STEM_FWD_WGS=1024 CDRL_STEM_FWD_R=4"
done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "CDRL_STEM_FWD_NP=2 CDRL_STEM_FWD_WGS=512" "CDRL_STEM_FWD_NP=3 CDRL_STEM_FWD_WGS=512" "CDRL_STEM_FWD_NP=3 CDRL_STEM_FWD_WGS=1024 CDRL_STEM_FWD_R=4" "CDRL_STEM_FWD_NP=2 CDRL_STEM_FWD_WGS=1024 CDRL_STEM_FWD_R=4"; do
  export $cfg
  o=gpurun_out/tl_sf; rm -rf $o; mkdir -p $o
  timeout 300 rocprofv3 --kernel-trace -d $o -o trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $o/run.log 2>&1
  echo "$cfg: $(python3 tools/rocpd_timeline.py $o/trace_results.db 3600 2>/dev/null | grep stem_fwd_band | awk -F'\t' '{a+=$2;n++} END{print a/n, "us"}')"
  unset CDRL_STEM_FWD_NP CDRL_STEM_FWD_WGS CDRL_STEM_FWD_R
done
