#!/bin/bash
# usage (GPU box): tools/ab_runtime_env.sh  -> gpurun_out/ab_runtime_env.txt
# A/B of HIP / ROCr runtime switches that could move the dependent-launch floor (4.8 us per tiny kernel): the default bench workload
# and tools/launch_cost.py under each.  The bench itself is unchanged; these are process environment variables read at HIP initialisation.
cd $GRAFT_REPO_ROOT
out=gpurun_out/ab_runtime_env.txt
: > $out
one() {
  echo "== $*" >> $out
  env "$@" timeout 300 python tools/launch_cost.py 2>/dev/null | grep "gemm_nn tiny" >> $out
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-rooflines --no-secondary --steps 60 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" >> $out 2>&1
}
one A=1
one HIP_FORCE_DEV_KERNARG=1
one HIP_FORCE_DEV_KERNARG=0
one HSA_ALLOCATE_QUEUE_DEV_MEM=1
one HSA_ALLOCATE_QUEUE_DEV_MEM=1 HIP_FORCE_DEV_KERNARG=1
one AMD_OPT_FLUSH=0
one AMD_OPT_FLUSH=1
one ROC_USE_FGS_KERNARG=0
one DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1
one ROC_SYSTEM_SCOPE_SIGNAL=0
one HSA_ENABLE_INTERRUPT=0
one A=1
cat $out
