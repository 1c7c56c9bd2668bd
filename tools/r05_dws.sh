#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "dwconv_bn_fused" 2>&1 | tail -3
echo "== new"; bash tools/iso.sh tools/iso_dwf.py dws_new "$@" 2>&1 | grep -E "dws|dwf_bwd"
echo "== old"; CDRL_DWS=0 CDRL_DWS2=0 bash tools/iso.sh tools/iso_dwf.py dws_old "$@" 2>&1 | grep -E "dws|dwf_bwd"
