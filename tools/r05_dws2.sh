#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "dwconv_bn_fused" 2>&1 | tail -3
for w in 256 512 1024 2048; do
echo "== CDRL_DWS_WGS=$w"; CDRL_DWS_WGS=$w bash tools/iso.sh tools/iso_dwf.py dws_w$w 2>&1 | grep -E "dws"
done
