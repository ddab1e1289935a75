#!/bin/bash
# A/B of two builds of the library on the same box: libbpltv_prev.so (BPLTV_LIB_PATH) vs libbpltv.so, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  echo "--- prev"; BPLTV_LIB_PATH=$R/bpldenoising_amd/libbpltv_prev.so python3 $R/tools/gpu_bcr2.py | tail -1
  echo "--- new"; python3 $R/tools/gpu_bcr2.py | tail -1
done
