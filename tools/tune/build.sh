#!/bin/bash
# builds one harness executable per (PW, DEPTH) variant into tools/tune/bin/ (cross-compiled here, run on the GPU box)
cd "$(dirname "$0")"; mkdir -p bin
for v in "4 2 100" "4 3 100"; do set -- $v
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -I../../cugraph-gnn_amd/csrc -DWG_MFMA_PW=$1 -DWG_MFMA_DEPTH=$2 -DWG_HARNESS_KSC=$3 \
    -x hip sage_mfma_harness.cpp -o bin/harness_pw$1_d$2_k$3 -L../../cugraph-gnn_amd/lib -lwholegraph_amd -Wl,-rpath,'$ORIGIN/../../../cugraph-gnn_amd/lib' \
    -Rpass-analysis=kernel-resource-usage 2> bin/res_pw$1_d$2_k$3.txt &
done; wait
grep -h -A8 "Function Name.*sage_layer_mfma" bin/res_*.txt | grep -E "VGPRs:|Spill" ; ls -la bin
