#!/bin/bash
# round 4, call C: the synthetic-producer ablation of the fat GEMM (item 1d's measurement), the 256x256 configuration (family
# table; the runtime's graph knobs), then the whole -m gpu suite
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out/r4c
bash scripts/gpu_abl_producer.sh > gpurun_out/r4c/producer.txt 2>&1 || { tail -20 gpurun_out/r4c/producer.txt; exit 1; }
cat gpurun_out/r4c/producer.txt
bash scripts/gpu_c2fam.sh || exit 1
bash scripts/gpu_c2.sh "A=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_HIP_GRAPH_DOT_PRINT=0 HIP_GRAPH_PACKET_CAPTURE=1" "BGAMD_NO_WGRAD_STREAM=1 BGAMD_NO_SIDE_STREAM=1 BGAMD_NO_G_PREFETCH=1" || exit 1
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4c/t_all.log 2>&1; rc=$?
tail -5 gpurun_out/r4c/t_all.log
exit $rc
