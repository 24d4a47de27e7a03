#!/bin/bash
# round 4, GPU call M: the K = 1024 ViT linears (the library is 25-33 % ahead there): 256x256 8-phase tile vs the 128x128 LDS-DMA tile
cd "$GRAFT_REPO_ROOT" || exit 1
for v in 1 3 2; do
  echo "VC_GEMM_VARIANT=$v"; VC_GEMM_VARIANT=$v timeout 100 python tools/kbench.py gemm 2>&1 | grep -E "vit|adapter"
done > gpurun_out/r04_m_kbench_gemm_vit_tiles.txt; cat gpurun_out/r04_m_kbench_gemm_vit_tiles.txt
