#!/bin/bash
# round 6, call U: the ViT's K = 1024 linears on the existing GEMM kernels (8-phase 256^2 = default, one-barrier 256^2, DMA 128^2)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in 1 3 2 0; do
  VC_GEMM_VARIANT=$v timeout 300 python tools/kbench.py gemm 2>&1 | grep "vit\|adapter" | sed "s/^/variant $v: /"
done | tee gpurun_out/r06_u_kbench_gemm_vit_variants.txt
