#!/bin/bash
# round 3, GPU call L: whole-tile flash body — as committed (whole), + VALU trims (default lib), + s_setprio around the MFMA clusters
# (prio), + deferred maximum with threshold 8 (defer).  LLM shape with VC_ATTN_SCHED=2, ViT shape with 0 and 3.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_l; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for form in whole trim prio defer; do
  L=$GRAFT_REPO_ROOT/vcoder_amd/lib/exp/libvcoder_hip_$form.so; [ $form = trim ] && L=$GRAFT_REPO_ROOT/vcoder_amd/lib/libvcoder_hip.so
  for sc in 2 0 3; do
    VCODER_HIP_LIB=$L VC_ATTN_SCHED=$sc timeout 200 python tools/kbench.py attn 2>&1 | grep attention | sed "s/^/form=$form sched=$sc /" >> $O/kbench_attn_forms.txt
  done
done
done
cat $O/kbench_attn_forms.txt
