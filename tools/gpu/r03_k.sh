#!/bin/bash
# round 3, GPU call K: forms of the flash kernel's tile body, A/B as separately built libraries (vcoder_amd/lib/exp/):
#   whole = whole-tile phases (S | softmax | PV);  (default lib) = half-tile phases with packed f32 ops;  scalar = half-tile, scalar ops
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_k; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for form in whole pk scalar; do
  L=$GRAFT_REPO_ROOT/vcoder_amd/lib/exp/libvcoder_hip_$form.so; [ $form = pk ] && L=$GRAFT_REPO_ROOT/vcoder_amd/lib/libvcoder_hip.so
  for var in 0 2; do
    VCODER_HIP_LIB=$L VC_ATTN_SCHED=2 VC_ATTN_VARIANT=$var timeout 200 python tools/kbench.py attn 2>&1 | grep attention | sed "s/^/form=$form /" >> $O/kbench_attn_forms.txt
  done
done
done
cat $O/kbench_attn_forms.txt
