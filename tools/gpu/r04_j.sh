#!/bin/bash
# round 4, GPU call J: HBM traffic (separate --pmc FETCH_SIZE pass; gfx950 reports 1/2 of wide streaming reads) of the round's new
# decode kernels: gemv_wg_kernel (incl. the split form at 32 rows), fp24 and e4m3 decode attention
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 170 rocprofv3 --pmc FETCH_SIZE -d $O/r04_j_pmc -o pmc -- python $ROOT/tools/kbench.py dattn_split dattn_kv8 gemv_wg > $O/r04_j_pmc_kbench.txt 2> $O/r04_j_pmc.err
DB=$(find $O/r04_j_pmc -name "*.db" | head -1)
python $ROOT/tools/pmc_summary.py "$DB" > $O/r04_j_pmc_summary.txt 2>> $O/r04_j_pmc.err
rm -rf $O/r04_j_pmc
grep -E "gemv_wg_kernel|attention_decode" $O/r04_j_pmc_summary.txt | head -30
