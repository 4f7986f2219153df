#!/bin/bash
# Seventh GPU call of round 4: same-box A/B of the attention stage-counter edit (old kernel in a side library), a few more decode knobs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/seventh
mkdir -p $O
cd $R
for i in 1 2; do
  CBX_LIB_PATH=$R/chatterbox_amd/build/libcbx_hip_oldattn.so python scripts/attn_planes_time.py 2>&1 | grep flash_attn | tee -a $O/attn_ab.log
  python scripts/attn_planes_time.py 2>&1 | grep flash_attn | tee -a $O/attn_ab.log
done
CBX_AB_SHORT=1 timeout 200 python scripts/decode_ab.py $O/decode_ab_short.json 2>&1 | grep -v amdgpu.ids | tee $O/decode_ab_short.log
