#!/bin/bash
# round 6, call j: rows-per-wave sweep of the batch-1 row kernels (scripts/row_micro.py) at the Turbo and Nano widths
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_j
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_models_gpu.py -q -m gpu -k "sampling_params" 2>&1 | tail -2
timeout 400 python scripts/row_micro.py turbo > $O/row_micro_turbo.log 2>&1; cat $O/row_micro_turbo.log | grep -v amdgpu
timeout 400 python scripts/row_micro.py nano > $O/row_micro_nano.log 2>&1; cat $O/row_micro_nano.log | grep -v amdgpu
