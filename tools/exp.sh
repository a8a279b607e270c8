#!/bin/bash
# scratch experiment: the new explicit tests
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ctc.py tests/test_gpu_frontend.py tests/test_gpu_train.py -x -q -m gpu -k "prepare or one_launch or staged" > gpurun_out/exp_pytest.log 2>&1
echo "pytest exit $?"; tail -15 gpurun_out/exp_pytest.log
