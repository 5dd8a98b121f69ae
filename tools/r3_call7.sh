#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
T=tests/test_gpu_model.py::test_reference_style_example_script_runs_end_to_end
timeout 300 python -m pytest $T -x -q > $O/a.log 2>&1; echo "default rc=$?"
DLWP_TRAIN_GRAPH=1 DLWP_TRAIN_FOLD=0 timeout 300 python -m pytest $T -x -q > $O/b.log 2>&1; echo "graph1 fold0 rc=$?"
DLWP_TRAIN_GRAPH=0 timeout 300 python -m pytest $T -x -q > $O/c.log 2>&1; echo "graph0 rc=$?"
AMD_LOG_LEVEL=0 HIP_LAUNCH_BLOCKING=1 timeout 300 python -m pytest $T -x -q > $O/d.log 2>&1; echo "default blocking rc=$?"
timeout 600 python -m pytest tests/test_gpu_bf16_octets.py -q > $O/oct.log 2>&1; echo "oct rc=$?"; grep -n "AssertionError\|assert \|config" $O/oct.log | head -10
