cd $GRAFT_REPO_ROOT
python tools/tune_conv.py --batch 256 --layers L1o,L1 --cfg -1 2>&1 | grep -v amdgpu | grep "heuristic"
DLWP_WINO_CIN4=1 python tools/tune_conv.py --batch 256 --layers L1o,L1 --cfg -1 2>&1 | grep -v amdgpu | grep "heuristic"
python tools/tune_conv.py --batch 64 --layers L1 --cfg -1 2>&1 | grep -v amdgpu | grep "heuristic"
DLWP_WINO_CIN4=1 python tools/tune_conv.py --batch 64 --layers L1 --cfg -1 2>&1 | grep -v amdgpu | grep "heuristic"
