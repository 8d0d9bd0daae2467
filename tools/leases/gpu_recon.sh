cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "generic_pipeline or config4 or different_state or zero_is" 2>&1 | tail -2
python tools/time_passes.py 2000000 200 16 2>&1 | grep -v amdgpu | head -4
PPCA_GENERIC_RECON=naive python tools/time_passes.py 2000000 200 16 2>&1 | grep -v amdgpu | sed -n 2,4p
python tools/time_passes.py 200000 1024 64 2>&1 | grep -v amdgpu | head -4
