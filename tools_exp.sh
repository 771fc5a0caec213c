cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_aggregators.py -m gpu -q -x --timeout 600 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --verbose --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | head -12
