# Winograd-depth kernel A/B: shipped library vs ms-nets_amd/libx_*.so -- parity of the layer tests, layer loop, network bench.
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_aggregators.py -m gpu -q -k "winograd" 2>&1 | tail -1
bash tools/tools_ab_layers.sh s1_32_32 2>&1 | grep -E "round|ms"
for so in "" $(ls ms-nets_amd/libx_*.so); do
  echo "== network: ${so:-shipped}"
  for r in 1 2; do MSNET_HIP_LIB=${so:+$PWD/$so} python bench.py --steps 10 --warmup 3 --verbose --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "wd_f16s|kernels"; done
done
