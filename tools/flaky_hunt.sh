# dev: loop one test file until it fails, then print the failure (run on the GPU box)
R=$GRAFT_REPO_ROOT
for i in $(seq 1 ${N:-10}); do
  timeout -k 10 400 python -m pytest $R/tests/test_gpu_entropy.py -m gpu -x -q > $R/gpurun_out/hunt.log 2>&1
  if grep -q " failed" $R/gpurun_out/hunt.log; then
    echo "failed at run $i"
    grep -n "^E \|Error\|^tests/.*py:[0-9]*:" $R/gpurun_out/hunt.log | head -20 | cut -c1-220
    exit 0
  fi
done
echo "no failure in ${N:-10} runs"
