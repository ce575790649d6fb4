# dev: loop the GPU suite (or FILES) until something fails, then print the failure (run on the GPU box)
R=$GRAFT_REPO_ROOT
for i in $(seq 1 ${N:-5}); do
  timeout -k 10 600 python -m pytest ${FILES:-$R/tests} -m gpu -x -q > $R/gpurun_out/hunt.log 2>&1
  if grep -q " failed\| error" $R/gpurun_out/hunt.log; then
    echo "failed at run $i"
    grep -n "^E \|Error\|^tests/.*py:[0-9]*:\|^FAILED" $R/gpurun_out/hunt.log | head -20 | cut -c1-220
    exit 0
  fi
done
echo "no failure in ${N:-5} runs"
