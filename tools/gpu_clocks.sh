# usage: bash tools/gpu_clocks.sh  -- the GPU's clocks (rocm-smi, sampled every 0.25 s) while the genome runs with one worker and with twelve
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
sample() { while [ -f /tmp/clk_on ]; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo; sleep 0.25; done; }
for v in "w1:workers=1,timing=0" "w12:workers=12,timing=0"; do
  touch /tmp/clk_on
  sample > gpurun_out/clocks_${v%%:*}.txt &
  SP=$!
  timeout -k 10 300 python tools/ab_bench.py --rounds 40 --variants "$v" 2>&1 | grep mean
  rm -f /tmp/clk_on
  wait $SP
  echo "--- ${v%%:*}: $(wc -l < gpurun_out/clocks_${v%%:*}.txt) samples"; sort gpurun_out/clocks_${v%%:*}.txt | uniq -c | sort -rn | head -8
done
