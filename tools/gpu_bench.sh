# usage: bash tools/gpu_bench.sh [bench args]   (run on the GPU box through gpurun)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
timeout -k 10 900 python bench.py "$@" 2> gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
