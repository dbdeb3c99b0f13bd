# usage: bash tools/gpu_dispatch_probe.sh  -- tools/dispatch_probe under its backgrounds (built on the CPU box: the binary travels)
cd $GRAFT_REPO_ROOT
P="timeout -k 5 60 tools/dispatch_probe"
$P none 0 16 2000 && $P burst 100 16 2000 1 && $P burst 100 16 2000 2 && $P burst 100 16 2000 3 && $P burst 40 16 2000 3 && $P burst 20 16 2000 3 && $P burst 100 16 300 3 96 64
