# usage: bash tools/gpu_dispatch_probe.sh  -- tools/dispatch_probe under its backgrounds (built on the CPU box: the binary travels)
cd $GRAFT_REPO_ROOT
P="timeout -k 5 60 tools/dispatch_probe"
$P none 0 16 2000 && $P write 1 16 2000 && $P write 4 16 2000 && $P write 8 16 2000 && $P write 4 1 2000 && $P write 4 16 300 96 64 && $P none 0 16 300 96 64
