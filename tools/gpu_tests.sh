set -e
cd $GRAFT_REPO_ROOT
make -f oracle/Makefile oracle/librsi_oracle.so >/dev/null
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -40
