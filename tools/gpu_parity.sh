set -e
cd $GRAFT_REPO_ROOT
make -f oracle/Makefile oracle/librsi_oracle.so >/dev/null
rocminfo | grep -m2 gfx || true
timeout -k 10 900 python -m pytest tests/test_hot_parity.py -m gpu -x -q 2>&1 | tail -40
