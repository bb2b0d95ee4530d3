set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t55.log 2>&1; echo "full gpu suite rc=$?"; tail -n 4 gpurun_out/r3_t55.log
python -c "import __graft_entry__ as g; g.smoke()"
