set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t37.log 2>&1; echo "full gpu suite rc=$?"; tail -n 5 gpurun_out/r3_t37.log
python -c "import __graft_entry__ as g; g.smoke()"
bash tools/profile_bench.sh r3c2b
