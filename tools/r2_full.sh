mkdir -p gpurun_out/r2full
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2full/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2full/pytest.log
tail -15 gpurun_out/r2full/pytest.log
timeout 600 python bench.py > gpurun_out/r2full/bench.json 2> gpurun_out/r2full/bench.err; cat gpurun_out/r2full/bench.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
