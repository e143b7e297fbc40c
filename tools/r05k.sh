set -x
mkdir -p gpurun_out/r05k
cd /root/repo
timeout 2400 python -m pytest tests/test_solver.py -q -m gpu > gpurun_out/r05k/test_solver.log 2>&1; tail -30 gpurun_out/r05k/test_solver.log
python tools/run_cmax.py --config_file tests/golden/config_hot_plate1.json --height 260 --width 346 > gpurun_out/r05k/run_cmax_ref_346x260.json 2> gpurun_out/r05k/err1.txt
python tools/run_cmax.py --config_file tests/golden/config_hot_plate1.json > gpurun_out/r05k/run_cmax_ref_720x1280.json 2> gpurun_out/r05k/err2.txt
python tools/run_cmax.py --config_file configs/cmax_hot_plate1.yaml --n-iter 600 > gpurun_out/r05k/run_cmax_own_600.json 2> gpurun_out/r05k/err3.txt
cat gpurun_out/r05k/*.json
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_solver.py > gpurun_out/r05k/tests_rest.log 2>&1; tail -5 gpurun_out/r05k/tests_rest.log
