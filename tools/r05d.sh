set -x
mkdir -p gpurun_out/r05d
cd /root/repo
timeout 1500 python -m pytest tests/test_solver.py -x -q -m gpu > gpurun_out/r05d/test_solver.log 2>&1; tail -25 gpurun_out/r05d/test_solver.log
python tools/run_cmax.py --config_file tests/golden/config_hot_plate1.json --height 260 --width 346 > gpurun_out/r05d/run_cmax_ref_346x260.json 2> gpurun_out/r05d/err1.txt
python tools/run_cmax.py --config_file configs/cmax_hot_plate1.yaml --n-iter 600 > gpurun_out/r05d/run_cmax_own_600.json 2> gpurun_out/r05d/err3.txt
cat gpurun_out/r05d/*.json; tail -n 3 gpurun_out/r05d/err*.txt
