set -x
mkdir -p gpurun_out/r05b
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blur" > gpurun_out/r05b/test_blur.log 2>&1; tail -15 gpurun_out/r05b/test_blur.log
timeout 900 python -m pytest tests/test_solver.py -x -q -m gpu > gpurun_out/r05b/test_solver.log 2>&1; tail -15 gpurun_out/r05b/test_solver.log
python tools/run_cmax.py --config_file tests/golden/config_hot_plate1.json --height 260 --width 346 > gpurun_out/r05b/run_cmax_ref_346x260.json 2> gpurun_out/r05b/err1.txt
python tools/run_cmax.py --config_file tests/golden/config_hot_plate1.json > gpurun_out/r05b/run_cmax_ref_720x1280.json 2> gpurun_out/r05b/err2.txt
python tools/run_cmax.py --config_file configs/cmax_hot_plate1.yaml --n-iter 600 > gpurun_out/r05b/run_cmax_own_600.json 2> gpurun_out/r05b/err3.txt
cat gpurun_out/r05b/*.json; tail -3 gpurun_out/r05b/err*.txt
