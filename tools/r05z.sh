set -x
cd /root/repo
COMMIT=$1 bash tools/collect_round_profiles.sh r05z > gpurun_out/r05z_collect.log 2>&1; tail -3 gpurun_out/r05z_collect.log
mkdir -p gpurun_out/r05z
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r05z/tests.log 2>&1; tail -4 gpurun_out/r05z/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05z/smoke.txt 2>&1; tail -3 gpurun_out/r05z/smoke.txt
python tools/run_cmax.py --config_file tests/golden/config_hot_plate1.json --height 260 --width 346 > gpurun_out/r05z/run_cmax_ref_346x260.json 2> gpurun_out/r05z/err1.txt
python tools/run_cmax.py --config_file tests/golden/config_hot_plate1.json > gpurun_out/r05z/run_cmax_ref_720x1280.json 2> gpurun_out/r05z/err2.txt
python tools/run_cmax.py --config_file configs/cmax_hot_plate1.yaml --n-iter 600 > gpurun_out/r05z/run_cmax_own_600.json 2> gpurun_out/r05z/err3.txt
python tools/run_cmax.py --config_file tests/golden/config_hot_plate1.json --height 260 --width 346 --fractional > gpurun_out/r05z/run_cmax_ref_346x260_fractional.json 2> gpurun_out/r05z/err4.txt
python tools/run_cmax.py --config_file configs/cmax_hot_plate1.yaml --n-iter 600 --fractional > gpurun_out/r05z/run_cmax_own_600_fractional.json 2> gpurun_out/r05z/err5.txt
cat gpurun_out/r05z/run_cmax*.json | cut -c1-1200
