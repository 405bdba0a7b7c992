cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06t
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06t/pytest_gpu.log 2>&1; echo "pytest -m gpu exit $?"; tail -4 gpurun_out/r06t/pytest_gpu.log
bash scripts/r06_build_in_process.sh r06t/build > gpurun_out/r06t/build_in_process.txt 2>&1; cat gpurun_out/r06t/build_in_process.txt
bash scripts/r06_three_frames_ab.sh r06t/three > gpurun_out/r06t/three_frames_ab.txt 2>&1; cat gpurun_out/r06t/three_frames_ab.txt
