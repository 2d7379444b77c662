mkdir -p gpurun_out/r2c
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ps or pipeline" > gpurun_out/r2c/pytest_ps.log 2>&1 || { tail -30 gpurun_out/r2c/pytest_ps.log; exit 1; }
tail -2 gpurun_out/r2c/pytest_ps.log
timeout -k 10 500 python tools/sweep_ps.py --lds 81920 --threads 1024 --remap 1 --iters 40 --persist 0,1,0,1,0,1 > gpurun_out/r2c/sweep_ps_persist.log 2>&1
grep "ab=" gpurun_out/r2c/sweep_ps_persist.log
timeout -k 10 500 python tools/sweep_ps.py --n 2000000 --s 500 --lds 81920 --threads 1024 --remap 1 --iters 10 --persist 0,1,0,1 > gpurun_out/r2c/sweep_ps_persist_2m.log 2>&1
grep "ab=" gpurun_out/r2c/sweep_ps_persist_2m.log
