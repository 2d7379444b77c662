mkdir -p gpurun_out/r2c
timeout -k 10 500 python tools/sweep_ps.py --lds 81920,53248,40960 --threads 1024,768,512 --remap 1 --iters 30 --chunk 0,52,64 > gpurun_out/r2c/sweep_ps_chunks.log 2>&1
grep "ab=" gpurun_out/r2c/sweep_ps_chunks.log | sort -t' ' -k12 -n | head -40
