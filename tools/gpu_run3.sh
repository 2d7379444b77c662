mkdir -p gpurun_out/r2c
for rep in 1 2; do
timeout -k 10 500 python tools/sweep_ps.py --lds 81920,40960,49152 --threads 1024,512,640 --remap 1 --iters 40 > gpurun_out/r2c/sweep_ps_small$rep.log 2>&1
grep "ab=" gpurun_out/r2c/sweep_ps_small$rep.log | awk '{print $5,$6,$7,$8,$11,$12}'
echo ---
done
timeout -k 10 500 python tools/sweep_ps.py --n 2000000 --s 500 --lds 81920,40960,49152 --threads 1024,512,640 --remap 1 --iters 10 > gpurun_out/r2c/sweep_ps_small_2m.log 2>&1
grep "ab=" gpurun_out/r2c/sweep_ps_small_2m.log | awk '{print $5,$6,$7,$8,$11,$12}'
