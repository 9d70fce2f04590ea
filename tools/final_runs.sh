for s in 16 64 128; do python bench.py --seeds-per-gpu $s --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final_full_$s.log 2>&1; tail -1 gpurun_out/final_full_$s.log | cut -c1-100; done
for s in 64 256; do python bench.py --mode gn --seeds-per-gpu $s --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final_gn_$s.log 2>&1; tail -1 gpurun_out/final_gn_$s.log | cut -c1-100; done
