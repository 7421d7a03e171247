for i in 1 2 3 4 5 6; do python bench.py --no-cpu --no-single --no-sweep > gpurun_out/r04_c$i.json 2>> gpurun_out/r04_c.err; done; echo done
