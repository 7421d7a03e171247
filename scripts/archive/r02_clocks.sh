#!/bin/bash
# Rate of the same transform, process after process, beside the clocks / power rocm-smi reports.
mkdir -p gpurun_out/r02/clocks
for i in 1 2 3 4 5 6; do
  ( while true; do echo "t $(date +%s.%N)"; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power|junction|memory\)" ; sleep 0.4; done ) > gpurun_out/r02/clocks/smi_$i.txt &
  SMI=$!
  timeout -k 10 100 python scripts/rate_vs_clocks.py 2.5 > gpurun_out/r02/clocks/rate_$i.txt 2>&1
  kill $SMI
  echo "== process $i"; grep Gsamples gpurun_out/r02/clocks/rate_$i.txt | awk '{print $4}' | tr '\n' ' '; echo
  grep -E "sclk|mclk|fclk|Power|junction" gpurun_out/r02/clocks/smi_$i.txt | tail -6
done
