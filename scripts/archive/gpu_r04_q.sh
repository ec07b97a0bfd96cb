#!/bin/bash
# r04q: feeders per GPU with the round-4 kernels (256 Mb, 12 steps), and 3100 Mb with 3 / 4 feeders
O=gpurun_out/${1:-r04q}; mkdir -p $O
for f in 2 3 4 5; do
  timeout 200 python bench.py --genome-mb 256 --no-extra-legs --steps 12 --warmup 4 --feeders $f --skip-cpu --skip-probe --skip-refwalk --skip-breakdown > $O/f$f.out 2> $O/f$f.err
  python -c "import json;d=json.loads(open('$O/f$f.out').readline());print('== feeders $f: %.0f reads/s, %.1f ms/step, launch %.1f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
