O=gpurun_out/${1:-r02f}; mkdir -p $O
timeout 240 python bench.py --workload paired --steps 2 --warmup 1 > $O/paired_help.json 2> $O/paired_help.err < /dev/null; tail -c 500 $O/paired_help.json
SNAPGPU_PAIRED_HELP_MIN=0 timeout 200 python bench.py --workload paired --steps 2 --warmup 1 --skip-cpu > $O/paired_nohelp.json 2> $O/paired_nohelp.err < /dev/null; tail -c 300 $O/paired_nohelp.json
