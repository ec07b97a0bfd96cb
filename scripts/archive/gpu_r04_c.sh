#!/bin/bash
# r04c: A/B of the affine-gap row loop (v1 = rounds 1-3, v2 = round 4: masks on the scalar unit, lazy F as cheap rounds) and of the function
# form of LV / affine gap in the single-end kernels, on the 256 Mb genome, parity on in every run (scripts/ab_bench.py)
O=gpurun_out/${1:-r04c}; mkdir -p $O
run() { tag=$1; shift; t0=$SECONDS; ( timeout ${T:-300} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ); echo "wall=$((SECONDS-t0)) s" >> $O/$tag.err; python - "$O/$tag.out" "$tag" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); r=d["roofline"]; pc=d.get("parity_check",{})
    print("== %s: %.0f reads/s, %.1f ms/step, avg launch %.1f ms, parity %s, breakdown %s" % (sys.argv[2], d["value"], d["ms_per_step"], r["avg_launch_ms"],
          {k:pc.get(k) for k in ("reads","pairs","mismatching_fields","mismatching_pairs","left_the_band")}, r.get("wave_cycle_breakdown")))
except Exception as e:
    print("== %s: FAILED %s" % (sys.argv[2], e)); print(open(sys.argv[1].replace(".out",".err")).read()[-800:])
PY
}
for v in ${VARIANTS:-v1inl v2inl v1fn v2fn}; do
  run s_$v python scripts/ab_bench.py run $v --genome-mb 256 --no-extra-legs --steps 12 --warmup 3 --skip-probe --skip-refwalk --cpu-seconds ${CPUS:-3}
done
for v in ${PVARIANTS:-v1inl v2inl}; do
  run p_$v python scripts/ab_bench.py run $v --genome-mb 256 --workload paired --no-extra-legs --steps 3 --warmup 1 --cpu-seconds 3
done
