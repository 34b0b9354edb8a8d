O=gpurun_out/r5l; mkdir -p $O
for L in -1 0 3; do FLUIDX_FREEZE_DENSE_LEVELS=$L python bench.py --reference-config --no-cpu-baseline --no-render > $O/ref_$L.json 2> $O/ref_$L.err; python - $O/ref_$L.json $L <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("levels", sys.argv[2], "ms", round(d["ms_per_step"],4), "developed", d.get("developed_plume"))
PY
done
