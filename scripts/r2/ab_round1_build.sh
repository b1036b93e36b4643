# same-box A/B of the round-1 final build (ab/r1 = git archive of a455e70, built locally) against the working tree:
# C3 (metric config), C2 medium at 512^3 (shear), C5 1024^3; interleaved, two rounds
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_ab; mkdir -p $O
for rep in 1 2; do
  for cfg in "C3:" "C2:--config C2 --size 512 512 512"; do
    tag=${cfg%%:*}; a=${cfg#*:}
    (cd ab/r1 && timeout 600 python bench.py $a --steps 300 --warmup 150 --no-cpu-baseline --no-dense-reference) > $O/r1_${tag}_$rep.json 2>/dev/null
    timeout 600 python bench.py $a --steps 300 --warmup 150 --no-cpu-baseline --no-shear-workload --no-kernel-pass > $O/r2_${tag}_$rep.json 2>/dev/null
  done
done
(cd ab/r1 && timeout 900 python bench.py --config C5 --lean-host --steps 60 --warmup 20 --no-cpu-baseline --no-dense-reference) > $O/r1_C5_1.json 2>/dev/null
timeout 900 python bench.py --config C5 --scaling strong --steps 60 --warmup 20 --no-cpu-baseline --no-kernel-pass > $O/r2_C5_1.json 2>/dev/null
(cd ab/r1 && timeout 900 python bench.py --config C4 --lean-host --steps 100 --warmup 30 --no-cpu-baseline --no-dense-reference) > $O/r1_C4_1.json 2>/dev/null
timeout 900 python bench.py --config C4 --scaling strong --steps 100 --warmup 30 --no-cpu-baseline --no-kernel-pass > $O/r2_C4_1.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2_ab/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4))
    except Exception as e: print(f, 'failed', e)
PY
