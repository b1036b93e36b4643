for r in 1 2 3; do for sw in "60 6" "300 50" "1000 100"; do set -- $sw
python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-dense-reference 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps $1 warmup $2', round(d['value']), round(d['ms_per_step'],4))"
done; done
