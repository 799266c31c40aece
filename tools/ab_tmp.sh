run() { (cd $2 && python bench.py --no-cpu-baseline --no-forward-section 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['frac'])"); }
for i in 1 2 3; do
run old ab_old
run new .
done
