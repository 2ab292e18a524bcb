cd "$GRAFT_REPO_ROOT"
for i in 1 2; do for f in 2 3; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-traffic --no-pipelined --no-kernel-pass --steps 36 --in-flight $f > gpurun_out/abf.json 2> gpurun_out/abf.err || { tail -3 gpurun_out/abf.err; exit 1; }
python -c "
import json; r=json.load(open('gpurun_out/abf.json')); print('in flight', $f, r['value'], r['ms_per_step'], 'serial', r['serial']['value'])"
done; done
