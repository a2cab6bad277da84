python -m pytest tests/test_gpu_plonk.py -x -q 2>&1 | tail -2
for c in mixed poseidon; do python bench.py --steps 5 --warmup 1 --circuit $c --no-cpu-baseline > gpurun_out/q_$c.json 2>gpurun_out/q_$c.err; done
python -c "
import json
for f in ('q_mixed','q_poseidon'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, round(d['ms_per_step'],2), 'plonk', round(d['phase_ms_per_step']['plonk'],2))
"
