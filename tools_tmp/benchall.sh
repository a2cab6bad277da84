python -m pytest tests/test_gpu_plonk.py -x -q -k "stream or batched" 2>&1 | tail -2
python bench.py > gpurun_out/r02i_bench.json 2> gpurun_out/r02i_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r02i_bench.err
python bench.py --workload batch64 --steps 2 > gpurun_out/r02i_batch64.json 2> gpurun_out/r02i_batch64.err; echo "batch64 rc=$?"; tail -2 gpurun_out/r02i_batch64.err
python bench.py --workload aggtree --steps 2 > gpurun_out/r02i_aggtree.json 2> gpurun_out/r02i_aggtree.err; echo "aggtree rc=$?"; tail -2 gpurun_out/r02i_aggtree.err
GLP_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --workload aggtree --steps 1 --proofs 16 --log-n 12 > gpurun_out/r02i_aggtree_2rank.json 2> gpurun_out/r02i_aggtree_2rank.err; echo "2rank rc=$?"; tail -2 gpurun_out/r02i_aggtree_2rank.err
python - <<'PY'
import json
for f in ("r02i_bench","r02i_batch64","r02i_aggtree","r02i_aggtree_2rank"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f))
    except Exception as e:
        print(f, "no json", e); continue
    print(f, "value", round(d["value"],3), d["unit"], "ms/step", round(d["ms_per_step"],2), "verified", d.get("proof_verified_by_restated_verifier"))
    for k in ("phase_ms_per_step","poseidon_only_circuit","small_circuit_batches","host_resident_witness","roofline_quotient","levels","time_in_collective_ms_per_step"):
        if k in d: print("   ", k, json.dumps(d[k])[:600])
    if d.get("cpu_baseline"): print("    cpu", json.dumps({k:v for k,v in d["cpu_baseline"].items() if k!="sample"})[:800])
PY
