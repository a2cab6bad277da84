"""Clocks, power and temperature (rocm-smi) sampled while one thread proves the 2^20-row 14-gate circuit back to back for 30 s: is the proof time a function of the box's sustained clock?"""
import importlib, os, subprocess, sys, threading, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.load_package()
circ = importlib.import_module(pkg.__name__ + ".circuit")
ctx = pkg.Context(0)
lc = circ.MixedGateCircuit(20, 16)
pc = pkg.ProverCircuit(ctx, lc.data)
w, pis = lc.witness_device(ctx, pkg, seed=0)
pc.prove(w, pis)
stop = False
times = []
def loop():
    import torch
    torch.cuda.set_device(0)
    while not stop:
        t0 = time.perf_counter(); pc.prove(w, pis); times.append(time.perf_counter() - t0)
th = threading.Thread(target=loop); th.start()
for i in range(12):
    time.sleep(2.5)
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True).stdout
    keep = [ln.split(":", 1)[1].strip() if ":" in ln else ln for ln in out.splitlines() if any(k in ln for k in ("sclk", "Power (W)", "junction"))]
    print("t=%4.1f s  proofs so far %3d  last %.2f ms | %s" % (2.5 * (i + 1), len(times), 1e3 * times[-1] if times else 0, " | ".join(keep)))
stop = True; th.join()
print("first 5: %s  last 5: %s" % ([round(1e3 * t, 2) for t in times[:5]], [round(1e3 * t, 2) for t in times[-5:]]))
