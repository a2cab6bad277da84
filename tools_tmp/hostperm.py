import ctypes, time, sys, importlib
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg = g.load_package()
mod = importlib.import_module(pkg.__name__)
lib = mod.lib()
class Ch(ctypes.Structure):
    _fields_=[("sponge_state",ctypes.c_uint64*12),("input_buffer",ctypes.c_uint64*8),("output_buffer",ctypes.c_uint64*8),("n_input",ctypes.c_uint32),("n_output",ctypes.c_uint32)]
ch=Ch()
lib.glp_challenger_init(ctypes.byref(ch))
n=800000
data=(ctypes.c_uint64*n)(*range(1,n+1))
for _ in range(2):
    t0=time.perf_counter()
    lib.glp_challenger_observe(ctypes.byref(ch), data, ctypes.c_size_t(n))
    v=(ctypes.c_uint64*1)()
    lib.glp_challenger_get_n(ctypes.byref(ch), v, ctypes.c_size_t(1))
    dt=time.perf_counter()-t0
    print("host permutation: %.3f us" % (dt/(n/8)*1e6))
