import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flan_amd as fa
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_lib as O
lib = fa.lib
dft, hop, ch, n, W = [int(v) for v in sys.argv[1:6]]
sr = 48000.0
F = O.num_pv_frames(n, hop); bins = dft // 2 + 1
ar = np.float32(sr) / np.float32(hop)
def dev_alloc(nbytes):
    p = ctypes.c_void_p(); fa.check(lib.flanhip_malloc(ctypes.byref(p), nbytes)); return p
ws_bytes = lib.flanhip_synthesize_workspace_bytes(ch, F, bins, sr, ar, W)
d_x = dev_alloc(ch * n * 4)
d_pv, d_out, d_ws, d_flag = dev_alloc(ch * F * bins * 8), dev_alloc(ch * F * hop * 4), dev_alloc(ws_bytes), dev_alloc(4)
res = []
for seed in range(1, 13):
    x = O.noise(ch, n, seed=seed)
    fa.check(lib.flanhip_memcpy_h2d(d_x, x.ctypes.data_as(ctypes.c_void_p), x.nbytes, None))
    outs = []
    for variant in (0, 2):
        lib.flanhip_debug_option(fa.DEBUG_SYN_VARIANT, variant)
        fa.check(lib.flanhip_analyze_dev_fused(d_x, ch, n, sr, W, hop, dft, d_pv, d_ws, None))
        fa.check(lib.flanhip_synthesize_dev_fused(d_pv, ch, F, bins, sr, ar, W, d_out, d_ws, d_flag, None))
        out = np.empty((ch, F * hop), np.float32)
        fa.check(lib.flanhip_memcpy_d2h(out.ctypes.data_as(ctypes.c_void_p), d_out, out.nbytes, None))
        fa.check(lib.flanhip_stream_synchronize(None))
        outs.append(out)
    res.append(int((outs[0].view(np.uint32) != outs[1].view(np.uint32)).sum()))
lib.flanhip_debug_option(fa.DEBUG_SYN_VARIANT, 0)
print(sys.argv[1:6], "differing samples by seed:", res)
