import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_lib as O, flan_amd as fa
from test_gpu_conversions import p1_metrics
x = O.sine(48000); sr = 48000.0
for (W, hop, dft) in ((256, 64, 256), (512, 128, 512)):
    ref = O.analyze(x, sr, W, hop, dft)
    for mode in (0, 1):
        with fa.debug_options(no_sub=mode):
            pv = fa.analyze(x, sr, W, hop, dft)
        print(dft, 'no_sub', mode, 'rel_m %.3e wrms %.3e same %.4f turns %d' % p1_metrics(pv, ref, sr / hop))
    with fa.debug_options(force_generic=1):
        pv = fa.analyze(x, sr, W, hop, dft)
    print(dft, 'generic', 'rel_m %.3e wrms %.3e same %.4f turns %d' % p1_metrics(pv, ref, sr / hop))
