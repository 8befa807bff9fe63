#!/usr/bin/env python3
"""Device-resident timing of Audio::resample for a few rate pairs (2 ch x 60 s), events on the null stream:
    python tools/bench_resample.py [src:dst ...]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import flan_amd as fa
    pairs = [tuple(float(v) for v in a.split(":")) for a in sys.argv[1:]] or [(96000.0, 48000.0), (44100.0, 48000.0), (48000.0, 44100.0), (48000.0, 96000.0), (44100.0, 48001.0),
                                                                             (192000.0, 48000.0), (8000.0, 44100.0)]
    fa.check(fa.lib.flanhip_set_device(0))
    dev = torch.device("cuda", 0)
    for src, dst in pairs:
        ch, n = 2, int(60 * src)
        n_out = int(fa.lib.flanhip_resample_out_frames(n, src, dst))
        x = torch.empty((ch, n), dtype=torch.float32, device=dev)
        fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(x.data_ptr()), ch, n, 7, None))
        y = torch.empty((ch, max(n_out, 1)), dtype=torch.float32, device=dev)
        call = lambda: fa.check(fa.lib.flanhip_resample_dev(ctypes.c_void_p(x.data_ptr()), ch, n, src, dst, ctypes.c_void_p(y.data_ptr()), None))
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("resample %8g -> %8g  2 ch x 60 s (%9d -> %9d frames): %8.3f ms   %7.1f M output samples/s" % (src, dst, n, n_out, ms, ch * n_out / ms / 1e3))


if __name__ == "__main__":
    main()
