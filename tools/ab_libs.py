#!/usr/bin/env python3
"""A/B of TWO BUILDS of libflanhip.so in one process, interleaved rounds (the module flan_amd is loaded twice, once per library).

    python tools/ab_libs.py --a flan_amd/libflanhip_base.so --b flan_amd/libflanhip.so [--dft 2048] [--hop 512] [--rounds 9] [--reps 20]

Per library: median / min ms of the fused analysis launch, of everything convert_to_audio launches, and of the whole step, on the bench
shape (8 ch x 60 s); and whether B's PV and audio are bit-identical to A's.
"""
import argparse
import ctypes
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name, path):
    os.environ["FLAN_AMD_LIB"] = os.path.abspath(path)
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "flan_amd", "__init__.py"),
                                                  submodule_search_locations=[os.path.join(ROOT, "flan_amd")])
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", required=True)
    ap.add_argument("--b", required=True)
    ap.add_argument("--dft", type=int, default=2048)
    ap.add_argument("--hop", type=int, default=512)
    ap.add_argument("--window", type=int, default=2048)
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import numpy as np
    import torch
    libs = {"A": load("fa_a", args.a), "B": load("fa_b", args.b)}
    W, HOP, DFT, SR = args.window, args.hop, args.dft, 48000.0
    BINS = DFT // 2 + 1
    dev = torch.device("cuda", 0)
    ch, n = args.channels, int(args.seconds * SR)
    ar = SR / HOP
    st = {}
    for k, fa in libs.items():
        fa.check(fa.lib.flanhip_set_device(0))
        F = int(fa.lib.flanhip_num_pv_frames(n, HOP))
        audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
        fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 1234, None))
        st[k] = dict(F=F, audio=audio, pv=torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev),
                     out=torch.empty((ch, F * HOP), dtype=torch.float32, device=dev), flag=torch.zeros(1, dtype=torch.int32, device=dev),
                     ws=torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, ar, W), dtype=torch.uint8, device=dev),
                     ana=[], syn=[], step=[])
    torch.cuda.synchronize()

    def ana(k):
        s, fa = st[k], libs[k]
        fa.analyze_dev_fused(s["audio"], ch, n, SR, W, HOP, DFT, s["pv"], s["ws"], None)

    def syn(k):
        s, fa = st[k], libs[k]
        fa.synthesize_dev_fused(s["pv"], ch, s["F"], BINS, SR, ar, W, s["out"], s["ws"], s["flag"], None)

    def timed(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for k in libs:                                   # warm-up (and the device's clocks)
        for _ in range(150):
            ana(k); syn(k)
    torch.cuda.synchronize()
    for r in range(args.rounds):
        for k in (("A", "B") if r % 2 == 0 else ("B", "A")):
            st[k]["ana"].append(timed(lambda: ana(k), args.reps))
            st[k]["syn"].append(timed(lambda: syn(k), args.reps))
            st[k]["step"].append(timed(lambda: (ana(k), syn(k)), args.reps))
    res = {"shape": {"channels": ch, "seconds": args.seconds, "window": W, "hop": HOP, "dft": DFT, "frames": ch * st["A"]["F"]}}
    for k in libs:
        res[k] = {"lib": args.a if k == "A" else args.b}
        for what in ("ana", "syn", "step"):
            v = sorted(st[k][what])
            res[k][what + "_ms_median"] = round(v[len(v) // 2], 5)
            res[k][what + "_ms_min"] = round(v[0], 5)
    res["B_over_A"] = {w: round(res["B"][w + "_ms_median"] / res["A"][w + "_ms_median"], 4) for w in ("ana", "syn", "step")}
    res["pv_bit_identical"] = bool(torch.equal(st["A"]["pv"].view(torch.int32), st["B"]["pv"].view(torch.int32)))
    res["audio_bit_identical"] = bool(torch.equal(st["A"]["out"].view(torch.int32), st["B"]["out"].view(torch.int32)))
    line = json.dumps(res)
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
