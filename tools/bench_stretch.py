#!/usr/bin/env python3
"""Config 3's stretch stage alone, with this box's yardsticks beside it: ms of flanhip_modify_time_dev_fused (stretch x2 of 8 ch x 60 s,
the pre-pass left for convert_to_audio) and of the plain flanhip_modify_time_dev, against a device copy of the input PV and a fill of the
output PV timed in the same process.  (Two builds are compared with FLAN_AMD_LIB=... runs of this script; the PV / chain-sum checksums
printed are equal when the outputs are.)"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flan_amd as fa

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--factor", type=float, default=2.0)
ap.add_argument("--channels", type=int, default=8)
args = ap.parse_args()
variants = [0]
SR, W, HOP, DFT = 48000.0, 2048, 512, 2048
BINS = DFT // 2 + 1
ch, n = args.channels, 60 * 48000
lib = fa.lib
dev = torch.device("cuda", 0)
F = int(lib.flanhip_num_pv_frames(n, HOP))
ar = SR / HOP
P = lambda t: ctypes.c_void_p(t.data_ptr())
audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
fa.check(lib.flanhip_noise_dev(P(audio), ch, n, 1234, None))
pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
fa.check(lib.flanhip_analyze_dev(P(audio), ch, n, SR, W, HOP, DFT, P(pv), None))
grid = torch.empty((F, BINS), dtype=torch.float32, device=dev)
dmax = torch.empty(1, dtype=torch.float32, device=dev)
fa.check(lib.flanhip_fill_dev(P(grid), F * BINS, args.factor, None))
fa.check(lib.flanhip_stretch_map_dev(P(grid), F, BINS, SR, HOP, P(dmax), None))
torch.cuda.synchronize()
Fo = int(torch.ceil(dmax * SR / HOP).item())
st = torch.empty((ch, Fo, BINS, 2), dtype=torch.float32, device=dev)
ws = torch.empty(fa.synthesize_workspace_bytes(ch, Fo, BINS, SR, ar, W), dtype=torch.uint8, device=dev)
fused = lambda: fa.check(lib.flanhip_modify_time_dev_fused(P(pv), ch, F, BINS, SR, ar, P(grid), Fo, P(st), W, P(ws), None))
plain = lambda: fa.check(lib.flanhip_modify_time_dev(P(pv), ch, F, BINS, SR, HOP, P(grid), Fo, P(st), None))
# yardsticks on this box, now: a copy of the input PV into the first half of the output and a fill of the whole output
import time
t_end = time.perf_counter() + 0.1
while time.perf_counter() < t_end:
    fused(); torch.cuda.synchronize()
def ev(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
flat_out, flat_in = st.view(-1), pv.view(-1)
yard = {"copy_in_to_out_ms": round(ev(lambda: flat_out[: flat_in.numel()].copy_(flat_in)), 4), "fill_out_ms": round(ev(lambda: st.fill_(1.0)), 4)}
yard["copy_TBs"] = round(2 * flat_in.numel() * 4 / yard["copy_in_to_out_ms"] / 1e9, 2)
yard["fill_TBs"] = round(st.numel() * 4 / yard["fill_out_ms"] / 1e9, 2)
res = {v: {"fused": [], "plain": []} for v in variants}
ref = {}
same = {}
for v in variants:
    st.zero_(); ws.zero_()
    fused(); torch.cuda.synchronize()
    a, b = st.clone(), ws[: ch * 8 * BINS * 64].clone()           # PV and the head of the chain sums
    st.zero_(); plain(); torch.cuda.synchronize()
    c = st.clone()
    if not ref:
        ref = dict(a=a, b=b, c=c)
    same[v] = dict(pv_fused_checksum=int(a.view(torch.int32).to(torch.int64).sum().item()), sums_checksum=int(b.to(torch.int64).sum().item()),
                   pv_plain_equals_fused=bool(torch.equal(c.view(torch.int32), a.view(torch.int32))))
    del a, b, c
for r in range(args.rounds):
    for v in variants:
        for name, fn in (("fused", fused), ("plain", plain)):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            res[v][name].append(e0.elapsed_time(e1) / args.reps)
out = {"F": F, "Fo": Fo, "yardsticks": yard, "bytes_moved_MB": round((ch * (F + Fo) * BINS * 8 + F * BINS * 4) / 1e6, 1)}
for v in variants:
    out.update({k: {"median_ms": round(sorted(x)[len(x) // 2], 4), "min_ms": round(min(x), 4)} for k, x in res[v].items()})
    out["outputs"] = same[v]
print(json.dumps(out, indent=1))
