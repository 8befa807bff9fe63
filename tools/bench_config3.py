#!/usr/bin/env python3
"""BASELINE config 3 on one MI355X, device-resident: 8 ch x 60 s noise -> convert_to_PV(2048,512,2048) -> stretch(x2) ->
convert_to_audio, timed per stage with events (also repitch(x2) and shape(f+100) on the same PV)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flan_amd as fa

SR, W, HOP, DFT = 48000.0, 2048, 512, 2048
BINS = DFT // 2 + 1
ch, n = 8, 60 * 48000
lib = fa.lib
dev = torch.device("cuda", 0)
F = int(lib.flanhip_num_pv_frames(n, HOP))
ar = SR / HOP
vp = ctypes.c_void_p


def P(t):
    return vp(t.data_ptr())


audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
fa.check(lib.flanhip_noise_dev(P(audio), ch, n, 1234, None))
pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
grid = torch.empty((F, BINS), dtype=torch.float32, device=dev)
dmax = torch.empty(1, dtype=torch.float32, device=dev)
Fo = 2 * F
st = torch.empty((ch, Fo, BINS, 2), dtype=torch.float32, device=dev)
out = torch.empty((ch, Fo * HOP), dtype=torch.float32, device=dev)
ws = torch.empty(fa.synthesize_workspace_bytes(ch, Fo, BINS, SR, ar, W), dtype=torch.uint8, device=dev)
inmod = torch.empty((ch, F, BINS), dtype=torch.float32, device=dev)
rp = torch.empty_like(pv)
sh = torch.empty_like(pv)

grid_t = torch.empty((F, BINS), dtype=torch.float32, device=dev)      # the x2 time map, kept for the hand-over stage (grid is reused below)
fa.check(lib.flanhip_fill_dev(P(grid_t), F * BINS, 2.0, None))
fa.check(lib.flanhip_stretch_map_dev(P(grid_t), F, BINS, SR, HOP, P(dmax), None))
ws_sh = torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, ar, W), dtype=torch.uint8, device=dev)
out_sh = torch.empty((ch, F * HOP), dtype=torch.float32, device=dev)
stages = {
    "convert_to_PV": lambda: fa.check(lib.flanhip_analyze_dev(P(audio), ch, n, SR, W, HOP, DFT, P(pv), None)),
    "stretch: fill+map": lambda: (fa.check(lib.flanhip_fill_dev(P(grid), F * BINS, 2.0, None)),
                                  fa.check(lib.flanhip_stretch_map_dev(P(grid), F, BINS, SR, HOP, P(dmax), None))),
    "stretch: modify_time": lambda: fa.check(lib.flanhip_modify_time_dev(P(pv), ch, F, BINS, SR, HOP, P(grid), Fo, P(st), None)),
    "stretch: modify_time + convert_to_audio, pre-pass handed over": lambda: (
        fa.check(lib.flanhip_modify_time_dev_fused(P(pv), ch, F, BINS, SR, ar, P(grid_t), Fo, P(st), W, P(ws), None)),
        fa.check(lib.flanhip_synthesize_dev_fused_checked(P(st), ch, Fo, BINS, SR, ar, W, P(out), P(ws), None, None))),
    "convert_to_audio(stretched)": lambda: fa.check(lib.flanhip_synthesize_dev(P(st), ch, Fo, BINS, SR, ar, W, P(out), P(ws), None, None)),
    "repitch: fill+map": lambda: (fa.check(lib.flanhip_fill_dev(P(grid), F * BINS, 2.0, None)),
                                  fa.check(lib.flanhip_repitch_map_dev(P(pv), ch, F, BINS, SR, P(grid), P(inmod), None))),
    "repitch: modify_frequency": lambda: fa.check(lib.flanhip_modify_frequency_dev(P(pv), ch, F, BINS, SR, P(grid), P(inmod), P(rp), None)),
    "repitch: fused (fill + scan + modify_frequency)": lambda: (fa.check(lib.flanhip_fill_dev(P(grid), F * BINS, 2.0, None)),
                                                                fa.check(lib.flanhip_repitch_dev(P(pv), ch, F, BINS, SR, P(grid), P(rp), None))),
    "shape(f+100)": lambda: fa.check(lib.flanhip_shape_affine_dev(P(pv), ch, F, BINS, SR, 1.0, 0.0, 1.0, 100.0, 0, P(sh), None)),
    "shape(f+100) + convert_to_audio, pre-pass handed over": lambda: (
        fa.check(lib.flanhip_shape_affine_dev_fused(P(pv), ch, F, BINS, SR, ar, 1.0, 0.0, 1.0, 100.0, P(sh), W, P(ws_sh), None)),
        fa.check(lib.flanhip_synthesize_dev_fused(P(sh), ch, F, BINS, SR, ar, W, P(out_sh), P(ws_sh), None, None))),
    "shape(f+100) + convert_to_audio, separate": lambda: (
        fa.check(lib.flanhip_shape_affine_dev(P(pv), ch, F, BINS, SR, 1.0, 0.0, 1.0, 100.0, 0, P(sh), None)),
        fa.check(lib.flanhip_synthesize_dev(P(sh), ch, F, BINS, SR, ar, W, P(out_sh), P(ws_sh), None, None))),
    "shape(f*2, aligned)": lambda: fa.check(lib.flanhip_shape_affine_dev(P(pv), ch, F, BINS, SR, 1.0, 0.0, 2.0, 0.0, 1, P(sh), None)),
}
# further frame processors on the same 8 ch x 60 s PV: ms and the HBM rate of the MF traffic (8 B in + 8 B out)
Fr = int(lib.flanhip_resonate_out_frames(F, 1.0, SR, HOP))
res_out = torch.empty((ch, Fr, BINS, 2), dtype=torch.float32, device=dev)
decay = torch.full((Fr, BINS), 0.7, dtype=torch.float32, device=dev)
ext_start, ext_end, ext_Fo = 1000, 4000, 6000
samples = torch.linspace(0, 1, ext_Fo - ext_start, dtype=torch.float32, device=dev)
ext_out = torch.empty((ch, ext_Fo, BINS, 2), dtype=torch.float32, device=dev)
stages.update({
    "replace_amplitudes(0.5)": lambda: fa.check(lib.flanhip_replace_amplitudes_dev(P(pv), ch, F, BINS, P(rp), ch, F, BINS, None, 0.5, P(sh), None)),
    "subtract_amplitudes(grid)": lambda: fa.check(lib.flanhip_subtract_amplitudes_dev(P(pv), ch, F, BINS, P(rp), ch, F, BINS, P(grid), 0.0, P(sh), None)),
    "resonate(1 s, 0.7)": lambda: fa.check(lib.flanhip_resonate_dev(P(pv), ch, F, BINS, SR, HOP, Fr, None, 0.7, P(res_out), None)),
    "resonate(1 s, grid)": lambda: fa.check(lib.flanhip_resonate_dev(P(pv), ch, F, BINS, SR, HOP, Fr, P(decay), 0.0, P(res_out), None)),
    "retain_n_loudest(64)": lambda: fa.check(lib.flanhip_n_loudest_partials_dev(P(pv), ch, F, BINS, None, 64, 0, P(sh), None)),
    "desample(0.1)": lambda: fa.check(lib.flanhip_desample_dev(P(pv), ch, F, BINS, None, 0.1, 0, P(sh), None)),
    "time_extrapolate(2000 frames)": lambda: fa.check(lib.flanhip_time_extrapolate_dev(P(pv), ch, F, BINS, SR, ext_start, ext_end, ext_Fo,
                                                                                         P(samples), P(ext_out), None)),
})
# selecting / re-placing methods on the same PV
fz_src_host = fa.freeze_plan(F, SR, HOP, [10.0, 20.0, 30.0], [1.0, 2.0, 0.5])
fz_Fo = len(fz_src_host)
fz_src = torch.from_numpy(fz_src_host).to(dev)
fz_out = torch.empty((ch, fz_Fo, BINS, 2), dtype=torch.float32, device=dev)
sel_grid = torch.empty((F, BINS, 2), dtype=torch.float32, device=dev)
sel_grid[..., 0] = (torch.arange(F, device=dev, dtype=torch.float32) / (SR / HOP))[:, None] * 0.5 + 1.0      # half speed from 1 s in
sel_grid[..., 1] = (torch.arange(BINS, device=dev, dtype=torch.float32) * (SR / DFT))[None, :] * 0.75
H_oct = 15
ser_oct = torch.full((F, H_oct), 0.5, dtype=torch.float32, device=dev)
ser_har = torch.full((F, BINS), 0.5, dtype=torch.float32, device=dev)
sm_e = int(0.1 * SR / HOP)
sm_n = 4 * sm_e
sm_dist = (0.5 * (1.0 + torch.cos(torch.pi * torch.arange(-2 * sm_e, 2 * sm_e, device=dev, dtype=torch.float32) / (2 * sm_e)))).contiguous()
sm_out = torch.empty((ch, F - 1 + 2 * sm_e, BINS, 2), dtype=torch.float32, device=dev)
warp = torch.empty((F, BINS, 2), dtype=torch.float32, device=dev)                # modify: twice as slow, a fifth up
warp[..., 0] = (torch.arange(F, device=dev, dtype=torch.float32) / (SR / HOP))[:, None] * 2.0
warp[..., 1] = (torch.arange(BINS, device=dev, dtype=torch.float32) * (SR / DFT))[None, :] * 1.2
warp_f = (pv[..., 1] * 1.2).contiguous()
warp_Fo = 2 * (F - 1)                                                            # = flanhip_modify_out_frames of this grid
warp_out = torch.empty((ch, warp_Fo, BINS, 2), dtype=torch.float32, device=dev)
import numpy as _np
sp_steps = _np.full(F - 1, 2, _np.uint32)
sp_out = torch.empty((ch, 2 * (F - 1), BINS, 2), dtype=torch.float32, device=dev)
stages.update({
    "stretch_spline(x2)": lambda: fa.check(lib.flanhip_stretch_spline_dev(P(pv), ch, F, BINS, fa._vp(sp_steps.ctypes.data), 2 * (F - 1), P(sp_out), None)),
    "modify(2 t, 1.2 f)": lambda: fa.check(lib.flanhip_modify_dev(P(pv), ch, F, BINS, SR, HOP, P(warp), P(warp_f), 0, warp_Fo, P(warp_out), None)),
    "smear_time(0.1 s, 5)": lambda: fa.check(lib.flanhip_smear_time_dev(P(pv), ch, F, BINS, SR, HOP, None, 0.1, None, 5, P(sm_dist), sm_n, -sm_e, F - 1 + 2 * sm_e, P(sm_out), None)),
    "freeze(3 pauses)": lambda: fa.check(lib.flanhip_select_frames_dev(P(pv), ch, F, BINS, P(fz_src), fz_Fo, P(fz_out), None)),
    "cut_frames(1000:5000)": lambda: fa.check(lib.flanhip_cut_frames_dev(P(pv), ch, F, BINS, 1000, 4000, P(sh), None)),
    "join(one input)": lambda: fa.check(lib.flanhip_place_frames_dev(P(pv), ch, F, BINS, P(fz_out), ch, fz_Fo, BINS, 100, None)),
    "select(grid)": lambda: fa.check(lib.flanhip_select_dev(P(pv), ch, F, BINS, SR, HOP, P(sel_grid), F, P(sh), None)),
    "add_octaves": lambda: fa.check(lib.flanhip_harmonic_scale_dev(P(pv), ch, F, BINS, SR, P(ser_oct), H_oct, 0, P(sh), None)),
    "add_harmonics": lambda: fa.check(lib.flanhip_harmonic_scale_dev(P(pv), ch, F, BINS, SR, P(ser_har), BINS, 1, P(sh), None)),
})
# device warm-up: an idle MI355X needs tens of milliseconds of load before its clocks settle (see bench.py --preroll-ms)
import time
t_end = time.perf_counter() + 0.1
while time.perf_counter() < t_end:
    stages["convert_to_PV"](); stages["shape(f+100)"](); torch.cuda.synchronize()
res = {}
for name, fn in stages.items():
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    res[name] = round(e0.elapsed_time(e1) / reps, 4)
# config 5's resample stage: stereo 60 s at 96 kHz -> 48 kHz
x96 = torch.empty((2, 60 * 96000), dtype=torch.float32, device=dev)
fa.check(lib.flanhip_noise_dev(P(x96), 2, 60 * 96000, 7, None))
x48 = torch.empty((2, 60 * 48000), dtype=torch.float32, device=dev)
fn5 = lambda: fa.check(lib.flanhip_resample_dev(P(x96), 2, 60 * 96000, 96000.0, 48000.0, P(x48), None))
fn5(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    fn5()
e1.record(); torch.cuda.synchronize()
res["config5: resample 2x60s 96k->48k"] = round(e0.elapsed_time(e1) / 5, 4)
total = res["convert_to_PV"] + res["stretch: fill+map"] + res["stretch: modify_time + convert_to_audio, pre-pass handed over"]
total_plain = res["convert_to_PV"] + res["stretch: fill+map"] + res["stretch: modify_time"] + res["convert_to_audio(stretched)"]
print(json.dumps({"config": "3: 8ch 60s stretch x2", "input_frames": ch * F, "stage_ms": res, "config3_total_ms": round(total, 3), "config3_total_ms_without_handover": round(total_plain, 3),
                  "input_frames_per_s": round(ch * F / (total * 1e-3), 1)}))
