"""Round trips of small inputs (BASELINE config 1: mono 5 s; a stereo minute): GPU time per round trip by events, the host side's launch cost,
and the latency of one round trip with a synchronisation.   python tools/small_latency.py"""
import ctypes, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import flan_amd as fa
W, HOP, DFT, SR = 2048, 512, 2048, 48000.0
BINS = DFT // 2 + 1
dev = torch.device("cuda", 0)
fa.check(fa.lib.flanhip_set_device(0))
for ch, secs in ((1, 5.0), (2, 60.0)):
    n = int(secs * SR)
    F = int(fa.lib.flanhip_num_pv_frames(n, HOP)); ar = SR / HOP
    audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 1, None))
    pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
    out = torch.empty((ch, F * HOP), dtype=torch.float32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, ar, W), dtype=torch.uint8, device=dev)
    def step():
        fa.analyze_dev_fused(audio, ch, n, SR, W, HOP, DFT, pv, ws, None)
        fa.synthesize_dev_fused(pv, ch, F, BINS, SR, ar, W, out, ws, flag, None)
    for _ in range(200): step()
    torch.cuda.synchronize()
    # GPU time per step (events) and host time per step (launch cost, no sync)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); t0 = time.perf_counter()
    for _ in range(500): step()
    t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize()
    print("%d ch x %g s (%d frames): GPU %.1f us per round trip, host launch side %.1f us" % (ch, secs, ch * F, e0.elapsed_time(e1) * 1000 / 500, (t1 - t0) * 1e6 / 500))
    # single-shot latency: one step, then sync
    lat = []
    for _ in range(50):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); lat.append((time.perf_counter() - t0) * 1e6)
    lat.sort(); print("   single round trip incl. sync: median %.1f us" % lat[len(lat)//2])
