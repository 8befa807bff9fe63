"""dft 4096 with window 4096: the round-1 kernels (variant 0) against the team kernels' WBIG variants (variant 1), interleaved in one process:
plain analysis, fused analysis, fused round trip at hop 1024 and 512.   python tools/ab_w4096.py"""
import ctypes, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import flan_amd as fa
W, DFT, SR = 4096, 4096, 48000.0
BINS = DFT // 2 + 1
dev = torch.device("cuda", 0)
fa.check(fa.lib.flanhip_set_device(0))
for hop in (1024, 512):
    ch, n = 8, int(60 * SR)
    F = int(fa.lib.flanhip_num_pv_frames(n, hop))
    ar = SR / hop
    audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 1234, None))
    pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
    out = torch.empty((ch, F * hop), dtype=torch.float32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, ar, W), dtype=torch.uint8, device=dev)
    res = {}
    def t(fn, reps=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    for rnd in range(5):
        for v in (0, 1):
            fa.lib.flanhip_debug_option(fa.DEBUG_ANA4096_OLD, 1 - v); fa.lib.flanhip_debug_option(fa.DEBUG_SYN4096_OLD, 1 - v)
            for _ in range(20 if rnd == 0 else 3):
                fa.analyze_dev(audio, ch, n, SR, W, hop, DFT, pv, None)
            res.setdefault((v, "ana_plain"), []).append(t(lambda: fa.analyze_dev(audio, ch, n, SR, W, hop, DFT, pv, None)))
            res.setdefault((v, "ana_fused"), []).append(t(lambda: fa.analyze_dev_fused(audio, ch, n, SR, W, hop, DFT, pv, ws, None)))
            res.setdefault((v, "step_fused"), []).append(t(lambda: (fa.analyze_dev_fused(audio, ch, n, SR, W, hop, DFT, pv, ws, None), fa.synthesize_dev_fused(pv, ch, F, BINS, SR, ar, W, out, ws, flag, None))))
    for k, ms in sorted(res.items()):
        ms = sorted(ms); print("hop", hop, "variant", k[0], k[1], "median %.4f min %.4f ms" % (ms[len(ms)//2], ms[0]))
    fa.lib.flanhip_debug_option(fa.DEBUG_ANA4096_OLD, 1 - 1); fa.lib.flanhip_debug_option(fa.DEBUG_SYN4096_OLD, 1 - 1)
