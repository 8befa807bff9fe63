#!/usr/bin/env python3
"""When does each wavefront of the fused analysis launch start and finish?  (diagnostic build `clock`: tools/scripts/build_diag.sh clock)

    FLAN_AMD_LIB=tools/ubench/libflanhip_clock.so python tools/wave_spans.py

The launch lasts as long as its LAST wavefront: this prints the distribution of the wavefronts' lives (10 ns ticks of s_memrealtime),
the skew of their starts, and which chains finish last (first / last chain of a channel walk the clamped-load loop body)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import flan_amd as fa
    lib = ctypes.CDLL(fa.LIB_PATH)
    if not hasattr(lib, "flanhip_debug_read_spans"):
        raise SystemExit("not a stamped build: set FLAN_AMD_LIB=tools/ubench/libflanhip_clock.so")
    W, HOP, DFT, SR = 2048, 512, 2048, 48000.0
    BINS = DFT // 2 + 1
    dev = torch.device("cuda", 0)
    fa.check(fa.lib.flanhip_set_device(0))
    ch, n = 8, int(60 * SR)
    F = int(fa.lib.flanhip_num_pv_frames(n, HOP))
    stream = torch.cuda.current_stream().cuda_stream
    audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 1234, ctypes.c_void_p(stream)))
    pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
    ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, SR / HOP, W), dtype=torch.uint8, device=dev)
    synth = "--syn" in sys.argv
    for a in sys.argv:
        if a.startswith("--ana-variant="):
            fa.lib.flanhip_debug_option(fa.DEBUG_ANA_VARIANT, int(a.split("=")[1]))
            print("analysis kernel variant", a.split("=")[1])
    if "--dft4096" in sys.argv:
        DFT = 4096
        BINS = DFT // 2 + 1
        pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
        ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, SR / HOP, W), dtype=torch.uint8, device=dev)
    out = torch.empty((ch, F * HOP), dtype=torch.float32, device=dev)
    nan_flag = torch.zeros(1, dtype=torch.int32, device=dev)
    for _ in range(300):
        fa.analyze_dev_fused(audio, ch, n, SR, W, HOP, DFT, pv, ws, stream)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8192)()
    lives, starts, ends = [], [], []
    print("synthesis (main kernel)" if synth else "analysis")
    for rep in range(5):
        fa.analyze_dev_fused(audio, ch, n, SR, W, HOP, DFT, pv, ws, stream)
        if synth:
            fa.synthesize_dev_fused(pv, ch, F, BINS, SR, SR / HOP, W, out, ws, nan_flag, stream)
        lib.flanhip_debug_read_spans(buf)
        a = np.array(buf, dtype=np.uint64).reshape(4096, 2)[:2048].astype(np.int64)
        t0 = a[:, 0].min()
        starts.append((a[:, 0] - t0) * 0.01)
        ends.append((a[:, 1] - t0) * 0.01)
        lives.append((a[:, 1] - a[:, 0]) * 0.01)
    st, en, li = np.median(starts, 0), np.median(ends, 0), np.median(lives, 0)
    print("2048 wavefronts (256 per channel: chain c of a channel = wavefront c; c = 0 and 255 are the edge chains), microseconds, median of 5 launches")
    print("start after the first wavefront's: median %.2f  p95 %.2f  max %.2f" % (np.median(st), np.percentile(st, 95), st.max()))
    print("life: min %.1f  p5 %.1f  median %.1f  p95 %.1f  max %.1f" % (li.min(), np.percentile(li, 5), np.median(li), np.percentile(li, 95), li.max()))
    print("end after the first start: median %.1f  p95 %.1f  max %.1f  (the launch lasts until the max)" % (np.median(en), np.percentile(en, 95), en.max()))
    order = np.argsort(-en)[:12]
    print("last to finish: " + "  ".join("ch%d/chain%d:%.1f" % (w // 256, w % 256, en[w]) for w in order))
    edge_set = (0, 1, 254, 255) if DFT == 4096 else (0, 255)          # dft 4096: a chain is a team of two wavefronts
    edge = np.array([w for w in range(2048) if w % 256 in edge_set])
    inner = np.array([w for w in range(2048) if w % 256 not in edge_set])
    print("edge chains (first / last of a channel): life median %.1f   interior chains: %.1f" % (np.median(li[edge]), np.median(li[inner])))
    inner_blocks = np.array([b for b in range(256) if b % 32 not in (0, 31)])
    print("life by wave slot of the block (interior blocks, median): " + "  ".join("%d:%.1f" % (w, np.median(li.reshape(256, 8)[inner_blocks, w])) for w in range(8)))
    print("start by wave slot: " + "  ".join("%d:%.2f" % (w, np.median(st.reshape(256, 8)[inner_blocks, w])) for w in range(8)))
    by_group = st.reshape(8, 32, 8).mean(2)          # [channel][group of 8 chains]: when the group's wavefronts enter their frame loop
    print("start by group of the channel (median over channels): " + " ".join("%.1f" % v for v in np.median(by_group, 0)))
    by_group_end = en.reshape(8, 32, 8).max(2)
    print("end of the group's last wavefront (median over channels): " + " ".join("%.0f" % v for v in np.median(by_group_end, 0)))
    per_simd = li.reshape(256, 8)
    print("per block: fastest wave %.1f  slowest %.1f (median over blocks)" % (np.median(per_simd.min(1)), np.median(per_simd.max(1))))


if __name__ == "__main__":
    main()
