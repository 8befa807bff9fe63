"""The C-ABI collective (include/flanhip.h: flanhip_comm_* / flanhip_allgather_audio) on the one GPU of the test box: a
world of one rank exercises the run-time RCCL binding, communicator set-up and the in-place all-gather call; the multi-rank
data layout is covered on CPU by tests/test_multigpu_gloo.py and on 8 GPUs by bench.py's torch.distributed path."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_allgather_world_of_one():
    import flan_amd as fa
    lib = fa.lib
    assert lib.flanhip_device_count() > 0
    uid = C.create_string_buffer(128)
    fa.check(lib.flanhip_comm_unique_id(uid))
    assert any(uid.raw)
    comm = C.c_void_p()
    fa.check(lib.flanhip_comm_init(uid, 1, 0, C.byref(comm)))
    assert comm.value
    x = np.arange(2 * 4096, dtype=np.float32).reshape(2, 4096)          # this rank's shard: 2 channels
    d = fa.DeviceArray(host=x)
    fa.check(lib.flanhip_allgather_audio(comm, C.c_void_p(d.ptr), x.size, 0, None))
    assert np.array_equal(d.to_host(x.shape), x)
    # argument checks
    assert lib.flanhip_allgather_audio(None, C.c_void_p(d.ptr), x.size, 0, None) == fa.ERR_INVALID_ARG
    assert lib.flanhip_comm_init(uid, 2, 5, C.byref(C.c_void_p())) == fa.ERR_INVALID_ARG
    fa.check(lib.flanhip_comm_destroy(comm))


def test_bench_gather_legs_through_the_c_abi_with_one_rank():
    """bench.py's N > 1 path rehearsed with ONE rank (FLAN_BENCH_FORCE_DIST=1, RCCL backend): the headline's step with its in-place all-gather and
    BASELINE config 4's legs, both through flanhip_comm_init / flanhip_allgather_audio -- the code the driver's N = 2, 4, 8 runs execute."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["FLAN_BENCH_FORCE_DIST"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu", "--no-configs", "--seconds", "5",
                        "--config4-seconds", "7", "--preroll-ms", "0"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["value_includes_gather"] is True and d["rccl_nranks"] == 1
    ag = d["allgather"]
    assert ag["api"].startswith("flanhip_allgather_audio") and ag["api_error"] is None and ag["slots_verified"] is True
    assert d["value"] > 0 and d["value_compute_only"] > 0 and d["value_compute_then_gather"] > 0          # (no timing relation asserted: three-step windows)
    c4 = d["config4"]
    assert c4["api"].startswith("flanhip_allgather_audio") and c4["slots_verified"] is True
    F4 = 7 * 48000 // 512 + 1
    assert c4["frames_per_step"] == 8 * F4 and c4["allgather_bytes_per_rank"] == 8 * F4 * 512 * 4
    for leg in ("compute_only", "compute_then_allgather", "overlapped"):
        assert c4[leg]["frames_per_s"] > 0
