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
