"""The ctypes layer's device guard (vit_colmap_amd/_lib.py): every call runs on the GPU its tensors live on and
mixed-device calls are refused.  CPU part: no kernel is launched."""
import ctypes

import pytest
import torch

from vit_colmap_amd import _lib


class _FakeTensor:
    def __init__(self, device):
        self.device, self.is_cuda = torch.device(device), True

    def data_ptr(self):
        return 0x1000


def test_ptr_remembers_the_device_and_none_stays_none():
    assert _lib.ptr(None) is None
    p = _lib.ptr(_FakeTensor("cuda:3"))
    assert isinstance(p, ctypes.c_void_p) and p.device == torch.device("cuda:3") and p.value == 0x1000
    assert _lib.ptr(torch.zeros(4)).device is None      # host tensors carry no GPU


def test_mixed_device_call_is_refused_before_it_reaches_the_library():
    called = []
    guarded = _lib._guarded(lambda *a: called.append(a) or 0, "vc_fake")
    with pytest.raises(_lib.HipLibraryError, match="different devices"):
        guarded(_lib.ptr(_FakeTensor("cuda:0")), _lib.ptr(_FakeTensor("cuda:1")), _lib.stream_ptr())
    assert not called


def test_host_only_calls_pass_through():
    lib = _lib.load()
    assert lib.vc_theta_table(None, 4, None) == -1       # argument validation, no device involved


@pytest.mark.gpu
def test_call_runs_on_the_tensors_device_not_the_current_one():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from vit_colmap_amd.matching import theta_table

    torch.cuda.set_device(0)
    t1 = theta_table(1024, device="cuda:1")
    t0 = theta_table(1024, device="cuda:0")
    assert torch.cuda.current_device() == 0
    assert torch.equal(t0.cpu(), t1.cpu())
