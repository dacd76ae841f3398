"""-m gpu: every HIP kernel, called through the C ABI, against a PyTorch fp32 reference of the same op."""
import pytest
import torch

from kernel_cases import ALL_CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,fn", ALL_CASES, ids=[n for n, _ in ALL_CASES])
def test_kernel_parity(name, fn):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    res = fn()
    torch.cuda.synchronize()
    assert res["ok"], res
