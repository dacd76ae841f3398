"""-m gpu: every HIP kernel, called through the C ABI, against a PyTorch fp32 reference of the same op."""
import pytest
import torch

from kernel_cases import ALL_CASES, AT_SIZE_CASES, case_gemm_choice

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,fn", ALL_CASES, ids=[n for n, _ in ALL_CASES])
def test_kernel_parity(name, fn):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    res = fn()
    torch.cuda.synchronize()
    assert res["ok"], res


@pytest.mark.parametrize("name,fn", AT_SIZE_CASES, ids=[n for n, _ in AT_SIZE_CASES])
def test_kernel_parity_at_baseline_size(name, fn):
    """the kernels at the sizes of BASELINE.json config 2 (every tile configuration the table / rules select there)"""
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    res = fn()
    torch.cuda.synchronize()
    assert res["ok"], res


def test_at_size_cases_reach_the_intended_kernels():
    """the at-size cases are only worth their name if the dispatch sends them where the benchmark goes: whatever the measured
    table lists for the level-0 .. level-2 problems (exact-match entries), split-K on the 8x8-latent level"""
    for M, N, K, mode in ((1664, 1280, 11520, 1), (832, 1280, 11520, 1), (832, 1280, 3840, 2), (1664, 1280, 3840, 2)):
        c, ns = case_gemm_choice(M, N, K, mode)
        assert ns >= 2, (M, N, K, mode, c, ns)
    for M, N, K, mode in ((106496, 320, 320, 0), (106496, 960, 320, 0), (6656, 1280, 1280, 0), (106496, 320, 2880, 1)):
        c, ns = case_gemm_choice(M, N, K, mode)
        assert ns == 1 and c >= 0, (M, N, K, mode, c, ns)
