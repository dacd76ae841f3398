"""-m gpu: whole-network parity, HIP model vs the CPU oracle on identical seeded weights/inputs (|delta|max < 1e-2)."""
import pytest
import torch

from model_cases import MODEL_CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,fn", MODEL_CASES, ids=[n for n, _ in MODEL_CASES])
def test_unet_parity(name, fn):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    res = fn()
    assert res["ok"], res
