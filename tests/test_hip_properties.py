"""Size-independent properties at the BENCHMARK sizes (U-Net batch 8, 64x64 latent), where the CPU oracle is too slow to be the
checker: invariances the reference's arithmetic has by construction.  `pytest -m gpu`."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adaface_dev_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _rnd(shape, seed, scale=1.0):
    return (torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale).half()


def test_attention_rows_are_convex_combinations_full_size(dev):
    """softmax rows sum to 1: with V = a per-channel constant the output equals that constant for every query, for the 64x64
    self-attention (two-chain kernel, 4096 keys), the 32x32 one and the 77-key cross-attention (short-key kernel)."""
    from adaface_dev_amd import ops
    for B, N, L, H, d in ((8, 4096, 4096, 8, 40), (8, 1024, 1024, 8, 80), (8, 4096, 77, 8, 40), (8, 256, 77, 8, 160)):
        C = H * d
        q, k = _rnd((B * N, C), 1).to(dev), _rnd((B * L, C), 2).to(dev)
        const = _rnd((C,), 3)
        v = const[None, None, :].expand(B, L, C).contiguous().to(dev)
        vt = ops.transpose_tokens(v.reshape(B * L, C), B, L, C, C)
        o = ops.attention(q, k, vt, B=B, Nq=N, L=L, heads=H, d=d, ldq=C, ldk=C)
        err = (o.float() - const.float().to(dev)[None, :]).abs().max().item()
        assert err < 4e-3 * const.float().abs().max().item() + 1e-3, (N, L, d, err)


def test_gemm_and_conv_are_linear_full_size(dev):
    """f(a x1 + x2) = a f(x1) + f(x2) (no bias) for the largest linear layer and the largest 3x3 convolution of the step."""
    from adaface_dev_amd import ops
    M, K, N = 32768, 320, 320
    w = _rnd((N, K), 4, K ** -0.5)
    pw = ops.pack_matrix(w, None, dev)
    x1, x2 = _rnd((M, K), 5).to(dev), _rnd((M, K), 6).to(dev)
    lhs = ops.gemm((x1.float() * 0.5 + x2.float()).half(), pw).float()
    rhs = ops.gemm(x1, pw).float() * 0.5 + ops.gemm(x2, pw).float()
    assert rel_l2(lhs.cpu().numpy(), rhs.cpu().numpy()) < 2e-3
    wc = _rnd((320, 640, 3, 3), 7, (640 * 9) ** -0.5)
    pc = ops.pack_conv3x3(wc, None, dev)
    y1, y2 = _rnd((8, 64, 64, 640), 8).to(dev), _rnd((8, 64, 64, 640), 9).to(dev)
    lhs = ops.conv3x3((y1.float() * 0.5 + y2.float()).half(), pc).float()
    rhs = ops.conv3x3(y1, pc).float() * 0.5 + ops.conv3x3(y2, pc).float()
    assert rel_l2(lhs.cpu().numpy(), rhs.cpu().numpy()) < 2e-3


def test_norms_are_invariant_to_input_affine_maps_full_size(dev):
    """GroupNorm(a x + b) = GroupNorm(x) for a > 0 (per tensor; eps aside) and LayerNorm likewise; GroupNorm of the concatenation
    equals GroupNorm over the two-source form."""
    from adaface_dev_amd import ops
    B, HW, C = 8, 4096, 320
    x = _rnd((B, 64, 64, C), 10).to(dev)
    gamma, beta = (1 + 0.1 * torch.randn(C)).to(dev), (0.1 * torch.randn(C)).to(dev)
    y = ops.groupnorm(x, gamma, beta, 1e-6, True)
    y2 = ops.groupnorm((x.float() * 2.0 + 3.0).half(), gamma, beta, 1e-6, True)
    assert rel_l2(y2.float().cpu().numpy(), y.float().cpu().numpy()) < 4e-3
    ycat = ops.groupnorm(x[..., :192].contiguous(), gamma, beta, 1e-6, True, x2=x[..., 192:].contiguous())
    assert torch.equal(ycat, y)
    r = _rnd((B * HW, C), 11).to(dev)
    ln = ops.layernorm(r, gamma, beta)
    ln2 = ops.layernorm((r.float() * 4.0 - 1.0).half(), gamma, beta)
    assert rel_l2(ln2.float().cpu().numpy(), ln.float().cpu().numpy()) < 4e-3


def test_cfg_ddim_step_identities(dev):
    """guidance 1 ignores the unconditional half; alpha_prev = alpha_t returns x unchanged (ddim.py:253-302 arithmetic)."""
    from adaface_dev_amd import ops
    x = torch.randn(4, 4, 64, 64, generator=torch.Generator().manual_seed(12)).to(dev)
    e2 = torch.randn(8, 4, 64, 64, generator=torch.Generator().manual_seed(13)).to(dev)
    a, _ = ops.cfg_ddim_step(e2, x, 1.0, 0.5, 0.6, True)
    b, _ = ops.cfg_ddim_step(e2[:4].contiguous(), x, 1.0, 0.5, 0.6, False)
    assert torch.allclose(a, b, atol=1e-6)
    same, _ = ops.cfg_ddim_step(e2, x, 3.0, 0.5, 0.5, True)
    assert torch.allclose(same, x, atol=2e-5)
