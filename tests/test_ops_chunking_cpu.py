"""ops.lora_down / ops.lora_wgrad above 64 ranks: the host side cuts the product into 64-rank launches on views of ONE slab (rank offset inside
the [hi | lo | hi] block, `split` >= the chunk's rank count).  Here the two launch functions are replaced by CPU stand-ins that do what one
kernel launch does on exactly the views it is handed (the kernels themselves are checked on the GPU: tests/test_gpu_r3_kernels.py), so the
slicing — operand rows, slab columns, transposed / strided gradient destinations — is checked against the oracle on the whole rank."""
import pytest
import torch

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd import ops
from oracle import ref_ops


def _down_launch(x, pmat, out, scale, mult, rows_per_batch, x_seg, M, p_lo, split, tmask, tmask_rows_per_batch):
    Rc = pmat.shape[0]
    assert Rc <= 64 and x_seg is None
    M = x.shape[0] if M is None else M
    t = x[:M].float() @ (pmat.float() + (p_lo.float() if p_lo is not None else 0)).t() * scale
    if mult is not None:
        t = t * mult[torch.arange(M) // rows_per_batch][:, None]
    if tmask is not None:  # the kernel indexes the mask by the rank INSIDE the launch: a chunk brings its own contiguous [rows, Rc] matrix
        assert tmask.is_contiguous() and tmask.shape[1] == Rc
        t = t * (tmask[torch.arange(M) // tmask_rows_per_batch] if tmask_rows_per_batch else tmask[:M])
    if not split:
        out[:M, :Rc] = t.to(out.dtype)
        return out
    assert split >= Rc
    hi = t.to(out.dtype)
    out[:M, :Rc] = hi
    out[:M, split:split + Rc] = (t - hi.float()).to(out.dtype)
    out[:M, 2 * split:2 * split + Rc] = hi
    return out


def _wgrad_launch(s, g, out, R, L, accumulate, g_seg, M, split, out_strides, transpose_out, second=None, defer=False):
    assert R <= 64 and g_seg is None
    M = s.shape[0] if M is None else M
    sv = (s[:M, :R].float() + s[:M, split:split + R].float()) if split else s[:M, :R].float()
    res = sv.t() @ g[:M].float()  # [R, L]
    sr, sl = out_strides if out_strides is not None else ((1, R) if transpose_out else (L, 1))
    flat = out.view(-1)
    idx = (torch.arange(R)[:, None] * sr + torch.arange(L)[None, :] * sl).reshape(-1)
    flat[idx] = (flat[idx] if accumulate else 0) + res.reshape(-1)
    return out


@pytest.mark.parametrize("R", [80, 128, 144])
def test_rank_chunks_write_one_slab_and_one_gradient(monkeypatch, R):
    monkeypatch.setattr(ops, "_lora_down_launch", _down_launch)
    monkeypatch.setattr(ops, "_lora_wgrad_launch", _wgrad_launch)
    g = torch.Generator().manual_seed(R)
    M, K, L = 300, 96, 40
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    p32 = torch.randn(R, K, generator=g) * K ** -0.5
    hi = p32.to(torch.bfloat16)
    lo = (p32 - hi.float()).to(torch.bfloat16)
    mult = torch.tensor([0.5, 1.5, -1.0])
    T, Tr = torch.zeros(M, 3 * R, dtype=torch.bfloat16), torch.zeros(M, 3 * R, dtype=torch.bfloat16)
    ops.lora_down(x, hi, T, scale=0.7, mult=mult, rows_per_batch=100, M=M, p_lo=lo, split=R)
    ref_ops.lora_down(x, hi, Tr, scale=0.7, mult=mult, rows_per_batch=100, M=M, p_lo=lo, split=R)
    assert torch.equal(T, Tr)
    Tp, Tpr = torch.zeros(M, R, dtype=torch.bfloat16), torch.zeros(M, R, dtype=torch.bfloat16)
    ops.lora_down(x, hi, Tp, scale=0.7, M=M)
    ref_ops.lora_down(x, hi, Tpr, scale=0.7, M=M)
    assert torch.equal(Tp, Tpr)
    gy = torch.randn(M, L, generator=g).to(torch.bfloat16)
    for transpose in (False, True):
        shape = (L, R) if transpose else (R, L)
        a, b = torch.ones(shape), torch.ones(shape)
        ops.lora_wgrad(Tr, gy, a, transpose_out=transpose, accumulate=True, M=M, split=R)
        ref_ops.lora_wgrad(Tr, gy, b, transpose_out=transpose, accumulate=True, M=M, split=R)
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5), transpose
    a, b = torch.zeros(R, L), torch.zeros(R, L)
    ops.lora_wgrad(Tpr, gy, a, M=M)
    ref_ops.lora_wgrad(Tpr, gy, b, M=M)
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-5)
    # dropout / rank_dropout masks above 64 ranks (round 5): every 64-rank launch takes its own contiguous slice of the mask
    for rpb, rows in ((0, M), (100, 3)):  # neuron dropout (one mask row per token) / rank dropout (one per sample)
        tm = (torch.rand(rows, R, generator=g) > 0.3).float() / 0.7
        Tm, Tmr = torch.zeros(M, 3 * R, dtype=torch.bfloat16), torch.zeros(M, 3 * R, dtype=torch.bfloat16)
        ops.lora_down(x, hi, Tm, scale=0.7, M=M, split=R, p_lo=lo, tmask=tm, tmask_rows_per_batch=rpb)
        ref_ops.lora_down(x, hi, Tmr, scale=0.7, M=M, split=R, p_lo=lo, tmask=tm, tmask_rows_per_batch=rpb)
        assert torch.equal(Tm, Tmr) and not torch.equal(Tm, T)
