"""Per-kernel parity on the MI355X through the C ABI: hardware-layout probes, LoRA-fused GEMM (both staging modes,
ragged / segmented / every epilogue), LoRA skinny kernels, adaLN / gate / QK-norm-RoPE kernels, attention fwd+bwd at
ragged and full (S=4608) sizes, step kernels — each against fp32 torch math or the oracle's function of the same name."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ok(res):
    assert res.get("ok"), res


def test_hardware_layout_probes():
    from tools import gpu_check as g

    _ok(g.probe_mfma())
    _ok(g.probe_tr16())
    _ok(g.probe_glds())


@pytest.mark.parametrize("stage", [0, 1])
def test_gemm_lora_fused(stage):
    from ai_toolkit_amd import ops
    from tools import gpu_check as g

    _ok(g.gemm_case(256, 256, 128, stage=stage))
    _ok(g.gemm_case(200, 328, 192, stage=stage))  # ragged M/N, K tail inside a tile
    _ok(g.gemm_case(384, 512, 256, r=16, stage=stage))
    _ok(g.gemm_case(384, 256, 64, r=48, stage=stage))
    _ok(g.gemm_case(256, 384, 128, r=16, flags=ops.EPI_ACCUM, stage=stage))
    _ok(g.gemm_case(256, 384, 128, r=16, flags=ops.EPI_GELU, stage=stage))
    _ok(g.gemm_case(256, 384, 128, flags=ops.EPI_DGELU, stage=stage))
    _ok(g.gemm_case(256, 384, 128, r=16, flags=ops.EPI_GATE_RES, stage=stage))
    _ok(g.gemm_case(300, 256, 128, r=16, seg=True, stage=stage))
    _ok(g.gemm_case(1024, 3072, 3072, r=16, stage=stage))
    _ok(g.gemm_case(4608, 64, 3072, stage=stage))  # proj_out shape (N < tile)
    _ok(g.gemm_case(512, 3072, 64, stage=stage))  # x_embedder shape (K = one step)


@pytest.mark.parametrize("seed", [0, 1])
def test_gemm_persistent_8phase(seed):
    """gemm8.hip (stage_mode 4 forces it; default launches pick it for big problems): K tails and narrow LoRA slabs read
    zeros through the buffer descriptor, several output tiles per workgroup with cross-tile prefetch (tiles > CUs), odd
    K-tile counts, segmented rows, every epilogue through the LDS-transposed store path.  Two seeds = a small race screen."""
    from ai_toolkit_amd import ops
    from tools import gpu_check as g

    orig = ops.gemm_nt

    def forced(a, b, out, **kw):
        kw.pop("stage_mode", None)
        return orig(a, b, out, stage_mode=4, tile_mode=2, **kw)

    ops.gemm_nt = forced
    try:
        _ok(g.gemm_case(512, 512, 128, stage=1, seed=seed))
        _ok(g.gemm_case(300, 200, 192, r=16, stage=1, seed=seed))
        _ok(g.gemm_case(700, 1000, 80, r=16, stage=1, seed=seed))            # K tail (80 = 64 + 16)
        _ok(g.gemm_case(260, 520, 64, r=48, stage=1, seed=seed))             # one base K-tile + 48-wide slab
        _ok(g.gemm_case(512, 768, 128, r=16, flags=ops.EPI_GELU, stage=1, seed=seed))
        _ok(g.gemm_case(512, 768, 128, r=16, flags=ops.EPI_GATE_RES, stage=1, seed=seed))
        _ok(g.gemm_case(512, 768, 128, r=48, flags=ops.EPI_ACCUM, stage=1, seed=seed))
        _ok(g.gemm_case(600, 512, 128, r=16, seg=True, stage=1, seed=seed))
        _ok(g.gemm_case(8192, 4096, 256, r=16, flags=ops.EPI_GELU, stage=1, seed=seed))        # 512 tiles: 2 per workgroup
        _ok(g.gemm_case(6000, 5000, 128, r=32, flags=ops.EPI_DGELU | ops.EPI_ACCUM, stage=1, seed=seed))
        _ok(g.gemm_case(9000, 3072, 192, stage=1, seed=seed))                # odd K-tile count across tile boundaries
        _ok(g.gemm_case(1024, 3072, 12288, r=16, stage=1, seed=seed))
    finally:
        ops.gemm_nt = orig


def test_lora_skinny_kernels():
    from tools import gpu_check2 as g

    _ok(g.t_lora_down(4608, 3072, 16))
    _ok(g.t_lora_down(1000, 12288, 16))
    _ok(g.t_lora_down(512, 3072, 48, mult=True))
    _ok(g.t_lora_down(600, 1024, 64, seg=True))
    _ok(g.t_lora_down(2, 18432, 16))
    # long launches (>= 16384 rows): 64-row workgroups, every rank width, row counts that are no multiple of 64, segmented rows, per-sample multipliers
    _ok(g.t_lora_down(16400, 1024, 16))
    _ok(g.t_lora_down(16418, 1024, 32, mult=True))
    _ok(g.t_lora_down(16450, 1024, 48, seg=True))
    _ok(g.t_lora_down(16390, 1024, 64, mult=True, seg=True))
    _ok(g.t_lora_down_split(16400, 1024, 64, 16, mult=True))
    _ok(g.t_lora_down_split(16434, 1024, 48, 16, seg=True))
    _ok(g.t_lora_wgrad(4608, 16, 3072))
    _ok(g.t_lora_wgrad(1000, 16, 3072, transpose=True))
    _ok(g.t_lora_wgrad(700, 48, 1024, accumulate=True))
    _ok(g.t_lora_wgrad(300, 64, 520))
    _ok(g.t_lora_wgrad(2, 16, 18432, transpose=True))


def test_split_precision_skinny_kernels():
    """hi + lo shadows, [hi | lo | hi] slab layout: the adapter branch carries fp32-class precision on bf16 MFMA (the reference's
    adapter is fp32, toolkit/network_mixins.py:309)."""
    from tools import gpu_check2 as g

    _ok(g.t_lora_down_split(4608, 3072, 16, 16))
    _ok(g.t_lora_down_split(1000, 3072, 64, 16, mult=True))      # a q,k,v,proj_mlp group: four rank blocks of 16
    _ok(g.t_lora_down_split(600, 1024, 32, 32, seg=True))         # rank 32
    _ok(g.t_lora_down_split(512, 1040, 16, 16))                   # K % 32 != 0: 32x32x16 fallback kernel
    _ok(g.t_lora_down_split(2, 18432, 16, 16))
    _ok(g.t_lora_wgrad_split(4608, 16, 16, 3072))
    _ok(g.t_lora_wgrad_split(1000, 64, 16, 3072))
    _ok(g.t_lora_wgrad_split(700, 32, 32, 1024, transpose=True, accumulate=True))
    _ok(g.t_lora_wgrad_split(2, 16, 16, 18432, transpose=True))
    _ok(g.t_lora_down_mask(1000, 3072, 16, per_sample=True))     # rank_dropout: one mask row per sample
    _ok(g.t_lora_down_mask(1000, 3072, 16, per_sample=False))    # neuron dropout: one mask row per token
    # a few rows over a long contraction (adaLN adapters' backward): K slices across workgroups + finish pass (aitk_lora_down_ksplit, ABI 11)
    _ok(g.t_lora_down_mask(8, 18432, 16, per_sample=True))
    _ok(g.t_lora_down_mask(7, 9216, 16, per_sample=False))
    _ok(g.t_lora_down_split(32, 18432, 16, 16, mult=True))


def test_adapter_branch_matches_fp32_adapter_arithmetic():
    """North-star tolerance on LoRA quantities is 1e-3 relative; the split branch delivers ~1e-5 on the fp32 gradient outputs
    (single-bf16 shadows, the round-1 arithmetic: ~3e-3)."""
    from tools import gpu_check2 as g

    r = g.t_adapter_branch(4608, 3072, 3072, 16)
    _ok(r)
    assert r["dA_single_bf16"] > 1e-3 and r["dB_single_bf16"] > 1e-3, r
    _ok(g.t_adapter_branch(2048, 3072, 12288, 32))


def test_lora_down_ksplit_equals_the_single_workgroup_launch():
    """Same operands through both routes: equal up to the fp32 summation order (and the route is really taken: the entry point is called)."""
    import torch

    from ai_toolkit_amd import _capi, ops

    dev, bf = "cuda", torch.bfloat16
    g = torch.Generator().manual_seed(5)
    for M, K in ((7, 18432), (1, 9216), (32, 6144)):
        x = torch.randn(M, K, generator=g).to(bf).to(dev)
        p32 = (torch.randn(16, K, generator=g) * 0.05).to(dev)
        p_hi = p32.to(bf)
        p_lo = (p32 - p_hi.float()).to(bf)
        mult = torch.rand(M, generator=g).to(dev) + 0.5
        outs = []
        for on in (False, True):
            old, ops.KSPLIT = ops.KSPLIT, on
            try:
                out = torch.zeros(M, 48, dtype=bf, device=dev)
                ops.lora_down(x, p_hi, out, scale=0.5, mult=mult, rows_per_batch=1, p_lo=p_lo, split=16)
                outs.append(out[:, :16].float() + out[:, 16:32].float())
            finally:
                ops.KSPLIT = old
        torch.cuda.synchronize()
        want = 0.5 * mult[:, None] * (x.float() @ p32.t())
        e_route = ((outs[0] - outs[1]).norm() / outs[0].norm()).item()
        e_ref = ((outs[1] - want).norm() / want.norm()).item()
        assert e_route < 1e-5 and e_ref < 2e-3, (M, K, e_route, e_ref)
    assert _capi.lib().aitk_lora_down_ksplit_workspace_bytes(7, 16, 12) == 7 * 16 * 12 * 4


def test_norm_and_elementwise_kernels():
    from tools import gpu_check2 as g

    _ok(g.t_ln_mod(2, 200, 3072))
    _ok(g.t_ln_mod(1, 37, 1536))
    _ok(g.t_gate_bwd(2, 200, 3072))
    _ok(g.t_ln_mod(2, 1000, 3072))   # 63 row blocks per sample: the 16-group column-sum finish (colsum_finish16_kernel, from 32 row blocks up)
    _ok(g.t_gate_bwd(3, 4608, 3072))  # 288 row blocks per sample, as in the FLUX single blocks
    _ok(g.t_ln_mod(1, 520, 1536))
    _ok(g.t_qkv_post(2, 24, 100, 4))
    _ok(g.t_qkv_post(1, 7, 33, 3))  # heads not a multiple of the four a wave walks at a time
    _ok(g.t_small())


@pytest.mark.parametrize("shape", [(1, 2, 256), (2, 3, 200), (1, 2, 1111), (1, 4, 4608)])
def test_attention_fwd_bwd(shape):
    from tools import gpu_check2 as g

    _ok(g.t_attn(*shape))


def test_step_kernels_vs_oracle_ops():
    from tools import gpu_check3 as g

    _ok(g.t_gemv())
    _ok(g.t_noise_mse())
    _ok(g.t_adamw())
    _ok(g.t_shadows())


def test_masked_mse_kernel_vs_oracle():
    import torch

    from ai_toolkit_amd import ops
    from oracle import ref_ops

    g = torch.Generator().manual_seed(9)
    B, T, F = 3, 520, 64
    pred = torch.randn(B, T, F, generator=g).to(torch.bfloat16).cuda()
    tgt = torch.randn(B, T, F, generator=g).to(torch.bfloat16).cuda()
    mask = torch.rand(B, T, 4, generator=g).cuda()
    w = torch.tensor([1.0, 0.25, 3.0]).cuda()
    outs = []
    for o_ in (ops, ref_ops):
        dp = torch.empty_like(pred)
        lps, loss = torch.zeros(B, device="cuda"), torch.zeros(1, device="cuda")
        o_.mse_loss_grad(pred, tgt, dp, lps, loss, weight=w, mask=mask)
        outs.append((dp.float(), lps, loss))
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-5) and torch.allclose(outs[0][2], outs[1][2], rtol=1e-5)
    assert (outs[0][0] - outs[1][0]).abs().max() <= 2 ** -8 * outs[1][0].abs().max()


def test_adamw_ema_kernel_vs_torch_adamw_and_reference_ema_golden():
    """aitk_adamw_ema_step against the committed vectors of clip_grad_norm_ -> torch.optim.AdamW -> the reference's own
    toolkit/ema.py ExponentialMovingAverage.update() (tests/golden/optimizer_ema.safetensors, three steps)."""
    import os

    from safetensors.torch import load_file

    from ai_toolkit_amd import ops

    t = load_file(os.path.join(os.path.dirname(__file__), "golden", "optimizer_ema.safetensors"))
    p, ema = t["p0"].cuda(), t["p0"].cuda()
    m, v, norm = torch.zeros_like(p), torch.zeros_like(p), torch.zeros(1, device="cuda")
    for k in range(3):
        ops.adamw_ema_step(p, t["grads"][k].cuda().contiguous(), m, v, lr=3e-3, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.01,
                           step=k + 1, max_norm=1.0, ema=ema, ema_decay=0.9, norm_out=norm)
        assert abs(norm.item() - t["norms"][k].item()) <= 1e-5 * t["norms"][k].item()
    assert torch.allclose(p.cpu(), t["p3"], rtol=2e-5, atol=2e-7), (p.cpu() - t["p3"]).abs().max()
    assert torch.allclose(ema.cpu(), t["ema3"], rtol=2e-5, atol=2e-7), (ema.cpu() - t["ema3"]).abs().max()


def test_fused_linear_matches_eager_adapter_and_runtime_scale_updates_apply_on_device():
    """GPU leg of the reference's third adapter test (testing/test_lora_compile_scalars.py:94-150), through the C ABI in bf16."""
    from ai_toolkit_amd import ops
    from tests.test_adapter_scalars_cpu import run_fused_vs_eager

    run_fused_vs_eager(ops, torch.bfloat16, "cuda", rank=4)   # the reference's rank; lives in a zero-padded 16-wide rank block
    run_fused_vs_eager(ops, torch.bfloat16, "cuda", rank=8)


@pytest.mark.parametrize("M,L,R,opts", [(4608, 3072, 16, {}), (2304 + 17, 12288, 16, {"mult": True}), (2100, 3072, 32, {"tmask": True}),
                                        (4608, 18432, 16, {}), (2048, 3008, 16, {})])
def test_lora_bwd_fused_equals_lora_down_plus_lora_wgrad(M, L, R, opts):
    """aitk_lora_bwd_fused: dT = c (dY (P + P_lo)^T) and lora_up.weight.grad += dY^T T from ONE read of dY.  lora_up.weight.grad is the
    same code path as aitk_lora_wgrad (bit-identical); dT differs from aitk_lora_down's only in the fp32 summation order (column-tile
    partials in a fixed order instead of four K-quarters): compared on the fp32 value the [hi | lo | hi] slab carries."""
    from ai_toolkit_amd import ops

    g = torch.Generator().manual_seed(M + L)
    dy = (torch.randn(M, L, generator=g) * 0.5).to(torch.bfloat16).cuda()
    w = torch.randn(R, L, generator=g) * 0.05
    p_hi = w.to(torch.bfloat16)
    p_lo = (w - p_hi.float()).to(torch.bfloat16)
    p_hi, p_lo = p_hi.cuda(), p_lo.cuda()
    t = torch.randn(M, R, generator=g) * 0.3
    t_hi = t.to(torch.bfloat16)
    T = torch.cat((t_hi, (t - t_hi.float()).to(torch.bfloat16), t_hi), 1).cuda().contiguous()
    mult = torch.tensor([0.5, -1.5, 2.0], device="cuda") if opts.get("mult") else None
    rpb = (M + 2) // 3 if mult is not None else 0
    tmask = ((torch.rand(M, R, generator=g) > 0.3).float() / 0.7).cuda() if opts.get("tmask") else None
    kw = dict(scale=0.75, mult=mult, rows_per_batch=rpb, M=M, split=R, tmask=tmask, tmask_rows_per_batch=0)
    g0 = (torch.randn(L, R, generator=g) * 0.01).cuda()
    dT_a, gu_a = torch.zeros(M, 3 * R, dtype=torch.bfloat16, device="cuda"), g0.clone()
    ops.lora_down(dy, p_hi, dT_a, p_lo=p_lo, **kw)
    ops.lora_wgrad(T, dy, gu_a, transpose_out=True, accumulate=True, M=M, split=R)
    dT_b, gu_b = torch.zeros(M, 3 * R, dtype=torch.bfloat16, device="cuda"), g0.clone()
    ops.lora_bwd_fused(dy, T, p_hi, p_lo, dT_b, gu_b, **kw)
    assert torch.equal(gu_a, gu_b)
    va = dT_a[:, :R].float() + dT_a[:, R:2 * R].float()
    vb = dT_b[:, :R].float() + dT_b[:, R:2 * R].float()
    assert torch.equal(dT_b[:, :R], dT_b[:, 2 * R:])
    want = (dy.float() @ (p_hi.float() + p_lo.float()).t()) * 0.75
    if mult is not None:
        want = want * mult.repeat_interleave(rpb)[:M, None]
    if tmask is not None:
        want = want * tmask
    for v in (va, vb):
        assert ((v - want).norm() / want.norm()).item() < 2e-5
    assert ((va - vb).norm() / va.norm()).item() < 1e-5
    # deterministic: a second launch writes the same bits
    dT_c, gu_c = torch.zeros_like(dT_b), g0.clone()
    ops.lora_bwd_fused(dy, T, p_hi, p_lo, dT_c, gu_c, **kw)
    assert torch.equal(dT_b, dT_c) and torch.equal(gu_b, gu_c)
    # every form of the kernel (1 / 2 / 4 column tiles per workgroup; AITK_LORA_BWD_CT forces one): same dB bits, dT to summation order
    import os

    try:
        for ct in ("1", "2", "4"):
            os.environ["AITK_LORA_BWD_CT"] = ct
            dT_f, gu_f = torch.zeros_like(dT_b), g0.clone()
            ops.lora_bwd_fused(dy, T, p_hi, p_lo, dT_f, gu_f, **kw)
            assert torch.equal(gu_f, gu_a), ct
            vf = dT_f[:, :R].float() + dT_f[:, R:2 * R].float()
            assert ((vf - want).norm() / want.norm()).item() < 2e-5 and torch.equal(dT_f[:, :R], dT_f[:, 2 * R:]), ct
    finally:
        os.environ.pop("AITK_LORA_BWD_CT", None)
