"""Deferred weight-gradient finishes (ops.wgrad_defer_begin / aitk_lora_wgrad_main + aitk_lora_wgrad_finish_multi, ABI 12): the finish passes of the lora_down /
adaLN weight gradients of the FLUX backward are collected and launched eight at a time.  Same sums in the same order: loss, every gradient and the adapter after AdamW
must be bit for bit what the two-launch calls give — eagerly, under hipGraph capture and with gradient accumulation — and the route must really be taken."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair():
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from tests.test_gpu_e2e import _build

    kw = dict(lr=1e-3, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99)
    _, _, nat_a, net_a = _build()
    _, _, nat_b, net_b = _build()
    return FluxLoRATrainStep(nat_a, net_a, ops, **kw), FluxLoRATrainStep(nat_b, net_b, ops, **kw)


def _with(on, fn):
    from ai_toolkit_amd import ops

    old, ops.WGRAD_DEFER = ops.WGRAD_DEFER, on
    try:
        return fn()
    finally:
        ops.WGRAD_DEFER = old


def test_deferred_finishes_give_the_two_launch_bits():
    from ai_toolkit_amd import _capi, ops
    from tests.test_gpu_e2e import _batch

    plain, deferred = _pair()
    for k, shp in enumerate([dict(B=2, Hl=16, Wl=12), dict(B=3, Hl=12, Wl=16), dict(B=1, Hl=16, Wl=16)]):
        lat, emb, pooled, noise, ts = _batch(seed=250 + k, **shp)
        l0 = _with(False, lambda: plain.step(lat, emb, pooled, noise=noise, timesteps=ts).clone())
        g0 = plain.network.arena_g.clone()
        mains, multis = [], []
        lib = _capi.lib()
        o_main, o_multi = lib.aitk_lora_wgrad_main, lib.aitk_lora_wgrad_finish_multi

        class Spy:
            def __init__(self, fn, log):
                self.fn, self.log = fn, log

            def __call__(self, *a):
                self.log.append(a[1] if self.fn is o_multi else 1)
                return self.fn(*a)

        lib.aitk_lora_wgrad_main, lib.aitk_lora_wgrad_finish_multi = Spy(o_main, mains), Spy(o_multi, multis)
        try:
            l1 = _with(True, lambda: deferred.step(lat, emb, pooled, noise=noise, timesteps=ts).clone())
        finally:
            lib.aitk_lora_wgrad_main, lib.aitk_lora_wgrad_finish_multi = o_main, o_multi
        assert ops._wdefer is None  # closed (and flushed) at the end of the backward pass
        assert len(mains) > 8 and sum(multis) == len(mains), (len(mains), multis)  # every producer got its finish ...
        assert len(multis) <= -(-len(mains) // 8) + 2 and max(multis) == 8, multis  # ... eight at a time
        assert torch.equal(l0, l1), (k, l0.item(), l1.item())
        assert torch.equal(g0, deferred.network.arena_g), k
        assert torch.equal(plain.network.arena_p, deferred.network.arena_p), k
        assert torch.equal(plain.network.arena_ema, deferred.network.arena_ema), k


def test_deferred_finishes_under_graph_capture_and_accumulation():
    from tests.test_gpu_e2e import _batch

    plain, graphed = _pair()
    for k in range(3):
        lat, emb, pooled, noise, ts = _batch(seed=270 + k, B=2, Hl=16, Wl=12)
        l0 = _with(False, lambda: plain.step(lat, emb, pooled, noise=noise, timesteps=ts).clone())
        g0 = plain.network.arena_g.clone()
        l1 = _with(True, lambda: graphed.step_graphed(latents=lat, prompt_embeds=emb, pooled_embeds=pooled, noise=noise, timesteps=ts).clone())
        assert torch.equal(l0, l1), k
        assert torch.equal(g0, graphed.network.arena_g), k
        assert torch.equal(plain.network.arena_p, graphed.network.arena_p), k
    a, b = _pair()
    mbs = []
    for k in range(2):
        lat, emb, pooled, noise, ts = _batch(seed=280 + k, B=2, Hl=16, Wl=12)
        mbs.append(dict(latents=lat, prompt_embeds=emb, pooled_embeds=pooled, noise=noise, timesteps=ts))
    l0 = _with(False, lambda: a.step_list(mbs).clone())
    l1 = _with(True, lambda: b.step_list(mbs).clone())
    assert torch.equal(l0, l1)
    assert torch.equal(a.network.arena_g, b.network.arena_g)
    assert torch.equal(a.network.arena_p, b.network.arena_p)


def test_finish_multi_refuses_two_jobs_on_one_output():
    import ctypes as C

    from ai_toolkit_amd import _capi

    arr = (_capi.LoraWgradArgs * 2)()
    buf = torch.zeros(64, device="cuda")
    for i in range(2):
        arr[i].M, arr[i].R, arr[i].L = 64, 16, 8
        arr[i].partial = C.c_void_p(buf.data_ptr())
        arr[i].out = C.c_void_p(buf.data_ptr())
    rc = _capi.lib().aitk_lora_wgrad_finish_multi(arr, 2, _capi.stream_ptr())
    assert rc != 0
