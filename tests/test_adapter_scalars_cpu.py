"""Mirror of the reference's own adapter test, testing/test_lora_compile_scalars.py:26-92 (AdapterScaleTest): every adapter class
keeps `scale` as a python float plus a non-persistent, non-trainable `_runtime_scale` buffer, and `_set_runtime_scale` updates
both in place.  Same constructor calls (alpha as a bf16 tensor) on the fused path's module classes."""
import torch

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.lora import DoRAModule, LoKrModule, LoRAModule


class _Network:
    network_type = "lora"
    is_lorm = False
    is_active = True
    is_merged_in = False
    _multiplier = 1.0


def _linear(n=8):
    return torch.nn.Linear(n, n, bias=False)


def test_adapters_keep_float_metadata_and_nonpersistent_runtime_buffer():
    network = _Network()
    modules = [
        LoRAModule("lora_scale", _linear(), lora_dim=4, alpha=torch.tensor(8, dtype=torch.bfloat16), network=network),
        DoRAModule("dora_scale", _linear(), lora_dim=4, alpha=torch.tensor(8, dtype=torch.bfloat16), network=network),
        # 64 -> (8, 8) factors: the kron kernel wants factor dims in multiples of 8 (the reference test uses an 8x8 Linear)
        LoKrModule("lokr_scale", _linear(64), lora_dim=4, alpha=torch.tensor(4, dtype=torch.bfloat16), network=network),
    ]
    want = [2.0, 2.0, 1.0]  # alpha / rank; LoKr with both factors full forces alpha = rank (lokr.py:203-206)
    for module, w in zip(modules, want):
        assert type(module.scale) is float and module.scale == w, type(module).__name__
        assert module._runtime_scale.item() == module.scale
        assert "_runtime_scale" not in module.state_dict()
        assert not module._runtime_scale.requires_grad
        # the alpha buffer is persisted by LoRA / LoKr; the reference's DoRAModule has none (models/DoRA.py:67)
        assert ("alpha" in module.state_dict()) == (not isinstance(module, DoRAModule))


def test_set_runtime_scale_updates_in_place():
    module = LoRAModule("extract_scale", _linear(), lora_dim=4, alpha=torch.tensor(8, dtype=torch.bfloat16), network=_Network())
    runtime_scale = module._runtime_scale
    module._set_runtime_scale(1.0)
    assert module._runtime_scale is runtime_scale and module.scale == 1.0 and module._runtime_scale.item() == 1.0


# ---- third reference test (testing/test_lora_compile_scalars.py:94-150): a wrapped Linear executed by the fast path matches the eager
# adapter (forward + both weight gradients, rtol 2e-2 / atol 5e-2) and a runtime scale update applies without rebuilding anything
# (delta * 0.25 after scale 2 -> 0.5).  The reference compiles the module; here the "compiled" artefact is the fused kernel path.
def one_linear_model(n, ops, dtype, device):
    from ai_toolkit_amd.graph import FusedGraphBase, Linear, _Holder

    class OneLinear(FusedGraphBase):
        def __init__(self):
            super().__init__()
            self._init_graph(ops, dtype)
            blk = _Holder()
            blk.proj = Linear(n, n, bias=False, dtype=dtype, device=device)
            self.transformer_blocks = torch.nn.ModuleList([blk])

        def _token_linears(self):
            return [self.transformer_blocks[0].proj]

    return OneLinear()


def run_fused_vs_eager(ops, dtype, device, n=256, M=192, rank=4):
    from ai_toolkit_amd.lora import FusedLoRANetwork

    torch.manual_seed(0)
    model = one_linear_model(n, ops, dtype, device)
    lin = model.transformer_blocks[0].proj
    with torch.no_grad():
        lin.weight.copy_((torch.randn(n, n) / n ** 0.5).to(dtype))
    net = FusedLoRANetwork(model, lora_dim=rank, target_lin_modules=("OneLinear",), peft_format=True)
    mod = net.unet_loras[0]
    mod._set_runtime_scale(2.0)  # the reference test's alpha 8 / rank 4 (PEFT-format networks force alpha = rank at construction)
    with torch.no_grad():
        mod.lora_up.weight.normal_()
    net.apply_to()
    net.build_arena(device)
    net.refresh_shadows(ops)
    model.attach_network(net)
    model.prepare()
    value = torch.randn(M, n).to(dtype).to(device)
    W, A, Bu = lin.weight.float(), mod.lora_down.weight.float(), mod.lora_up.weight.float()

    def eager(scale):  # toolkit/network_mixins.py:304-342 on one Linear
        Ar, Br = A.detach().clone().requires_grad_(True), Bu.detach().clone().requires_grad_(True)
        out = (value.float() @ W.t() + (value.float() @ Ar.t()) @ Br.t() * scale).to(dtype)
        out.float().square().mean().backward()
        return out, Ar.grad, Br.grad

    def fused():
        out = torch.empty(M, n, dtype=dtype, device=device)
        with net:
            T = model._lin_fwd(lin, value, out, M=M, rows_per_batch=M, B=1)
            dy = (out.float() * (2.0 / out.numel())).to(dtype)
            dx = torch.empty_like(value)
            net.zero_grad_arena()
            model._lin_bwd(lin, dy, T, value, dx, M=M, rows_per_batch=M, B=1)
        return out

    want, g_down, g_up = eager(2.0)
    got = fused()
    torch.testing.assert_close(got, want, rtol=2e-2, atol=5e-2)
    torch.testing.assert_close(mod.lora_down.weight.grad, g_down, rtol=2e-2, atol=5e-2)
    torch.testing.assert_close(mod.lora_up.weight.grad, g_up, rtol=2e-2, atol=5e-2)
    base = (value.float() @ W.t()).to(dtype)
    original_delta = got.float() - base.float()
    mod._set_runtime_scale(0.5)
    updated = fused()
    torch.testing.assert_close(updated.float() - base.float(), original_delta * 0.25, rtol=2e-2, atol=5e-2)
    assert float(original_delta.abs().max()) > 0.5  # the adapter term is not lost in the tolerance


def test_fused_linear_matches_eager_adapter_and_runtime_scale_updates_apply():
    from oracle import ref_ops

    run_fused_vs_eager(ref_ops, torch.float32, "cpu")
