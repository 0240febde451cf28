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
