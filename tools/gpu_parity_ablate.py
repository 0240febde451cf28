"""Where does the difference between the HIP path and the rounding-matched oracle come from?  (tests/test_gpu_parity_r2.py)

The fused graph runs with a HYBRID kernel table: every op from the oracle's plain-torch table (fp32 math, one rounding per output)
except ONE family taken from the HIP library.  For each family: adapter-gradient relative error against the all-oracle run on
identical inputs (tiny FLUX, B = 2).  The family with the largest number is the kernel whose internal arithmetic (bf16 P in flash
attention, fast GELU, summation order ...) contributes most.  Prints one JSON line per family; not part of the product path."""
import json
import math
import sys

import torch

sys.path.insert(0, ".")
bf = torch.bfloat16

FAMILIES = {
    "gemm_nt": ["gemm_nt"],
    "attn_fwd": ["attn_fwd"],
    "attn_bwd": ["attn_bwd"],
    "attn_fwd+bwd": ["attn_fwd", "attn_bwd"],
    "lora_down+wgrad": ["lora_down", "lora_wgrad"],
    "ln_mod": ["ln_mod_fwd", "ln_mod_bwd"],
    "qkv_post": ["qkv_post_fwd", "qkv_post_bwd"],
    "gate_bwd": ["gate_bwd"],
    "gemv+ew+temb": ["gemv_nt", "ew", "timestep_embed"],
    "noise_mse": ["flow_noise_pack", "mse_loss_grad"],
}


class Hybrid:
    def __init__(self, base, fast, names):
        self._base, self._fast, self._names = base, fast, set(names)

    def __getattr__(self, k):
        return getattr(self._fast if k in self._names else self._base, k)


def main():
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.flux import FluxTransformer2DModel
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from oracle import ref_ops
    from tests.test_gpu_e2e import CFG, _batch, _build

    ref, ref_net, nat, net = _build()
    lat, emb, pooled, noise, ts = _batch(2)
    kw = dict(lr=0.0, weight_decay=0.0, max_grad_norm=0.0)

    def run(table):
        m = FluxTransformer2DModel(**CFG, dtype=bf, device="cuda", ops=table)
        m.load_state_dict(nat.state_dict(), strict=True)
        n = FusedLoRANetwork(m, lora_dim=16)
        with torch.no_grad():
            for a, b in zip(n.unet_loras, net.unet_loras):
                a.lora_down.weight.copy_(b.lora_down.weight.detach().cpu())
                a.lora_up.weight.copy_(b.lora_up.weight.detach().cpu())
        n.apply_to()
        n.build_arena("cuda", groups=m.lora_groups())  # bf16 split (hi + lo) shadows, as the product
        n.refresh_shadows(table)
        m.attach_network(n)
        m.prepare()
        loss = FluxLoRATrainStep(m, n, table, **kw).step(lat, emb, pooled, noise=noise, timesteps=ts).item()
        g = torch.cat([p.grad.reshape(-1) for x in n.unet_loras for p in (x.lora_down.weight, x.lora_up.weight)]).clone()
        return loss, g

    l0, g0 = run(ref_ops)
    lh, gh = run(ops)
    print(json.dumps({"family": "ALL (HIP path)", "loss": lh, "loss_oracle_table": l0,
                      "grad_rel_err_vs_oracle_table": ((gh - g0).norm() / g0.norm()).item()}))
    for name, fns in FAMILIES.items():
        try:
            l, g = run(Hybrid(ref_ops, ops, fns))
            print(json.dumps({"family": name, "loss_rel": abs(l - l0) / abs(l0), "grad_rel_err_vs_oracle_table": ((g - g0).norm() / g0.norm()).item()}))
        except Exception as ex:  # a family whose HIP entry needs state the oracle table does not provide
            print(json.dumps({"family": name, "error": f"{type(ex).__name__}: {ex}"[:160]}))
    assert math.isfinite(lh)


if __name__ == "__main__":
    main()
