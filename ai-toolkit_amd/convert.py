"""LoRA file-format converters (host logic).

  * peft_to_kohya / kohya_to_peft ... FLUX transformer adapters between the PEFT layout this toolkit (and the reference,
    toolkit/network_mixins.py:607-624) saves — `transformer.<path>.lora_A.weight` [r,in] / `.lora_B.weight` [out,r], no
    alpha (alpha == rank) — and the kohya layout `lora_transformer_<path_with_underscores>.lora_down.weight /
    .lora_up.weight / .alpha` (prefix + naming: toolkit/lora_special.py:463-502; conversion restated from
    scripts/convert_lora_to_peft_format.py).
  * Both directions are exact renames; alpha is written as rank so scale = alpha/rank = 1 stays 1.
"""
from collections import OrderedDict

import torch

PEFT_PREFIX = "transformer."
KOHYA_PREFIX = "lora_transformer_"


def peft_to_kohya(sd):
    out = OrderedDict()
    for k, v in sd.items():
        if not k.startswith(PEFT_PREFIX):
            out[k] = v
            continue
        for tag, new in ((".lora_A.weight", ".lora_down.weight"), (".lora_B.weight", ".lora_up.weight")):
            if k.endswith(tag):
                base = KOHYA_PREFIX + k[len(PEFT_PREFIX):-len(tag)].replace(".", "_")
                out[base + new] = v
                if new == ".lora_down.weight":
                    out[base + ".alpha"] = torch.tensor(float(v.shape[0]), dtype=v.dtype)
                break
        else:
            out[k] = v
    return out


def kohya_to_peft(sd, module_paths):
    """`module_paths`: dotted module paths of the target model (e.g. from FluxTransformer2DModel.named_modules()); kohya
    names lose the dot/underscore distinction, so the mapping is resolved against the real module tree."""
    lookup = {p.replace(".", "_"): p for p in module_paths}
    out = OrderedDict()
    for k, v in sd.items():
        if not k.startswith(KOHYA_PREFIX):
            out[k] = v
            continue
        base, _, leaf = k[len(KOHYA_PREFIX):].partition(".")
        if leaf == "alpha":
            continue
        path = lookup.get(base)
        if path is None:
            raise KeyError(f"{k}: no module named like '{base}' in the target model")
        tag = {"lora_down.weight": ".lora_A.weight", "lora_up.weight": ".lora_B.weight"}[leaf]
        out[PEFT_PREFIX + path + tag] = v
    return out


def scale_for_alpha(sd):
    """kohya files may carry alpha != rank: fold alpha/rank into lora_up so the PEFT form (scale 1) is equivalent."""
    out = OrderedDict(sd)
    for k in list(sd.keys()):
        if k.endswith(".alpha"):
            base = k[: -len(".alpha")]
            down, up = sd.get(base + ".lora_down.weight"), sd.get(base + ".lora_up.weight")
            if down is not None and up is not None:
                s = float(sd[k]) / down.shape[0]
                if s != 1.0:
                    out[base + ".lora_up.weight"] = (up.float() * s).to(up.dtype)
                    out[k] = torch.tensor(float(down.shape[0]), dtype=sd[k].dtype)
    return out
