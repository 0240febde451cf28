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


# ------------------------------------------------------------------------------------------------ Wan2.1 key formats
# The reference saves Wan adapters under the ORIGINAL Wan repository's module names and converts back on load
# (toolkit/models/wan21/wan21.py:726-730 -> toolkit/models/wan21/wan_lora_convert.py).  Same mapping, written as one
# table of path-component pairs (diffusers name, original name) applied on '.'-separated components.
_WAN_BLOCK_PARTS = (("attn1", "self_attn"), ("attn2", "cross_attn"))
_WAN_PROJ_PARTS = (("to_q", "q"), ("to_k", "k"), ("to_v", "v"), ("add_k_proj", "k_img"), ("add_v_proj", "v_img"))


def wan_lora_to_original(sd):
    """diffusers/PEFT keys `transformer.blocks.N.attn1.to_q.lora_A.weight` -> `diffusion_model.blocks.N.self_attn.q...`."""
    fwd = dict(_WAN_BLOCK_PARTS + _WAN_PROJ_PARTS)
    out = OrderedDict()
    for key, v in sd.items():
        parts = key.split(".")
        if parts[0] == "transformer":
            parts[0] = "diffusion_model"
        res = []
        i = 0
        while i < len(parts):
            p = parts[i]
            if p == "to_out" and i + 1 < len(parts) and parts[i + 1] == "0":
                res.append("o")
                i += 2
            elif p == "ffn" and parts[i + 1:i + 4] == ["net", "0", "proj"]:
                res += ["ffn", "0"]
                i += 4
            elif p == "ffn" and parts[i + 1:i + 3] == ["net", "2"]:
                res += ["ffn", "2"]
                i += 3
            else:
                res.append(fwd.get(p, p))
                i += 1
        out[".".join(res)] = v
    return out


def wan_lora_to_diffusers(sd):
    """Inverse of wan_lora_to_original (adapter files written by the reference or by the original Wan tooling)."""
    back = {b: a for a, b in _WAN_BLOCK_PARTS + _WAN_PROJ_PARTS}
    out = OrderedDict()
    for key, v in sd.items():
        parts = key.split(".")
        if parts[0] == "diffusion_model":
            parts[0] = "transformer"
        res = []
        i = 0
        while i < len(parts):
            p = parts[i]
            if p == "o":
                res += ["to_out", "0"]
            elif p == "ffn" and i + 1 < len(parts) and parts[i + 1] == "0":
                res += ["ffn", "net", "0", "proj"]
                i += 1
            elif p == "ffn" and i + 1 < len(parts) and parts[i + 1] == "2":
                res += ["ffn", "net", "2"]
                i += 1
            else:
                res.append(back.get(p, p))
            i += 1
        out[".".join(res)] = v
    return out
