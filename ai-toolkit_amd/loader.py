"""Checkpoint loading for the fused models: a diffusers-format component directory (`<root>/transformer`, `<root>/unet`, `<root>/vae`) holds
either one `diffusion_pytorch_model.safetensors` or N shards `diffusion_pytorch_model-0000k-of-0000N.safetensors` plus
`diffusion_pytorch_model.safetensors.index.json` ({"weight_map": {parameter name: shard file}}).  The reference reaches these files through
`FluxTransformer2DModel.from_pretrained(transformer_path, subfolder=..., torch_dtype=dtype)`
(extensions_built_in/diffusion_models/flux_kontext/flux_kontext.py:84-104; Wan: toolkit/models/wan21/wan21.py:344-392) — the parameter names in
the files are diffusers' names, which are the names of our modules, so loading is a streamed copy: one shard is opened at a time
(safetensors `safe_open`, memory-mapped), each tensor goes to its parameter's device in the parameter's dtype, nothing is ever held twice on the
host (FLUX.1-dev: 23.8 GB of bf16 in three shards).  Resolution of a model path mirrors the reference's: `name_or_path` may be the pipeline
root (then `subfolder` is appended) or the component directory itself.
"""
import json
import os

import torch

WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
INDEX_NAME = WEIGHTS_NAME + ".index.json"


def resolve_component_dir(name_or_path, subfolder):
    """`<name_or_path>/<subfolder>` when that exists (pipeline root, flux_kontext.py:84-88), else name_or_path itself when it already is the
    component directory; raises FileNotFoundError otherwise (there is no hub download here: no network on the training boxes)."""
    for cand in (os.path.join(name_or_path, subfolder) if subfolder else None, name_or_path):
        if cand and os.path.isdir(cand) and (os.path.exists(os.path.join(cand, INDEX_NAME)) or os.path.exists(os.path.join(cand, WEIGHTS_NAME))
                                             or any(f.endswith(".safetensors") for f in os.listdir(cand))):
            return cand
    raise FileNotFoundError(f"no diffusers-format '{subfolder}' weights under {name_or_path!r} (expected {WEIGHTS_NAME} or {INDEX_NAME})")


def shard_map(component_dir):
    """{shard path: [tensor names]} in file order.  Sharded: from the index; single file: every key of it."""
    idx = os.path.join(component_dir, INDEX_NAME)
    if os.path.exists(idx):
        with open(idx) as fh:
            wm = json.load(fh)["weight_map"]
        out = {}
        for name, fname in wm.items():
            out.setdefault(os.path.join(component_dir, fname), []).append(name)
        return out
    single = os.path.join(component_dir, WEIGHTS_NAME)
    if not os.path.exists(single):
        cands = sorted(f for f in os.listdir(component_dir) if f.endswith(".safetensors"))
        if len(cands) != 1:
            raise FileNotFoundError(f"{component_dir}: expected {WEIGHTS_NAME}, an index, or exactly one .safetensors file; found {cands}")
        single = os.path.join(component_dir, cands[0])
    from safetensors import safe_open

    with safe_open(single, framework="pt", device="cpu") as f:
        return {single: list(f.keys())}


@torch.no_grad()
def load_component(model, component_dir, *, strict=True, rename=None):
    """Stream every tensor of the component into `model`'s parameters / buffers (names = diffusers' names; `rename`: optional
    callable file-key -> model-key or None to skip).  Returns (missing, unexpected) like nn.Module.load_state_dict; raises on shape
    mismatch always and on missing / unexpected names when strict."""
    from safetensors import safe_open

    target = dict(model.named_parameters())
    target.update(dict(model.named_buffers()))
    seen, unexpected = set(), []
    for path, names in shard_map(component_dir).items():
        with safe_open(path, framework="pt", device="cpu") as f:
            for name in names:
                key = rename(name) if rename is not None else name
                if key is None:
                    continue
                dst = target.get(key)
                if dst is None:
                    unexpected.append(name)
                    continue
                src = f.get_tensor(name)
                if src.dim() == 4 and dst.dim() == 2 and tuple(src.shape[2:]) == (1, 1) and tuple(src.shape[:2]) == tuple(dst.shape):
                    src = src[:, :, 0, 0]  # 1x1 Conv2d weights of the diffusers file ([out, in, 1, 1]) held as the [out, in] matrix of a token GEMM (unet.Conv1x1)
                if tuple(src.shape) != tuple(dst.shape):
                    raise ValueError(f"{name}: checkpoint shape {tuple(src.shape)} != model shape {tuple(dst.shape)}")
                dst.copy_(src.to(dst.dtype) if src.dtype != dst.dtype and not dst.is_cuda else src, non_blocking=False)
                seen.add(key)
    missing = [k for k in target if k not in seen]
    if strict and (missing or unexpected):
        raise KeyError(f"load_component({component_dir}): missing {missing[:5]}{'...' if len(missing) > 5 else ''} "
                       f"unexpected {unexpected[:5]}{'...' if len(unexpected) > 5 else ''}")
    return missing, unexpected


@torch.no_grad()
def save_component(model, component_dir, *, max_shard_bytes=10 * 2 ** 30, dtype=None):
    """The inverse (BaseModel.save_model -> save_pretrained(safe_serialization=True), toolkit/models/base_model.py:350-360): parameters in
    registration order, greedy shards of at most `max_shard_bytes`, index json when more than one shard."""
    from safetensors.torch import save_file

    os.makedirs(component_dir, exist_ok=True)
    sd = {k: (v.detach().to("cpu", dtype) if dtype is not None else v.detach().cpu()).contiguous() for k, v in model.state_dict().items()}
    shards, cur, cur_bytes = [], {}, 0
    for k, v in sd.items():
        nb = v.numel() * v.element_size()
        if cur and cur_bytes + nb > max_shard_bytes:
            shards.append(cur)
            cur, cur_bytes = {}, 0
        cur[k] = v
        cur_bytes += nb
    if cur:
        shards.append(cur)
    if len(shards) == 1:
        save_file(shards[0], os.path.join(component_dir, WEIGHTS_NAME), metadata={"format": "pt"})
        return [WEIGHTS_NAME]
    names, wm, total = [], {}, 0
    for i, sh in enumerate(shards):
        fname = f"diffusion_pytorch_model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(sh, os.path.join(component_dir, fname), metadata={"format": "pt"})
        names.append(fname)
        for k, v in sh.items():
            wm[k] = fname
            total += v.numel() * v.element_size()
    with open(os.path.join(component_dir, INDEX_NAME), "w") as fh:
        json.dump({"metadata": {"total_size": total}, "weight_map": wm}, fh, indent=2)
    return names
