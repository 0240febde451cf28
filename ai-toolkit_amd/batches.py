"""Producer of train-step batches from the reference's on-disk caches (host plumbing).

  * latent cache ........ `_latent_cache/<stem>_<hash>.safetensors`, tensor `latent` [C,h,w]  (toolkit/dataloader_mixins.py:1827-1842, 2076-2080)
  * text-embed cache .... PromptEmbeds.save: `text_embed` [1,512,4096], `pooled_embed` [1,768] (, `attention_mask`)
                          (toolkit/prompt_utils.py:119-141; loader 143-190); a batch concatenates along dim 0
                          (concat_prompt_embeds, toolkit/prompt_utils.py:259-315 — FLUX embeds are fixed-length, no padding)
  * batch = the items of ONE bucket batch (toolkit/data_transfer_object/data_loader.py:284-290 stacks `latents`)
Everything is staged to the GPU once per batch; the train step itself only sees device tensors.
"""
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch


@dataclass
class CachedItem:
    latent_path: str
    text_embed_path: str


@dataclass
class TrainBatch:
    latents: torch.Tensor        # [B, C, h, w]
    prompt_embeds: torch.Tensor  # [B, T, D]
    pooled_embeds: torch.Tensor  # [B, P]


def text_embedding_cache_path(image_path: str, caption: str, text_embedding_space_version: str, text_embedding_version: int = 1) -> str:
    """`_t_e_cache/<stem>_<hash>.safetensors` of the reference's text-embedding cache (toolkit/dataloader_mixins.py:2120-2163):
    md5 of the JSON of (caption, text_embedding_space_version, text_embedding_version), urlsafe-base64 without padding —
    the plain-caption case (no control image / first-frame conditioning keys)."""
    import base64
    import hashlib
    import json
    from collections import OrderedDict

    info = OrderedDict([("caption", caption), ("text_embedding_space_version", text_embedding_space_version),
                        ("text_embedding_version", text_embedding_version)])
    h = base64.urlsafe_b64encode(hashlib.md5(json.dumps(info, sort_keys=True).encode("utf-8")).digest()).decode("ascii").replace("=", "")
    stem = os.path.splitext(os.path.basename(image_path))[0]
    return os.path.join(os.path.dirname(image_path), "_t_e_cache", f"{stem}_{h}.safetensors")


def save_prompt_embeds(path: str, text_embed: torch.Tensor, pooled_embed: Optional[torch.Tensor] = None,
                       attention_mask: Optional[torch.Tensor] = None):
    from safetensors.torch import save_file

    sd = {"text_embed": text_embed.cpu().contiguous()}
    if pooled_embed is not None:
        sd["pooled_embed"] = pooled_embed.cpu().contiguous()
    if attention_mask is not None:
        sd["attention_mask"] = attention_mask.cpu().contiguous()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    save_file(sd, path)


def load_prompt_embeds(path: str):
    from safetensors.torch import load_file

    sd = load_file(path, device="cpu")
    if "text_embed" not in sd:
        raise ValueError(f"{path}: multi-encoder prompt embeds (text_embed_<i>) are not used by the FLUX path")
    return sd["text_embed"], sd.get("pooled_embed"), sd.get("attention_mask")


def load_batch(items: Sequence[CachedItem], device, dtype=torch.bfloat16) -> TrainBatch:
    from safetensors.torch import load_file

    lats, tes, pes = [], [], []
    for it in items:
        lats.append(load_file(it.latent_path)["latent"])
        te, pe, _ = load_prompt_embeds(it.text_embed_path)
        tes.append(te if te.dim() == 3 else te[None])
        pes.append(pe if pe.dim() == 2 else pe[None])
    shapes = {tuple(l.shape) for l in lats}
    if len(shapes) != 1:
        raise ValueError(f"a batch must come from one bucket, got latent shapes {sorted(shapes)}")
    return TrainBatch(torch.stack(lats).to(device, dtype), torch.cat(tes, 0).to(device, dtype), torch.cat(pes, 0).to(device, dtype))


class CachedDataset:
    """Epoch iterator: bucket batches (ai_toolkit_amd.buckets) -> rank shard -> TrainBatch on the device."""

    def __init__(self, items: List[CachedItem], latent_hw: Sequence[tuple], per_rank_batch: int, rank=0, world=1, seed=0):
        from . import buckets as bk

        self.items, self.rank, self.world, self.seed, self.per_rank = items, rank, world, seed, per_rank_batch
        self.buckets = {}
        for idx, (h, w) in enumerate(latent_hw):
            self.buckets.setdefault(f"{w}x{h}", bk.Bucket(w, h)).file_list_idx.append(idx)

    def epoch(self, epoch: int, device, dtype=torch.bfloat16):
        from . import buckets as bk

        gb = bk.epoch_batches(self.buckets, self.per_rank * self.world, self.seed, epoch)
        for idxs in bk.shard_batches(gb, self.rank, self.world):
            yield load_batch([self.items[i] for i in idxs], device, dtype)
