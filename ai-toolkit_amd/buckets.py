"""Aspect-ratio bucket batching + deterministic data-parallel sharding (host logic).

Restates
  * get_bucket_for_image_size ........ toolkit/buckets.py:17-48
  * setup_buckets / crop geometry .... toolkit/dataloader_mixins.py:219-316 (central crop, scale so both sides cover)
  * build_batch_indices .............. toolkit/dataloader_mixins.py:198-211 (one batch = `batch_size` items of ONE bucket,
                                        a short tail is padded by repeating its own items)
and adds what the reference lacks (its dataloader is not sharded under DP): `shard_batches` gives each of
P ranks a disjoint, equal-size slice of every global bucket batch (global batch = P x per-rank batch), so DP(P) steps
exactly the batches a single rank with the P-times larger batch size would step.
"""
import math
import random
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple


def get_bucket_for_image_size(width: int, height: int, resolution: int = 512, divisibility: int = 8) -> Dict[str, int]:
    """Bucket (multiples of `divisibility`) for an image: shrink to at most resolution^2 pixels keeping the aspect ratio, then of the
    four floor / ceil roundings of the two sides that fit the pixel budget take the one whose area is closest to the target (first
    wins on ties, width-floor before width-ceil).  Same results as toolkit/buckets.py:17-48 (pinned by tests/golden/buckets.json)."""
    area, budget = width * height, resolution * resolution
    goal = min(area, budget)
    shrink = (goal / area) ** 0.5
    cells_w, cells_h = (width * shrink) / divisibility, (height * shrink) / divisibility
    best = None
    for round_w in (math.floor, math.ceil):
        for round_h in (math.floor, math.ceil):
            w, h = round_w(cells_w) * divisibility, round_h(cells_h) * divisibility
            if w <= 0 or h <= 0 or w * h > budget:
                continue
            miss = abs(w * h - goal)
            if best is None or miss < best[0]:
                best = (miss, w, h)
    if best is None:  # degenerate input: one cell per side at least
        best = (0, max(divisibility, math.floor(cells_w) * divisibility), max(divisibility, math.floor(cells_h) * divisibility))
    return {"width": best[1], "height": best[2]}


@dataclass
class CropPlan:
    """Resize + central-crop geometry of one file (FileItemDTO fields set by setup_buckets)."""
    scale_to_width: int
    scale_to_height: int
    crop_x: int
    crop_y: int
    crop_width: int
    crop_height: int

    @property
    def bucket_key(self) -> str:
        return f"{self.crop_width}x{self.crop_height}"


def plan_crop(width: int, height: int, resolution: int, bucket_tolerance: int = 64, scale: float = 1.0) -> CropPlan:
    width, height = int(width * scale), int(height * scale)
    b = get_bucket_for_image_size(width, height, resolution=resolution, divisibility=bucket_tolerance)
    max_scale = max(b["width"] / width, b["height"] / height)
    sw, sh = int(math.ceil(width * max_scale)), int(math.ceil(height * max_scale))
    return CropPlan(sw, sh, int((sw - b["width"]) / 2), int((sh - b["height"]) / 2), b["width"], b["height"])


@dataclass
class Bucket:
    width: int
    height: int
    file_list_idx: List[int] = field(default_factory=list)


def build_buckets(sizes: Sequence[Tuple[int, int]], resolution: int, bucket_tolerance: int = 64) -> Dict[str, Bucket]:
    buckets: Dict[str, Bucket] = {}
    for idx, (w, h) in enumerate(sizes):
        p = plan_crop(w, h, resolution, bucket_tolerance)
        buckets.setdefault(p.bucket_key, Bucket(p.crop_width, p.crop_height)).file_list_idx.append(idx)
    return buckets


def build_batch_indices(buckets: Dict[str, Bucket], batch_size: int) -> List[List[int]]:
    batches = []
    for bucket in buckets.values():
        ids = bucket.file_list_idx
        for start in range(0, len(ids), batch_size):
            batch = ids[start:min(start + batch_size, len(ids))]
            if 0 < len(batch) < batch_size:
                batch = batch + [batch[i % len(batch)] for i in range(batch_size - len(batch))]
            batches.append(batch)
    return batches


def epoch_batches(buckets: Dict[str, Bucket], batch_size: int, seed: int, epoch: int) -> List[List[int]]:
    """Shuffle inside buckets, build batches, shuffle batch order — from (seed, epoch) only, so every rank derives the
    same global order without communicating."""
    rng = random.Random(seed * 1_000_003 + epoch)
    shuffled = {k: Bucket(b.width, b.height, rng.sample(b.file_list_idx, len(b.file_list_idx))) for k, b in buckets.items()}
    batches = build_batch_indices(shuffled, batch_size)
    rng.shuffle(batches)
    return batches


def shard_batches(global_batches: List[List[int]], rank: int, world: int) -> List[List[int]]:
    """Rank r takes items [r*b, (r+1)*b) of every global batch (b = global / world): disjoint, same bucket, equal size."""
    out = []
    for gb in global_batches:
        assert len(gb) % world == 0, "global batch must be a multiple of the world size"
        b = len(gb) // world
        out.append(gb[rank * b:(rank + 1) * b])
    return out


def latent_shape(bucket: Bucket, vae_scale: int = 8, latent_channels: int = 16) -> Tuple[int, int, int]:
    return latent_channels, bucket.height // vae_scale, bucket.width // vae_scale
