"""Bucket batching host logic vs vectors produced by the reference's own toolkit/buckets.py (golden) + DP sharding
invariants."""
import json
import os

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd import buckets as bk

G = os.path.join(os.path.dirname(__file__), "golden", "buckets.json")


def test_bucket_resolution_matches_reference_function():
    gold = json.load(open(G))
    for w, h, res, div, ew, eh in gold["cases"]:
        got = bk.get_bucket_for_image_size(w, h, resolution=res, divisibility=div)
        assert (got["width"], got["height"]) == (ew, eh), (w, h, res, div, got)


def test_baseline_bucket_mix_is_reachable_and_divisible():
    # BASELINE.md bucket mix for the bucketed FLUX run: all <= 1024^2 px and /64 divisible
    for w, h in ((1024, 1024), (832, 1216), (1216, 832), (896, 1152), (1152, 896)):
        p = bk.plan_crop(w * 2, h * 2, resolution=1024, bucket_tolerance=64)
        assert (p.crop_width, p.crop_height) == (w, h)
        assert p.scale_to_width >= w and p.scale_to_height >= h and p.crop_x >= 0 and p.crop_y >= 0


def test_batches_are_single_bucket_padded_by_repetition_and_sharded_disjointly():
    sizes = [(2048, 2048)] * 5 + [(1664, 2432)] * 3 + [(2432, 1664)] * 9
    buckets = bk.build_buckets(sizes, resolution=1024, bucket_tolerance=64)
    assert sorted(buckets) == ["1024x1024", "1216x832", "832x1216"]
    world, per_rank = 2, 2
    gb = bk.epoch_batches(buckets, batch_size=world * per_rank, seed=7, epoch=0)
    key_of = {}
    for k, b in buckets.items():
        for i in b.file_list_idx:
            key_of[i] = k
    seen = set()
    for batch in gb:
        assert len(batch) == world * per_rank
        assert len({key_of[i] for i in batch}) == 1  # one resolution per batch (dataloader_mixins.py:198-211)
        seen.update(batch)
    assert seen == set(range(len(sizes)))  # every file appears (short tails are padded by repetition, not dropped)
    shards = [bk.shard_batches(gb, r, world) for r in range(world)]
    for step in range(len(gb)):
        parts = [shards[r][step] for r in range(world)]
        assert sum(parts, []) == gb[step] and all(len(p) == per_rank for p in parts)
    # deterministic from (seed, epoch); different epochs reshuffle
    assert bk.epoch_batches(buckets, 4, 7, 0) == gb and bk.epoch_batches(buckets, 4, 7, 1) != gb
    assert bk.latent_shape(buckets["832x1216"]) == (16, 152, 104)


def test_cached_dataset_yields_single_bucket_batches_in_reference_cache_formats(tmp_path):
    import torch
    from ai_toolkit_amd import batches as bt
    from ai_toolkit_amd import vae as nvae

    items, hw = [], []
    for i in range(6):
        h, w = (16, 16) if i % 2 == 0 else (12, 20)
        lp = str(tmp_path / "_latent_cache" / f"img{i}_x.safetensors")
        nvae.save_latent_cache(lp, torch.full((16, h, w), float(i)).to(torch.bfloat16))
        tp = str(tmp_path / "_t_e_cache" / f"img{i}.safetensors")
        bt.save_prompt_embeds(tp, torch.full((1, 8, 32), float(i)), torch.full((1, 4), float(i)))
        items.append(bt.CachedItem(lp, tp))
        hw.append((h, w))
    seen = []
    for r in range(2):
        ds = bt.CachedDataset(items, hw, per_rank_batch=1, rank=r, world=2, seed=3)
        for b in ds.epoch(0, "cpu", torch.float32):
            assert b.latents.shape[0] == 1 and b.prompt_embeds.shape == (1, 8, 32) and b.pooled_embeds.shape == (1, 4)
            i = int(b.latents[0, 0, 0, 0])
            assert float(b.prompt_embeds[0, 0, 0]) == i and float(b.pooled_embeds[0, 0]) == i  # latents / embeds stay paired
            seen.append(i)
    assert sorted(set(seen)) == list(range(6))
