"""Round-5 ADVICE items (CPU): the skipped-step decision is taken identically on every data-parallel rank; a 3x3 convolution's implicit-GEMM
operands follow a load_state_dict; a parent load that carries nothing for an fp8-quantised Linear leaves it alone (and one that does, raises)."""
import os
import subprocess
import sys

import pytest
import torch

import ai_toolkit_amd  # noqa: F401
from tests.conftest import free_port


def test_conv3x3_operands_follow_load_state_dict_like_linear_weight_t():
    from ai_toolkit_amd.unet import Conv3x3

    g = torch.Generator().manual_seed(0)
    c = Conv3x3(8, 16, 1, torch.float32, "cpu")
    with torch.no_grad():
        c.weight.copy_(torch.randn(c.weight.shape, generator=g))
    c.prepare()
    wk0, wd0 = c.wk.clone(), c.wd.clone()
    new_w = torch.randn(c.weight.shape, generator=g)
    c.load_state_dict({"weight": new_w, "bias": torch.ones(16)})  # what the reference's merge_in does to org_module (network_mixins.py:452-462)
    assert not torch.equal(c.wk, wk0) and not torch.equal(c.wd, wd0)
    twin = Conv3x3(8, 16, 1, torch.float32, "cpu")
    with torch.no_grad():
        twin.weight.copy_(new_w)
        twin.bias.fill_(1.0)
    twin.prepare()
    assert torch.equal(c.wk, twin.wk) and torch.equal(c.wd, twin.wd) and torch.equal(c.bias_k, twin.bias_k)
    # a parent load that does not mention this layer leaves the operands alone (same objects)
    holder = torch.nn.Module()
    holder.conv, holder.other = c, torch.nn.Linear(2, 2)
    wk_obj = c.wk
    holder.load_state_dict({"other.weight": torch.zeros(2, 2), "other.bias": torch.zeros(2)}, strict=False)
    assert c.wk is wk_obj
    # an un-prepared layer is not prepared behind the caller's back
    fresh = Conv3x3(8, 16, 1, torch.float32, "cpu")
    fresh.load_state_dict({"weight": new_w, "bias": torch.ones(16)})
    assert fresh.wk is None


def test_fp8_linear_raises_only_when_its_own_weight_is_loaded():
    from tests.test_host_graph_cpu import build_pair

    _, _, nat, net = build_pair()
    nat.quantize_base_fp8()
    lin = nat.single_transformer_blocks[0].attn.to_q
    assert lin.qweight is not None
    sd = nat.state_dict()
    other = {k: v for k, v in sd.items() if k.startswith("x_embedder.")}
    nat.load_state_dict(other, strict=False)  # touches one bf16 embedder: the quantised layers are not in the dict -> no raise
    with pytest.raises(NotImplementedError, match="fp8"):
        nat.load_state_dict({"single_transformer_blocks.0.attn.to_q.weight": torch.zeros(lin.out_features, lin.in_features)}, strict=False)


_DP_CHILD = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import ai_toolkit_amd
from ai_toolkit_amd.trainer import FluxLoRATrainStep
from oracle import ref_ops
from tests.test_host_graph_cpu import build_pair
from tests.test_train_step_cpu import batch
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)
_, _, nat, net = build_pair(rank=4)
step = FluxLoRATrainStep(nat, net, ref_ops, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.0, process_group=dist.group.WORLD)
b = batch(2, seed=20 + rank)
step.step(*b[:3], noise=b[3], timesteps=b[4])                      # a healthy step on both ranks
# rank 1's only micro-batch: a non-finite loss over FINITE activations (infinite per-sample loss weights) -> gated, its dpred zeroed, so the
# arena it contributes is exactly zero (a NaN inside the activations would also put NaN into the gradients; that case is decided by the
# post-all-reduce norm check, which is the same on every rank by construction)
w = torch.full((2,), float("inf")) if rank == 1 else None
loss = step.step(b[0], b[1], b[2], noise=b[3], timesteps=b[4], loss_weight=w)
c = step.guard_counters()
torch.save({"p": net.arena_p.clone(), "m": net.arena_m.clone(), "v": net.arena_v.clone(), "c": c, "loss": float(loss)}, %(out)r + str(rank))
dist.barrier()
dist.destroy_process_group()
'''


def test_one_ranks_gated_batch_does_not_split_the_replicas(tmp_path):
    """the all-gated skip used a rank-LOCAL count.  Rank 1's only micro-batch has a non-finite loss, rank 0's is healthy: the
    all-reduced gradient is rank 0's / 2 on both ranks, so BOTH must apply it (the count is summed over ranks: 1 of 2 micro-batches gated) —
    before the fix rank 1 skipped and the replicas diverged for good."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _DP_CHILD % {"root": root, "out": str(tmp_path / "r")}
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    r0, r1 = torch.load(str(tmp_path / "r0")), torch.load(str(tmp_path / "r1"))
    assert r1["loss"] == 0.0 and r1["c"]["nonfinite_losses"] == 1 and r0["c"]["nonfinite_losses"] == 0
    for k in ("p", "m", "v"):
        assert torch.equal(r0[k], r1[k]), k
    assert r0["c"]["steps_applied"] == r1["c"]["steps_applied"] == 2 and r0["c"]["steps_skipped"] == r1["c"]["steps_skipped"] == 0
