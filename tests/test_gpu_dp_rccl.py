"""Data parallelism over RCCL on real devices — the GPU twin of tests/test_train_step_cpu.py's gloo test:
P ranks, each stepping its shard of the bucket batch, must end with bit-identical adapter arenas on every rank and must equal
ONE rank stepping the concatenated batch (up to fp32 summation order of the all-reduce).  Runs when >= 2 devices are visible
(the 8-GPU node of the scaling run); on a 1-GPU box the 2-rank case is skipped and a 1-rank RCCL group still walks the
collective path (two async all-reduce pieces overlapped with backward, stream-ordered wait, grad_scale 1/world)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_dir, shard, one_device=False, allreduce_dtype="fp32"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import datetime

    import torch.distributed as dist

    torch.cuda.set_device(0 if one_device else rank)
    dev = torch.device("cuda", 0 if one_device else rank)
    if one_device:  # every rank on device 0: RCCL refuses two ranks on one GPU, gloo reduces the device tensors through the host
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=300))
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from tests.test_gpu_e2e import _batch, _build

    ref, ref_net, nat, net = _build(dev=dev)
    step = FluxLoRATrainStep(nat, net, ops, lr=1e-3, max_grad_norm=0.5, ema_decay=0.9, process_group=dist.group.WORLD, allreduce_dtype=allreduce_dtype)
    per = 4 // world
    for k in range(2):
        lat, emb, pooled, noise, ts = _batch(4, dev=dev, seed=20 + k)
        sl = slice(rank * per, (rank + 1) * per) if shard else slice(0, 4)
        step.step(lat[sl], emb[sl], pooled[sl], noise=noise[sl], timesteps=ts[sl])
    torch.cuda.synchronize()
    torch.save({"p": net.arena_p.cpu(), "ema": net.arena_ema.cpu(), "loss": step.loss.cpu()}, os.path.join(out_dir, f"w{world}_r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(world, tmp_path, shard=True, one_device=False, allreduce_dtype="fp32"):
    import torch.multiprocessing as mp

    from tests.conftest import free_port

    mp.spawn(_worker, args=(world, free_port(), str(tmp_path), shard, one_device, allreduce_dtype), nprocs=world, join=True)
    return [torch.load(tmp_path / f"w{world}_r{r}.pt") for r in range(world)]


def _close_to_one_rank_on_the_whole_batch(p_dp):
    """Per-sample kernels are batch-independent (tests/test_gpu_fullsize.py), so DP(2) differs from one rank on the 4-sample batch only by
    the fp32 summation order of the gradients (two rank sums vs one sum; different row chunking inside aitk_lora_wgrad) in front of AdamW —
    whose first steps move every parameter by ~lr * sign(g), so single entries with a near-zero gradient may land a fraction of a step apart
    while the update as a whole agrees."""
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from tests.test_gpu_e2e import _batch, _build

    ref, ref_net, nat, net = _build()
    p0 = net.arena_p.detach().cpu().clone()
    step = FluxLoRATrainStep(nat, net, ops, lr=1e-3, max_grad_norm=0.5, ema_decay=0.9)
    for k in range(2):
        lat, emb, pooled, noise, ts = _batch(4, seed=20 + k)
        step.step(lat, emb, pooled, noise=noise, timesteps=ts)
    one = net.arena_p.cpu()
    rel = ((one - p_dp).norm() / (one - p0).norm()).item()
    worst = (one - p_dp).abs().max().item()
    print(f"DP(2) vs one rank on the whole batch: update rel diff {rel:.3e}, worst entry {worst:.3e} (two steps of lr 1e-3)")
    assert rel < 2e-2 and worst < 1e-3, (rel, worst)


def test_one_rank_rccl_group_equals_no_group(tmp_path):
    """world = 1: the all-reduce pieces are identities, so the arena after two steps equals a plain single-process run bit for bit."""
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from tests.test_gpu_e2e import _batch, _build

    got = _spawn(1, tmp_path, shard=False)[0]
    ref, ref_net, nat, net = _build()
    step = FluxLoRATrainStep(nat, net, ops, lr=1e-3, max_grad_norm=0.5, ema_decay=0.9)
    for k in range(2):
        lat, emb, pooled, noise, ts = _batch(4, seed=20 + k)
        step.step(lat, emb, pooled, noise=noise, timesteps=ts)
    assert torch.equal(net.arena_p.cpu(), got["p"]) and torch.equal(net.arena_ema.cpu(), got["ema"])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 visible devices (RCCL over xGMI)")
def test_dp2_rccl_equals_single_rank_on_concatenated_batch(tmp_path):
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from tests.test_gpu_e2e import _batch, _build

    r0, r1 = _spawn(2, tmp_path)
    assert torch.equal(r0["p"], r1["p"]) and torch.equal(r0["ema"], r1["ema"]), "replicas diverged"
    _close_to_one_rank_on_the_whole_batch(r0["p"])


def test_dp2_two_processes_on_one_device_equal_single_rank_on_concatenated_batch(tmp_path):
    """The data-parallel step on the HIP kernels with a real second rank, on a 1-GPU box: two processes share device 0 and all-reduce
    their gradient arenas over gloo (device tensors, reduced through the host).  Everything but the transport is the production path:
    per-rank shards, two asynchronous all-reduce pieces launched during backward, the wait in front of the optimizer kernel, 1 / world
    gradient scale, identical AdamW / EMA on every rank."""
    from ai_toolkit_amd import ops
    from ai_toolkit_amd.trainer import FluxLoRATrainStep
    from tests.test_gpu_e2e import _batch, _build

    r0, r1 = _spawn(2, tmp_path, one_device=True)
    assert torch.equal(r0["p"], r1["p"]) and torch.equal(r0["ema"], r1["ema"]), "replicas diverged"
    _close_to_one_rank_on_the_whole_batch(r0["p"])
    # each rank reports the loss of its own shard (the reference logs per process too): finite and different shards -> different numbers
    assert torch.isfinite(r0["loss"]).all() and torch.isfinite(r1["loss"]).all() and not torch.equal(r0["loss"], r1["loss"])


@pytest.mark.parametrize("n,off", [(1 << 20, 0), (100003, 3), (77, 5), (8 * 4096 + 1, 8)])
def test_grad_compress_expand_bf16_kernels(n, off):
    """aitk_grad_compress_bf16 / aitk_grad_expand_bf16 (transport format of the bf16 all-reduce) at aligned and unaligned arena offsets:
    exactly torch's round-to-nearest-even cast, neighbours untouched."""
    from ai_toolkit_amd import ops

    g = torch.randn(n + off + 16, device="cuda") * torch.logspace(-6, 3, n + off + 16, device="cuda")
    buf = torch.full((n + off + 16,), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.grad_compress_bf16(g[off:off + n], buf[off:off + n])
    assert torch.equal(buf[off:off + n], g[off:off + n].to(torch.bfloat16))
    assert float((buf[:off].float() - 7).abs().sum()) == 0 and float((buf[off + n:].float() - 7).abs().sum()) == 0
    back = torch.full_like(g, -3.0)
    ops.grad_expand_bf16(buf[off:off + n], back[off:off + n])
    assert torch.equal(back[off:off + n], buf[off:off + n].float())
    assert float((back[:off] + 3).abs().sum()) == 0 and float((back[off + n:] + 3).abs().sum()) == 0


def test_dp2_bf16_allreduce_on_the_hip_kernels_one_device(tmp_path):
    """allreduce_dtype="bf16" with a real second rank on the HIP kernels (two processes on device 0, gloo transport): replicas stay
    bit-identical and the two-step update stays within the bound stated for the fp32 transport plus the bf16 rounding of the gradient sum."""
    r0, r1 = _spawn(2, tmp_path, one_device=True, allreduce_dtype="bf16")
    assert torch.equal(r0["p"], r1["p"]) and torch.equal(r0["ema"], r1["ema"]), "replicas diverged"
    _close_to_one_rank_on_the_whole_batch(r0["p"])
