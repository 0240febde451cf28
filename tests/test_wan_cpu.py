"""Wan2.1 host-logic parity on CPU: the explicit forward/backward graph of ai_toolkit_amd.wan, driven by the oracle's
plain-torch kernel table in fp32, must reproduce autograd of oracle/wan_ref.py + the oracle LoRA layer."""
import torch

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd.lora import FusedLoRANetwork
from ai_toolkit_amd.trainer import WanLoRATrainStep
from ai_toolkit_amd.wan import WanTransformer3DModel
from oracle import lora_ref, ref_ops, wan_ref

CFG = dict(num_attention_heads=2, attention_head_dim=128, in_channels=16, out_channels=16, text_dim=48, freq_dim=256,
           ffn_dim=320, num_layers=3)


def build_pair(rank=8, multiplier=1.0, grouped=True):
    torch.manual_seed(0)
    ref = wan_ref.WanTransformer3DModel(**CFG)
    wan_ref.init_synthetic_(ref, seed=99, std=0.05)
    nat = WanTransformer3DModel(**CFG, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    ref_net = lora_ref.RefLoRANetwork(ref, rank, multiplier, target=("WanTransformer3DModel",), block_names=("blocks",))
    net = FusedLoRANetwork(nat, lora_dim=rank, multiplier=multiplier, target_lin_modules=("WanTransformer3DModel",),
                           transformer_block_names=["blocks"], base_model_version="wan_2.1")
    assert [m.lora_name for m in net.unet_loras] == [m.lora_name for m in ref_net.unet_loras]
    assert len(net.unet_loras) == 10 * CFG["num_layers"]
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            b.lora_up.weight.copy_(torch.randn(b.lora_up.weight.shape, generator=g) * 0.05)
            a.lora_down.weight.copy_(b.lora_down.weight)
            a.lora_up.weight.copy_(b.lora_up.weight)
    ref_net.apply_to()
    net.apply_to()
    net.build_arena("cpu", groups=nat.lora_groups() if grouped else None)
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    nat.prepare()
    return ref, ref_net, nat, net


def inputs(B=2, Fr=3, Hl=8, Wl=4, n_txt=5, seed=3):
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(B, 16, Fr, Hl, Wl, generator=g)
    txt = torch.randn(B, n_txt, CFG["text_dim"], generator=g)
    t = torch.tensor([310.0, 845.0][:B])
    return lat, txt, t


def _grad_check(net, ref_net, tol=2e-4):
    worst = 0.0
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        for x, y, nm in ((a.lora_down.weight.grad, b.lora_down.weight.grad, "down"), (a.lora_up.weight.grad, b.lora_up.weight.grad, "up")):
            err = ((x - y).norm() / (y.norm() + 1e-12)).item()
            worst = max(worst, err)
            assert err < tol, (a.lora_name, nm, err)
    assert worst > 0


def test_pack_helpers_match_oracle_layouts():
    lat, _, _ = inputs()
    tok = WanTransformer3DModel.pack_tokens(lat)
    assert torch.equal(tok, wan_ref.pack_video_latents(lat))
    assert torch.equal(WanTransformer3DModel.unpack_tokens(tok, (3, 4, 2)), lat)


def test_forward_and_lora_grads_match_oracle_autograd():
    ref, ref_net, nat, net = build_pair()
    lat, txt, t = inputs()
    with ref_net:
        pred_ref = ref(lat, t, txt)
        w5 = torch.randn(pred_ref.shape, generator=torch.Generator().manual_seed(11))
        (pred_ref * w5).sum().backward()
    grid = (3, 4, 2)
    with net:
        pred = nat.forward_native(nat.pack_tokens(lat), t, txt, grid)
        want = nat.pack_tokens(pred_ref.detach())  # (c, ph, pw) order
        assert torch.allclose(pred, want, rtol=1e-4, atol=2e-5), (pred - want).abs().max()
        net.zero_grad_arena()
        nat.backward_native(nat.pack_tokens(w5))
    _grad_check(net, ref_net)
    # the module-style call returns the reference's 5-D layout
    with torch.no_grad(), net:
        (p5,) = nat(lat, t, txt)
    assert torch.allclose(p5, pred_ref, rtol=1e-4, atol=2e-5)


def test_ungrouped_layout_and_per_sample_multiplier():
    ref, ref_net, nat, net = build_pair(grouped=False)
    lat, txt, t = inputs()
    net.multiplier = [0.5, -1.5]
    ref_net.torch_multiplier = torch.tensor([0.5, -1.5])
    with ref_net:
        pr = ref(lat, t, txt)
        pr.square().sum().backward()
    with net:
        pn = nat.forward_native(nat.pack_tokens(lat), t, txt, (3, 4, 2))
        assert torch.allclose(pn, nat.pack_tokens(pr.detach()), rtol=1e-4, atol=2e-5)
        net.zero_grad_arena()
        nat.backward_native((2 * pn).detach())
    _grad_check(net, ref_net)


def test_wan_train_step_matches_oracle_training():
    """Three optimizer steps: noise mix -> model -> MSE(noise - latents) -> backward -> clip -> AdamW, vs plain autograd."""
    ref, ref_net, nat, net = build_pair(rank=4)
    step = WanLoRATrainStep(nat, net, ref_ops, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    params = [p for m in ref_net.unet_loras for p in (m.lora_down.weight, m.lora_up.weight)]
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.01, eps=1e-6)
    for it in range(3):
        lat, txt, t = inputs(seed=20 + it)
        noise = torch.randn(lat.shape, generator=torch.Generator().manual_seed(50 + it))
        loss = step.step(lat, txt, noise=noise, timesteps=t)
        tt = (t / 1000).view(-1, 1, 1, 1, 1)
        noisy = (1 - tt) * lat + tt * noise
        opt.zero_grad()
        with ref_net:
            pred = ref(noisy, t, txt)
            loss_ref = (pred - (noise - lat)).pow(2).mean()
            loss_ref.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        assert abs(loss.item() - loss_ref.item()) < 2e-4 * max(1.0, abs(loss_ref.item())), (it, loss.item(), loss_ref.item())
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        assert torch.allclose(a.lora_up.weight, b.lora_up.weight, rtol=2e-3, atol=2e-6), a.lora_name
        assert torch.allclose(a.lora_down.weight, b.lora_down.weight, rtol=2e-3, atol=2e-6), a.lora_name


def test_wan_lora_state_dict_keys_and_original_format_round_trip():
    from ai_toolkit_amd import convert

    ref, ref_net, nat, net = build_pair(rank=4)
    sd = net.get_state_dict(dtype=torch.float32)
    assert "transformer.blocks.0.attn1.to_q.lora_A.weight" in sd and "transformer.blocks.2.ffn.net.2.lora_B.weight" in sd
    orig = convert.wan_lora_to_original(sd)
    assert "diffusion_model.blocks.0.self_attn.q.lora_A.weight" in orig
    assert "diffusion_model.blocks.1.cross_attn.o.lora_B.weight" in orig
    assert "diffusion_model.blocks.2.ffn.0.lora_A.weight" in orig
    back = convert.wan_lora_to_diffusers(orig)
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)


def test_wan_dora_network_matches_oracle_autograd():
    """network_type='dora' through the Wan graph (cross-attention k/v adapters take the weight-gradient-only path)."""
    torch.manual_seed(0)
    ref = wan_ref.WanTransformer3DModel(**CFG)
    wan_ref.init_synthetic_(ref, seed=99, std=0.05)
    nat = WanTransformer3DModel(**CFG, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    torch.manual_seed(3)
    ref_net = lora_ref.RefLoRANetwork(ref, 8, target=("WanTransformer3DModel",), block_names=("blocks",), network_type="dora")
    torch.manual_seed(3)
    net = FusedLoRANetwork(nat, lora_dim=8, target_lin_modules=("WanTransformer3DModel",), transformer_block_names=["blocks"],
                           base_model_version="wan_2.1", network_type="dora")
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            assert torch.equal(a.lora_down.weight, b.lora_down.weight)
            up = torch.randn(b.lora_up.weight.shape, generator=g) * 0.05
            mg = b.magnitude * (1 + 0.05 * torch.randn(b.magnitude.shape, generator=g))
            for m in (a, b):
                m.lora_up.weight.copy_(up)
                m.magnitude.copy_(mg)
    ref_net.apply_to()
    net.apply_to()
    net.build_arena("cpu", groups=nat.lora_groups())
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    nat.prepare()
    lat, txt, t = inputs()
    with ref_net:
        pred_ref = ref(lat, t, txt)
        w5 = torch.randn(pred_ref.shape, generator=torch.Generator().manual_seed(11))
        (pred_ref * w5).sum().backward()
    with net:
        pred = nat.forward_native(nat.pack_tokens(lat), t, txt, (3, 4, 2))
        assert torch.allclose(pred, nat.pack_tokens(pred_ref.detach()), rtol=2e-4, atol=2e-5)
        net.zero_grad_arena()
        nat.backward_native(nat.pack_tokens(w5))
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        for x, y, nm in ((a.lora_down.weight.grad, b.lora_down.weight.grad, "down"), (a.lora_up.weight.grad, b.lora_up.weight.grad, "up"),
                         (a.magnitude.grad, b.magnitude.grad, "magnitude")):
            err = ((x - y).norm() / (y.norm() + 1e-12)).item()
            assert err < 5e-4, (a.lora_name, nm, err)


def test_wan_lokr_network_matches_oracle_autograd():
    """network_type='lokr' (full Kronecker factors) through the Wan graph, incl. the weight-gradient-only cross-attention k/v."""
    torch.manual_seed(0)
    cfg = dict(CFG, ffn_dim=384)  # factor pairs must be multiples of 8 on the kernel path: 384 -> (16, 24); real Wan: 8960 -> (80, 112)
    ref = wan_ref.WanTransformer3DModel(**cfg)
    wan_ref.init_synthetic_(ref, seed=99, std=0.05)
    nat = WanTransformer3DModel(**cfg, dtype=torch.float32, device="cpu", ops=ref_ops)
    nat.load_state_dict(ref.state_dict(), strict=True)
    big = 9999999999
    torch.manual_seed(3)
    ref_net = lora_ref.RefLoRANetwork(ref, big, target=("WanTransformer3DModel",), block_names=("blocks",), network_type="lokr")
    torch.manual_seed(3)
    net = FusedLoRANetwork(nat, lora_dim=big, target_lin_modules=("WanTransformer3DModel",), transformer_block_names=["blocks"],
                           base_model_version="wan_2.1", network_type="lokr")
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for a, b in zip(net.unet_loras, ref_net.unet_loras):
            assert torch.equal(a.lokr_w1, b.lokr_w1)
            w2 = torch.randn(b.lokr_w2.shape, generator=g) * 0.05
            a.lokr_w2.copy_(w2)
            b.lokr_w2.copy_(w2)
    ref_net.apply_to()
    net.apply_to()
    net.build_arena("cpu", groups=nat.lora_groups())
    net.refresh_shadows(ref_ops)
    nat.attach_network(net)
    nat.prepare()
    lat, txt, t = inputs()
    with ref_net:
        pred_ref = ref(lat, t, txt)
        w5 = torch.randn(pred_ref.shape, generator=torch.Generator().manual_seed(11))
        (pred_ref * w5).sum().backward()
    with net:
        pred = nat.forward_native(nat.pack_tokens(lat), t, txt, (3, 4, 2))
        assert torch.allclose(pred, nat.pack_tokens(pred_ref.detach()), rtol=2e-4, atol=2e-5)
        net.zero_grad_arena()
        nat.backward_native(nat.pack_tokens(w5))
    for a, b in zip(net.unet_loras, ref_net.unet_loras):
        for x, y, nm in ((a.lokr_w1.grad, b.lokr_w1.grad, "w1"), (a.lokr_w2.grad, b.lokr_w2.grad, "w2")):
            err = ((x - y).norm() / (y.norm() + 1e-12)).item()
            assert err < 5e-4, (a.lora_name, nm, err)


def test_oracle_attention_equals_reference_wan_attention_processor():
    """oracle/wan_ref.Attention against outputs of the reference's WanAttnProcessor2_0 (toolkit/models/wan21/wan_attn.py) executed by
    tests/golden/make_golden.py on the same (seeded) module: self-attention with the float64 complex RoPE and text cross-attention."""
    import json
    import os

    from safetensors import safe_open
    from safetensors.torch import load_file

    path = os.path.join(os.path.dirname(__file__), "golden", "wan_attn.safetensors")
    t = load_file(path)
    with safe_open(path, "pt") as f:
        Fr, Hh, W = json.loads(f.metadata()["grid"])
    torch.manual_seed(31)
    attn = wan_ref.Attention(256, 2, 128)
    with torch.no_grad():
        for p_ in attn.parameters():
            p_.copy_(torch.randn(p_.shape) * 0.05)
        attn.norm_q.weight.copy_(1 + 0.2 * torch.randn(256))
        attn.norm_k.weight.copy_(1 + 0.2 * torch.randn(256))
        chk = torch.stack([v.double().abs().sum() for v in attn.state_dict().values()]).float()
        assert torch.allclose(chk, t["w_checksum"], rtol=1e-6)
        got_self = attn(t["x"], None, wan_ref.wan_rope_freqs(Fr, Hh, W))
        got_cross = attn(t["x"], t["enc"], None)
    assert torch.allclose(got_self, t["self"], rtol=1e-5, atol=1e-6), (got_self - t["self"]).abs().max()
    assert torch.allclose(got_cross, t["cross"], rtol=1e-5, atol=1e-6)


def _wan_dp_worker(rank, world, port, out):
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    torch.set_num_threads(2)
    ref, ref_net, nat, net = build_pair(rank=4)
    step = WanLoRATrainStep(nat, net, ref_ops, lr=1e-3, max_grad_norm=0.5, process_group=dist.group.WORLD)
    for k in range(2):  # every step = a batch list of two micro-batches (gradient accumulation): the all-reduce is issued once
        micro = []
        for j in range(2):
            lat, txt, t = inputs(B=2, seed=60 + 2 * k + j)
            noise = torch.randn(lat.shape, generator=torch.Generator().manual_seed(80 + 2 * k + j))
            sl = slice(rank, rank + 1)  # disjoint shard of each micro-batch
            micro.append(dict(latents=lat[sl], prompt_embeds=txt[sl], noise=noise[sl], timesteps=t[sl]))
        step.step_list(micro)
    torch.save(net.arena_p.clone(), os.path.join(out, f"p{rank}.pt"))
    dist.destroy_process_group()


def test_wan_dp2_gloo_with_accumulation_equals_single_rank(tmp_path):
    """Wan step, 2 ranks x 2 accumulated micro-batches (all-reduce pieces 'late' / 'early' issued by the last backward only)
    == one rank on the concatenated micro-batches; ranks end bit-identical."""
    import os

    import torch.multiprocessing as mp

    from tests.conftest import free_port

    port = free_port()
    mp.spawn(_wan_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = torch.load(tmp_path / "p0.pt"), torch.load(tmp_path / "p1.pt")
    assert torch.equal(p0, p1)
    ref, ref_net, nat, net = build_pair(rank=4)
    step = WanLoRATrainStep(nat, net, ref_ops, lr=1e-3, max_grad_norm=0.5)
    for k in range(2):
        micro = []
        for j in range(2):
            lat, txt, t = inputs(B=2, seed=60 + 2 * k + j)
            noise = torch.randn(lat.shape, generator=torch.Generator().manual_seed(80 + 2 * k + j))
            micro.append(dict(latents=lat, prompt_embeds=txt, noise=noise, timesteps=t))
        step.step_list(micro)
    # DP(2) averages rank gradients of per-rank means over 1 sample = the mean over the 2-sample micro-batch
    assert torch.allclose(net.arena_p, p0, rtol=1e-3, atol=1e-6), (net.arena_p - p0).abs().max()
