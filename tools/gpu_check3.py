"""GPU battery #3: step kernels vs oracle/ref_ops + full-size FLUX.1-dev step timing."""
import json
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import ops  # noqa: E402
from oracle import ref_ops  # noqa: E402

OUT = {}
dev = "cuda"
bf = torch.bfloat16


def rec(name, fn):
    t0 = time.time()
    try:
        OUT[name] = fn()
    except Exception as e:  # noqa: BLE001
        OUT[name] = {"ok": False, "error": repr(e), "tb": traceback.format_exc()[-1500:]}
    OUT[name]["secs"] = round(time.time() - t0, 3)
    print(name, json.dumps(OUT[name])[:700], flush=True)


def rel(x, ref):
    return ((x.float() - ref.float()).norm() / (ref.float().norm() + 1e-30)).item()


def R(*shape, s=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g) * s


def t_gemv():
    res = {}
    for (Bm, N, K, r) in ((1, 18432, 3072, 16), (4, 9216, 3072, 16), (3, 3072, 256, 0), (2, 3072, 768, 0)):
        x = R(Bm, K, seed=1).to(bf).to(dev)
        w = R(N, K, s=0.03, seed=2).to(bf).to(dev)
        bias = R(N, seed=3).to(bf).to(dev)
        kw = {}
        if r:
            kw = dict(t=R(Bm, r, seed=4).to(bf).to(dev), bl=R(N, r, s=0.1, seed=5).to(bf).to(dev))
        o1 = R(Bm, N, seed=6).to(bf).to(dev)
        o2 = o1.clone()
        ops.gemv_nt(x, w, o1, bias=bias, accumulate=True, **kw)
        ref_ops.gemv_nt(x, w, o2, bias=bias, accumulate=True, **kw)
        res[f"{Bm}x{N}x{K}r{r}"] = rel(o1, o2)
    res["ok"] = max(v for v in res.values()) < 6e-3
    return res


def t_noise_mse():
    B, Cc, Hh, W = 2, 16, 32, 24
    lat, noi = R(B, Cc, Hh, W, seed=7).to(bf).to(dev), R(B, Cc, Hh, W, seed=8).to(bf).to(dev)
    t = torch.tensor([123.0, 900.5], device=dev)
    n1, t1 = torch.empty(B, Hh * W // 4, 64, dtype=bf, device=dev), torch.empty(B, Hh * W // 4, 64, dtype=bf, device=dev)
    n2, t2 = torch.empty_like(n1), torch.empty_like(t1)
    ops.flow_noise_pack(lat, noi, t, n1, t1)
    ref_ops.flow_noise_pack(lat, noi, t, n2, t2)
    pred = R(B, Hh * W // 4, 64, seed=9).to(bf).to(dev)
    w = torch.tensor([1.0, 0.5], device=dev)
    d1, d2 = torch.empty_like(pred), torch.empty_like(pred)
    l1, l2 = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    lp1, lp2 = torch.zeros(B, device=dev), torch.zeros(B, device=dev)
    ops.mse_loss_grad(pred, t1, d1, lp1, l1, weight=w)
    ref_ops.mse_loss_grad(pred, t2, d2, lp2, l2, weight=w)
    torch.cuda.synchronize()
    res = {"noisy": rel(n1, n2), "target": rel(t1, t2), "dpred": rel(d1, d2), "loss": abs(l1.item() - l2.item()) / l2.item(),
           "lps": rel(lp1, lp2)}
    res["ok"] = res["noisy"] < 3e-3 and res["target"] < 3e-3 and res["dpred"] < 5e-3 and res["loss"] < 1e-5 and res["lps"] < 1e-5
    return res


def t_adamw():
    n = 1_000_003
    p = R(n, seed=10).to(dev)
    g = (R(n, seed=11) * 0.01).to(dev)
    m, v = (R(n, seed=12) * 0.01).to(dev), (R(n, seed=13).abs() * 1e-4).to(dev)
    ema = R(n, seed=14).to(dev)
    a = [t.clone() for t in (p, g, m, v, ema)]
    b = [t.clone() for t in (p, g, m, v, ema)]
    no1, no2 = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    kw = dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.01, step=3, max_norm=1.0, ema_decay=0.99, grad_scale=0.5)
    ops.adamw_ema_step(a[0], a[1], a[2], a[3], ema=a[4], norm_out=no1, **kw)
    ref_ops.adamw_ema_step(b[0], b[1], b[2], b[3], ema=b[4], norm_out=no2, **kw)
    torch.cuda.synchronize()
    res = {"p": rel(a[0] - p, b[0] - p), "m": rel(a[2], b[2]), "v": rel(a[3], b[3]), "ema": rel(a[4], b[4]),
           "norm": abs(no1.item() - no2.item()) / no2.item()}
    res["ok"] = max(res.values()) < 1e-4
    return res


def t_shadows():
    entries = []
    off = soff = 0
    shapes = [(16, 3072, 1), (3072, 16, 2), (16, 96, 1), (40, 16, 2), (48, 64, 0), (24, 8, 0)]  # (rows, cols, AitkShadowDesc kind)
    for r, c, kind in shapes:
        n = r * c
        if kind == 0:
            entries.append((off, r, c, 0, soff, soff + n, 0))
            soff += 2 * n
        elif kind == 1:
            entries.append((off, r, c, 1, soff, soff + n, soff + 2 * n))
            soff += 5 * n
        else:
            entries.append((off, r, c, 2, soff, soff + 3 * n, soff + 4 * n))
            soff += 5 * n
        off += n
    arena = R(off, seed=15).to(dev)
    s1 = torch.zeros(soff, dtype=bf, device=dev)
    s2 = torch.zeros(soff, dtype=bf, device=dev)
    ops.refresh_shadows(arena, s1, ops.make_shadow_table(entries, dev))
    ref_ops.refresh_shadows(arena, s2, ref_ops.make_shadow_table(entries, dev))
    torch.cuda.synchronize()
    return {"ok": bool(torch.equal(s1, s2))}


def full_flux_step(B=1, steps=3):
    from ai_toolkit_amd.flux import FluxTransformer2DModel
    from ai_toolkit_amd.lora import FusedLoRANetwork
    from ai_toolkit_amd.trainer import FluxLoRATrainStep

    t0 = time.time()
    model = FluxTransformer2DModel(dtype=bf, device=dev, ops=ops)
    g = torch.Generator(device=dev).manual_seed(1234)
    with torch.no_grad():
        for mod in model.modules():
            if mod.__class__.__name__ == "Linear":
                mod.weight.copy_((torch.randn(mod.weight.shape, device=dev, generator=g) * 0.02).to(bf))
    torch.manual_seed(1234)
    net = FusedLoRANetwork(model, lora_dim=16)
    with torch.no_grad():
        for m in net.unet_loras:
            m.lora_up.weight.normal_(0, 1e-3)
    net.apply_to()
    net.build_arena(dev, ema=True, groups=model.lora_groups())
    net.refresh_shadows(ops)
    model.attach_network(net)
    model.prepare()
    torch.cuda.synchronize()
    build_s = time.time() - t0
    step = FluxLoRATrainStep(model, net, ops, lr=1e-4, ema_decay=0.99, seed=42)
    gen = torch.Generator(device=dev).manual_seed(42)
    lat = torch.randn(B, 16, 128, 128, device=dev, generator=gen).to(bf)
    emb = (torch.randn(B, 512, 4096, device=dev, generator=gen) * 0.1).to(bf)
    pooled = (torch.randn(B, 768, device=dev, generator=gen) * 0.1).to(bf)
    losses = []
    times = []
    for k in range(steps):
        torch.cuda.synchronize()
        t1 = time.time()
        loss = step.step(lat, emb, pooled)
        torch.cuda.synchronize()
        times.append(time.time() - t1)
        losses.append(loss.item())
    mem = torch.cuda.max_memory_allocated() / 2 ** 30
    return {"ok": all(l == l and abs(l) < 1e4 for l in losses), "losses": losses, "step_s": times, "build_s": build_s,
            "peak_mem_GiB": mem, "n_lora": len(net.unet_loras), "n_params": net.arena_p.numel(),
            "grad_norm": step.grad_norm.item()}


def main():
    rec("gemv", t_gemv)
    rec("noise_mse", t_noise_mse)
    rec("adamw", t_adamw)
    rec("shadows", t_shadows)
    if "--no-full" not in sys.argv:
        rec("full_flux_B1", lambda: full_flux_step(1, 3))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/gpu_check3.json", "w") as fh:
        json.dump(OUT, fh, indent=1)
    print("FAILED:", [k for k, v in OUT.items() if not v.get("ok")])


if __name__ == "__main__":
    main()
