"""Diagnostic: fused (arena kernels) vs torch optimizer tail of the trainer loop on the small FLUX pair — per-step relative differences of
gradients / parameters / moments, and the same comparison with the model taken out (same gradients fed to both tails)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def rel(a, b):
    return float((a.detach().double() - b.detach().double()).norm() / b.detach().double().norm().clamp_min(1e-30))


def main():
    from tests.test_gpu_trainer_path import _loop
    from oracle.pairs import batch
    from types import SimpleNamespace

    a, nat_a = _loop(True)
    b, nat_b = _loop(False)
    for k in range(4):
        lat, emb, pooled, _, _ = batch(2, seed=60 + k)
        outs = []
        for loop, env in ((a, "1"), (b, "0")):
            os.environ["AITK_FUSE_TRAINER_STEP"] = env
            loop.optimizer.zero_grad()
            noisy, ts, target = loop.process_batch(lat)
            with loop.network:
                pred = loop.sd.get_noise_prediction(noisy, ts, SimpleNamespace(text_embeds=emb, pooled_embeds=pooled), guidance_embedding_scale=1.0)
                loss = torch.nn.functional.mse_loss(pred.float(), target.float(), reduction="none").mean([1, 2, 3]).mean()
                loss.backward()
            net = loop.sd.get_model_to_train().network
            g_raw = net.arena_g.clone()
            torch.nn.utils.clip_grad_norm_(loop.params, 1.0)
            g_clip = net.arena_g.clone()
            p_before = net.arena_p.clone()
            loop.optimizer.step()
            loop.optimizer.zero_grad(set_to_none=True)
            loop.ema.update()
            outs.append((loss.item(), g_raw, g_clip, p_before, net.arena_p.clone(), pred.detach().float().clone()))
        (la, gra, gca, pba, pa, pra), (lb, grb, gcb, pbb, pb, prb) = outs
        ma = torch.cat([a.optimizer.state[p]["exp_avg"].reshape(-1) for p in a.params])
        mb = torch.cat([b.optimizer.state[p]["exp_avg"].reshape(-1) for p in b.params])
        va = torch.cat([a.optimizer.state[p]["exp_avg_sq"].reshape(-1) for p in a.params])
        vb = torch.cat([b.optimizer.state[p]["exp_avg_sq"].reshape(-1) for p in b.params])
        print(f"step {k}: loss {la:.6f} {lb:.6f} pred {rel(pra, prb):.2e} g_raw {rel(gra, grb):.2e} g_clip {rel(gca, gcb):.2e} p_before {rel(pba, pbb):.2e} "
              f"p_after {rel(pa, pb):.2e} dp {rel(pa - pba, pb - pbb):.2e} m {rel(ma, mb):.2e} v {rel(va, vb):.2e}", flush=True)
    # the model taken out: identical p, g, m, v into both tails
    from ai_toolkit_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    n = 1 << 20
    p0 = torch.randn(n, device="cuda", generator=g) * 0.02
    for scale in (1e-3, 1e-5, 1e-7):
        pk, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        pt = torch.nn.Parameter(p0.clone())
        opt = torch.optim.AdamW([pt], lr=1e-3, eps=1e-6, weight_decay=0.01)
        for k in range(3):
            gr = torch.randn(n, device="cuda", generator=g) * scale
            ops.adamw_ema_step(pk, gr, m, v, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.01, step=k + 1, max_norm=0.0, ema=None)
            pt.grad = gr.clone()
            opt.step()
            print(f"isolated |g|~{scale:g} step {k}: p {rel(pk, pt):.2e} dp {rel(pk - p0, pt.detach() - p0):.2e} m {rel(m, opt.state[pt]['exp_avg']):.2e} v {rel(v, opt.state[pt]['exp_avg_sq']):.2e}")


if __name__ == "__main__":
    main()
