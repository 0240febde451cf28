"""To run FIRST when a GPU is available again (round 3 ended with the GPU budget spent): validates and times `recompute_gelu`.
  1. aitk_lora_wgrad2 against the oracle (gelu(u) alone; [o | gelu(u)]; ranks 16 / 32; ragged M) and against aitk_lora_wgrad on the
     materialised operand (must be bit-identical: same kernel body, same values);
  2. the FLUX step (small config) with the flag on == the flag off, bit for bit (loss and the gradient arena);
  3. peak memory and step time of the full-size step at B = 7 with and without the flag, then B = 8, 9, 10 with it.
On success: make `recompute_gelu` the default in bench.py's build_flux, raise the auto batch, move the checks of 1-2
into tests/test_gpu_*.py."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import ops  # noqa: E402
from oracle import ref_ops  # noqa: E402

bf = torch.bfloat16
out = {}


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


# ---- 1. kernel
g = torch.Generator().manual_seed(1)
for (M, R, d_o, d_u) in ((1000, 16, 0, 512), (2048, 16, 384, 1024), (777, 32, 128, 256)):
    S = (torch.randn(M, 3 * R, generator=g) * 0.3).to(bf).cuda()
    S[:, 2 * R:] = S[:, :R]
    u = torch.randn(M, d_u, generator=g).to(bf).cuda()
    o = torch.randn(M, d_o, generator=g).to(bf).cuda() if d_o else None
    L = d_o + d_u
    a, b, c = (torch.zeros(R, L, device="cuda") for _ in range(3))
    ops.lora_wgrad(S, o, a, M=M, split=R, g2=u, g2_act="gelu")
    ref_ops.lora_wgrad(S, o, b, M=M, split=R, g2=u, g2_act="gelu")
    h = torch.empty(M, d_u, dtype=bf, device="cuda")
    # the materialised operand through the GEMM's own GELU epilogue is what the forward pass feeds the layer; here: the same formula on the host
    h.copy_(torch.nn.functional.gelu(u.float(), approximate="tanh").to(bf))
    full = h if o is None else torch.cat((o, h), 1).contiguous()
    ops.lora_wgrad(S, full, c, M=M, split=R)
    out[f"kernel_M{M}_R{R}_{d_o}+{d_u}"] = {"vs_oracle": rel(a, b), "vs_materialised": rel(a, c)}
    print(f"wgrad2 M={M} R={R} [{d_o} | gelu {d_u}]: vs oracle {rel(a, b):.2e}, vs aitk_lora_wgrad on the materialised operand {rel(a, c):.2e}", flush=True)
    assert rel(a, b) < 2e-4 and rel(a, c) < 2e-3

# ---- 2. small FLUX step, flag on vs off
from ai_toolkit_amd.trainer import FluxLoRATrainStep  # noqa: E402
from tests.test_gpu_e2e import _batch, _build  # noqa: E402

res = {}
for flag in (False, True):
    ref, ref_net, nat, net = _build()
    nat.recompute_gelu = flag
    st = FluxLoRATrainStep(nat, net, ops, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
    lat, emb, pooled, noise, ts = _batch(2)
    loss = st.step(lat, emb, pooled, noise=noise, timesteps=ts).item()
    res[flag] = (loss, net.arena_g.clone())
same = res[False][0] == res[True][0] and torch.equal(res[False][1], res[True][1])
out["small_step_bit_identical"] = bool(same)
print("small FLUX step: flag on == flag off bit for bit:", same, " grad rel diff", rel(res[True][1], res[False][1]), flush=True)

# ---- 3. full size
import bench  # noqa: E402

dev = torch.device("cuda", 0)
model, net, _ = bench.build_flux(dev, rank=16)
step = FluxLoRATrainStep(model, net, ops, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99, timestep_type="linear", seed=1000)
for flag, B in ((False, 7), (True, 7), (True, 8), (True, 9), (True, 10)):
    model.recompute_gelu = flag
    torch.cuda.reset_peak_memory_stats()
    try:
        lat, emb, pooled = bench.make_batch(dev, B, seed=42)
        for _ in range(2):
            step.step(lat, emb, pooled)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            step.step(lat, emb, pooled)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 4 * 1e3
        out[f"full_B{B}_recompute{int(flag)}"] = {"ms_per_step": ms, "images_per_s": B / ms * 1e3, "peak_GiB": torch.cuda.max_memory_allocated() / 2 ** 30}
    except torch.OutOfMemoryError as e:
        out[f"full_B{B}_recompute{int(flag)}"] = {"error": str(e)[:120]}
    print(f"B={B} recompute={flag}:", out[f"full_B{B}_recompute{int(flag)}"], flush=True)
    del lat, emb, pooled
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r04_recompute_gelu.json", "w"), indent=1)
