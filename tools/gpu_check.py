"""GPU bring-up battery (run through gpurun): hardware-layout probes + per-kernel parity vs fp32 torch math,
each check isolated so one failure does not hide the rest.  Writes gpurun_out/gpu_check.json."""
import json
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402

import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import _capi, ops  # noqa: E402

OUT = {}
dev = "cuda"


def rec(name, fn):
    t0 = time.time()
    try:
        OUT[name] = fn()
        OUT[name]["ok"] = bool(OUT[name].get("ok", True))
    except Exception as e:  # noqa: BLE001
        OUT[name] = {"ok": False, "error": repr(e), "tb": traceback.format_exc()[-1500:]}
    OUT[name]["secs"] = round(time.time() - t0, 3)
    print(name, json.dumps(OUT[name])[:600], flush=True)


def probe_tr16():
    L = _capi.lib()
    res = {}
    for pitch in (64, 72):
        out = torch.zeros(256, dtype=torch.int16, device=dev)
        _capi.check(L.aitk_probe_tr16(C.c_void_p(out.data_ptr()), pitch, _capi.stream_ptr()), "probe_tr16")
        torch.cuda.synchronize()
        got = out.cpu().view(64, 4).tolist()
        # assumption: lane (g = l>>4, n = l&15) receives rows 0..3 of column 16 g + n
        exp = [[r * pitch + (l >> 4) * 16 + (l & 15) for r in range(4)] for l in range(64)]
        res[f"pitch{pitch}"] = {"match": got == exp, "got_first8": got[:8], "got_16_20": got[16:20], "got_32_34": got[32:34]}
    res["ok"] = all(v["match"] for v in res.values())
    return res


def probe_glds():
    L = _capi.lib()
    src = torch.arange(1024, dtype=torch.int32, device=dev)
    out = torch.zeros(1024, dtype=torch.int32, device=dev)
    _capi.check(L.aitk_probe_glds(C.c_void_p(src.data_ptr()), C.c_void_p(out.data_ptr()), _capi.stream_ptr()), "probe_glds")
    torch.cuda.synchronize()
    got = out.cpu().view(4, 64, 4)
    exp = torch.stack([torch.stack([src.cpu()[(w * 64 + (63 - l)) * 4:(w * 64 + (63 - l)) * 4 + 4] for l in range(64)]) for w in range(4)])
    return {"ok": bool((got == exp).all()), "got_w0_l0_3": got[0, :4].tolist(), "got_w1_l0_1": got[1, :2].tolist()}


def probe_mfma():
    L = _capi.lib()
    g = torch.Generator().manual_seed(1)
    a = torch.randint(-4, 5, (32, 16), generator=g).float()
    b = torch.randint(-4, 5, (16, 32), generator=g).float()
    ad, bd = a.to(dev, torch.bfloat16).contiguous(), b.to(dev, torch.bfloat16).contiguous()
    d = torch.zeros(32, 32, device=dev)
    _capi.check(L.aitk_probe_mfma32(C.c_void_p(ad.data_ptr()), C.c_void_p(bd.data_ptr()), C.c_void_p(d.data_ptr()), _capi.stream_ptr()), "probe_mfma32")
    torch.cuda.synchronize()
    err = (d.cpu() - a @ b).abs().max().item()
    return {"ok": err == 0.0, "max_err": err}


def relerr(x, ref):
    return ((x.float() - ref).norm() / (ref.norm() + 1e-30)).item()


def gemm_case(M, N, K, r=0, flags=0, stage=0, seg=False, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = (torch.randn(M, K, generator=g)).to(torch.bfloat16)
    b = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(torch.bfloat16)
    ref = a.float() @ b.float().t() + bias.float()
    kw = {"bias": bias.to(dev)}
    if r:
        a2 = torch.randn(M, r, generator=g).to(torch.bfloat16)
        b2 = (torch.randn(N, r, generator=g) * 0.1).to(torch.bfloat16)
        ref = ref + a2.float() @ b2.float().t()
        kw.update(a2=a2.to(dev), b2=b2.to(dev))
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    aux_ref = None
    if flags & ops.EPI_ACCUM:
        c0 = torch.randn(M, N, generator=g).to(torch.bfloat16)
        out = c0.to(dev).clone()
        ref = ref + c0.float()
    if flags & ops.EPI_GELU:
        u = ref.to(torch.bfloat16)
        aux_ref = u.float()
        ref = torch.nn.functional.gelu(u.float(), approximate="tanh")
        kw["aux_out"] = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    if flags & ops.EPI_DGELU:
        u = torch.randn(M, N, generator=g).to(torch.bfloat16)
        uu = u.float().requires_grad_(True)
        torch.nn.functional.gelu(uu, approximate="tanh").sum().backward()
        ref = ref * uu.grad
        kw["aux_in"] = u.to(dev)
    if flags & ops.EPI_GATE_RES:
        nb = 2 if M % 2 == 0 else 1
        gate = torch.randn(nb, N, generator=g).to(torch.bfloat16)
        res = torch.randn(M, N, generator=g).to(torch.bfloat16)
        y = ref.to(torch.bfloat16)
        aux_ref = y.float()
        ref = res.float() + gate.float().repeat_interleave(M // nb, 0) * y.float()
        kw.update(gate=gate.to(dev), gate_rows=M // nb, aux_in=res.to(dev), aux_out=torch.zeros(M, N, dtype=torch.bfloat16, device=dev))
    ad = a.to(dev)
    if seg:
        # A rows split in 2 segments inside a bigger buffer; C too
        half = M // 2
        abuf = torch.zeros(2, half + 5, K, dtype=torch.bfloat16, device=dev)
        abuf[:, 3:3 + half] = ad.view(2, half, K)
        cbuf = torch.full((2, half + 7, N), 7.0, dtype=torch.bfloat16, device=dev)
        ops.gemm_nt(abuf[0, 3:], b.to(dev), cbuf[0, 2:], flags=flags, M=M, a_seg=(half, (half + 5) * K), c_seg=(half, (half + 7) * N), stage_mode=stage, **kw)
        torch.cuda.synchronize()
        out = cbuf[:, 2:2 + half].reshape(M, N)
        untouched = bool((cbuf[:, :2] == 7).all() and (cbuf[:, 2 + half:] == 7).all())
    else:
        ops.gemm_nt(ad, b.to(dev), out, flags=flags, stage_mode=stage, **kw)
        torch.cuda.synchronize()
        untouched = True
    e = relerr(out.cpu(), ref)
    res = {"rel_err": e, "ok": e < 6e-3 and untouched, "untouched": untouched}
    if aux_ref is not None:
        res["aux_rel_err"] = relerr(kw["aux_out"].cpu(), aux_ref)
        res["ok"] = res["ok"] and res["aux_rel_err"] < 6e-3
    return res


def gemm_bench(M, N, K, stage, r=16, iters=20):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    a2 = torch.randn(M, r, device=dev).to(torch.bfloat16)
    b2 = torch.randn(N, r, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ops.gemm_nt(a, b, out, bias=bias, a2=a2, b2=b2, stage_mode=stage)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm_nt(a, b, out, bias=bias, a2=a2, b2=b2, stage_mode=stage)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / ms / 1e9
    # torch (hipBLASLt) comparator on the same data
    for _ in range(3):
        torch.nn.functional.linear(a, b, bias)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        torch.nn.functional.linear(a, b, bias)
    e1.record()
    torch.cuda.synchronize()
    ms_t = e0.elapsed_time(e1) / iters
    return {"ms": ms, "tflops": tf, "torch_ms": ms_t, "torch_tflops": 2.0 * M * N * K / ms_t / 1e9}


def main():
    print(torch.__version__, torch.cuda.get_device_name(0), flush=True)
    rec("probe_mfma32", probe_mfma)
    rec("probe_tr16", probe_tr16)
    rec("probe_glds", probe_glds)
    for stage in (0, 1):
        rec(f"gemm_256x256x128_s{stage}", lambda: gemm_case(256, 256, 128, stage=stage))
        rec(f"gemm_ragged_s{stage}", lambda: gemm_case(200, 328, 192, stage=stage))
        rec(f"gemm_lora16_s{stage}", lambda: gemm_case(384, 512, 256, r=16, stage=stage))
        rec(f"gemm_lora48_s{stage}", lambda: gemm_case(384, 256, 64, r=48, stage=stage))
        rec(f"gemm_accum_s{stage}", lambda: gemm_case(256, 384, 128, r=16, flags=ops.EPI_ACCUM, stage=stage))
        rec(f"gemm_gelu_s{stage}", lambda: gemm_case(256, 384, 128, r=16, flags=ops.EPI_GELU, stage=stage))
        rec(f"gemm_dgelu_s{stage}", lambda: gemm_case(256, 384, 128, flags=ops.EPI_DGELU, stage=stage))
        rec(f"gemm_gate_s{stage}", lambda: gemm_case(256, 384, 128, r=16, flags=ops.EPI_GATE_RES, stage=stage))
        rec(f"gemm_seg_s{stage}", lambda: gemm_case(300, 256, 128, r=16, seg=True, stage=stage))
        rec(f"gemm_big_s{stage}", lambda: gemm_case(1024, 3072, 3072, r=16, stage=stage))
    for stage in (0, 1):
        rec(f"bench_4608x3072x3072_s{stage}", lambda: gemm_bench(4608, 3072, 3072, stage))
        rec(f"bench_4608x12288x3072_s{stage}", lambda: gemm_bench(4608, 12288, 3072, stage))
        rec(f"bench_18432x3072x12288_s{stage}", lambda: gemm_bench(18432, 3072, 12288, stage))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/gpu_check.json", "w") as fh:
        json.dump(OUT, fh, indent=1)
    bad = [k for k, v in OUT.items() if not v["ok"]]
    print("FAILED:", bad)


if __name__ == "__main__":
    main()
