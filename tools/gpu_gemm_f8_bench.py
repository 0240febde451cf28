"""W8A8 (MX-scaled fp8 MFMA, b_scale_mode 3) vs bf16 persistent 8-phase GEMM on the FLUX step's shapes, one MI355X: TFLOP/s per shape,
random data, LoRA slab K2 = 48 attached, plus the per-token quantisation kernel's bandwidth.  Prints one JSON object."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ai_toolkit_amd  # noqa: E402,F401
from ai_toolkit_amd import _capi, ops  # noqa: E402

bf = torch.bfloat16
dev = "cuda"


def timeit(fn, n=10):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    rows = B * 4608
    shapes = [(rows, 3072, 3072, 0), (rows, 12288, 3072, _capi.EPI_GELU), (rows, 3072, 12288, 0), (rows, 3072, 15360, _capi.EPI_GATE_RES),
              (rows, 3072, 3072, _capi.EPI_GATE_RES), (B * 512, 3072, 3072, 0)]
    out = {"B": B, "shapes": []}
    g = torch.Generator(device=dev).manual_seed(1)
    for M, N, K, flags in shapes:
        x = torch.randn(M, K, device=dev, generator=g).to(bf)
        w = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(bf)
        ws = (w.float().abs().amax(1) / 448.0).contiguous()
        wq = (w.float() / ws[:, None]).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
        a2 = (torch.randn(M, 48, device=dev, generator=g) * 0.1).to(bf)
        b2 = (torch.randn(N, 48, device=dev, generator=g) * 0.1).to(bf)
        bias = torch.zeros(N, dtype=bf, device=dev)
        res = torch.randn(M, N, device=dev, generator=g).to(bf) if flags & _capi.EPI_GATE_RES else None
        gate = torch.randn(B, N, device=dev, generator=g).to(bf) if flags & _capi.EPI_GATE_RES else None
        aux = torch.empty(M, N, dtype=bf, device=dev) if flags else None
        o = torch.empty(M, N, dtype=bf, device=dev)
        xq, xs = torch.empty(M, K, dtype=torch.uint8, device=dev), torch.empty(M, device=dev)
        kw = dict(bias=bias, a2=a2, b2=b2, flags=flags, aux_out=aux, aux_in=res, gate=gate, gate_rows=M // B if gate is not None else 0)
        t_q = timeit(lambda: ops.quant_rows_fp8(x, xq, xs))
        t_bf = timeit(lambda: ops.gemm_nt(x, w, o, **kw))
        t_f8 = timeit(lambda: ops.gemm_nt(xq, wq, o, a_scale=xs, b_scale=ws, b_scale_mode=3, **kw))
        fl = 2.0 * M * N * (K + 48)
        out["shapes"].append({"M": M, "N": N, "K": K, "flags": flags, "bf16_ms": round(t_bf, 4), "bf16_tflops": round(fl / t_bf / 1e9, 1),
                              "f8_ms": round(t_f8, 4), "f8_tflops": round(fl / t_f8 / 1e9, 1), "quant_ms": round(t_q, 4),
                              "quant_GBps": round(3.0 * M * K / t_q / 1e6, 1), "speedup_incl_quant": round(t_bf / (t_f8 + t_q), 3)})
        print(out["shapes"][-1], flush=True)
        del x, w, wq, a2, b2, o, xq
    print(json.dumps(out))


if __name__ == "__main__":
    main()
