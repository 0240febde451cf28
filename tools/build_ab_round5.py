"""Same-box A/B libraries for two round-5 VALU questions (picked up through AITK_LIB_PATH by tools/gpu_valu_ab.py; not part of the product build):

  libaitk_abl_gelu_scalar.so    the GEMM / skinny objects compiled with the GELU(tanh) forms as they were before round 5's packed pairs
                                (common.h: x + k1 x^3 chain, two multiplies in front of v_exp_f32, per-element scalar code)
  libaitk_abl_attn_fwd_pk.so    attention forward with the softmax's scale-subtract (v_pk_fma_f32) and row sum (v_pk_add_f32) on register pairs
  libaitk_abl_attn_fwd_2sum.so  attention forward, scalar ops, row sum in two independent chains

Every variant is a copy of the product source with one text substitution, compiled into ai-toolkit_amd/build/ab5/ and linked with the product's other objects."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-toolkit_amd"))
import build as b  # noqa: E402

GELU_NEW_BEGIN = "// GELU(tanh) and its derivative on PAIRS"
GELU_NEW_END = "__device__ __forceinline__ float silu_f(float x)"
GELU_OLD = '''__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  return x * sigmoid_fast_f(2.0f * k0 * (x + k1 * x * x * x));
}
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float x2 = x * x;
  const float s = sigmoid_fast_f(2.0f * k0 * (x + k1 * x * x2));
  return s + 2.0f * x * s * (1.0f - s) * (k0 * (1.0f + 3.0f * k1 * x2));
}
__device__ __forceinline__ f32x2_t gelu_tanh_2(f32x2_t x) { return f32x2_t{gelu_tanh_f(x.x), gelu_tanh_f(x.y)}; }
__device__ __forceinline__ f32x2_t gelu_tanh_grad_2(f32x2_t x) { return f32x2_t{gelu_tanh_grad_f(x.x), gelu_tanh_grad_f(x.y)}; }
'''

FWD_OLD = '''    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(s[j][r], c2, -m_run));
        s[j][r] = e;
        ps += e;
      }
    l_run += ps;
'''
FWD_PK = '''    f32x2_t ps2 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2_t z = __builtin_elementwise_fma(f32x2_t{s[j][r], s[j][r + 1]}, f32x2_t{c2, c2}, f32x2_t{-m_run, -m_run});
        const f32x2_t e = {__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)};
        s[j][r] = e.x;
        s[j][r + 1] = e.y;
        ps2 += e;
      }
    l_run += ps2.x + ps2.y;
'''
FWD_2SUM = '''    float ps = 0.f, ps_b = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float e0 = __builtin_amdgcn_exp2f(fmaf(s[j][r], c2, -m_run));
        const float e1 = __builtin_amdgcn_exp2f(fmaf(s[j][r + 1], c2, -m_run));
        s[j][r] = e0;
        s[j][r + 1] = e1;
        ps += e0;
        asm volatile("" : "+v"(ps));  // keep the two chains apart (the SLP vectoriser would otherwise pair them into v_pk_add_f32)
        ps_b += e1;
        asm volatile("" : "+v"(ps_b));
      }
    l_run += ps + ps_b;
'''


def compile_variant(name, edits, files):
    """edits: {file name in csrc: (old, new) or callable}; files: the .hip sources to recompile against the edited tree."""
    d = os.path.join(b.OBJDIR, "ab5", name)
    shutil.rmtree(d, ignore_errors=True)
    shutil.copytree(b.CSRC, d)
    ah = os.path.join(d, "aitk_args.h")  # the copy sits at another depth: point the public header at its real place
    txt = open(ah).read().replace('"../../include/aitk_mi355.h"', '"%s"' % os.path.join(ROOT, "include", "aitk_mi355.h"))
    open(ah, "w").write(txt)
    for fn, ed in edits.items():
        p = os.path.join(d, fn)
        src = open(p).read()
        if callable(ed):
            src = ed(src)
        else:
            assert src.count(ed[0]) == 1, (name, fn)
            src = src.replace(ed[0], ed[1])
        open(p, "w").write(src)
    objs = {}
    procs = []
    for f in files:
        obj = os.path.join(d, f.replace(".hip", ".o"))
        objs[f.replace(".hip", ".o")] = obj
        procs.append(subprocess.Popen([b._hipcc()] + b.FLAGS + ["-I", os.path.join(ROOT, "include"), "-c", os.path.join(d, f), "-o", obj], stderr=subprocess.PIPE))
    for p in procs:
        _, err = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(err.decode()[-3000:])
    link = []
    for s in b.sources():
        base = os.path.splitext(os.path.basename(s))[0] + ".o"
        link.append(objs.get(base, os.path.join(b.OBJDIR, base)))
    out = os.path.join(ROOT, "ai-toolkit_amd", f"libaitk_abl_{name}.so")
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + link)
    print(out)
    return out


def gelu_scalar(src):
    i, j = src.index(GELU_NEW_BEGIN), src.index(GELU_NEW_END)
    return src[:i] + GELU_OLD + src[j:]


def main():
    b.build(verbose=False)
    which = set(sys.argv[1:]) or {"gelu", "attn"}
    if "gelu" in which:
        compile_variant("gelu_scalar", {"common.h": gelu_scalar}, ["gemm8.hip", "gemm.hip", "lora_skinny.hip"])
    if "attn" in which:
        compile_variant("attn_fwd_pk", {"attention.hip": (FWD_OLD, FWD_PK)}, ["attention.hip"])
        compile_variant("attn_fwd_2sum", {"attention.hip": (FWD_OLD, FWD_2SUM)}, ["attention.hip"])


if __name__ == "__main__":
    main()
