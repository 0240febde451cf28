"""Timing-only ablation library for tools/gpu_attn_ab.py: the product objects + attention.hip compiled with -DAITK_ABL_ATTN_B128 ->
ai-toolkit_amd/libaitk_abl_attn_b128.so (git-ignored; loaded through AITK_LIB_PATH by the A/B tool only)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util

spec = importlib.util.spec_from_file_location("aitk_build", os.path.join(os.path.dirname(__file__), "..", "ai-toolkit_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
b.build(verbose=False)
others = [os.path.join(b.OBJDIR, os.path.splitext(os.path.basename(s))[0] + ".o") for s in b.sources() if not s.endswith("attention.hip")]
VARIANTS = {"r1": (("b128", "AITK_ABL_ATTN_B128"), ("noexp", "AITK_ABL_NOEXP"), ("nomfma", "AITK_ABL_NOMFMA"), ("nolds", "AITK_ABL_NOLDS")),
            # experiments on the pipelined dK/dV kernel (numerically valid except `nodma`)
            "pipe": (("p_nodma", "AITK_PIPE_NODMA"),)}
for tag, macro in VARIANTS[sys.argv[1] if len(sys.argv) > 1 else "r1"]:
    obj = os.path.join(b.OBJDIR, f"attention_abl_{tag}.o")
    subprocess.check_call([b._hipcc()] + b.FLAGS + [f"-D{m}" if i == 0 else m for i, m in enumerate(macro.split())] + ["-c", os.path.join(b.CSRC, "attention.hip"), "-o", obj])
    out = os.path.join(b.HERE, f"libaitk_abl_attn_{tag}.so")
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + others + [obj])
    print(out)
