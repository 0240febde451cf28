"""Same-box A/B library for the XCD-aware workgroup order of the attention kernels: compiles a copy of csrc/attention.hip whose attn_wg_coords
walks the (batch, head, tile) list in plain block order, links it with the product's other objects into ai-toolkit_amd/libaitk_abl_attn_noxcd.so
(picked up by tools/gpu_attn_ab.py; bench.py takes it through AITK_LIB_PATH).  Not part of the product build."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-toolkit_amd"))
import build as b  # noqa: E402

b.build(verbose=False)
src = open(os.path.join(b.CSRC, "attention.hip")).read()
old = "  const int l = xcd * per + min(xcd, rem) + slot;"
assert src.count(old) == 1
src = src.replace(old, "  const int l = id + 0 * (xcd + per + rem + slot);")
tmp = os.path.join(b.OBJDIR, "attention_noxcd.hip")
open(tmp, "w").write(src)
obj = os.path.join(b.OBJDIR, "attention_noxcd.o")
subprocess.check_call([b._hipcc()] + b.FLAGS + ["-I", b.CSRC, "-c", tmp, "-o", obj], stderr=subprocess.DEVNULL)
objs = [os.path.join(b.OBJDIR, os.path.splitext(os.path.basename(s))[0] + ".o") for s in b.sources()]
objs = [obj if o.endswith("/attention.o") else o for o in objs]
out = os.path.join(ROOT, "ai-toolkit_amd", "libaitk_abl_attn_noxcd.so")
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
