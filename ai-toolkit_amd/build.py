"""Builds the in-tree HIP library: hipcc --offload-arch=gfx950 -> ai-toolkit_amd/libaitk_mi355.so.

Cross-compiles without a GPU.  Objects are cached per source mtime so rebuilds only touch edited files.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libaitk_mi355.so")
OBJDIR = os.path.join(HERE, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stamp(path):
    h = hashlib.sha1()
    for f in [path] + [os.path.join(CSRC, x) for x in sorted(os.listdir(CSRC)) if x.endswith((".h", ".inc"))] + [
        os.path.join(HERE, "..", "include", "aitk_mi355.h")
    ]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(verbose=True, force=False):
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in sources():
        base = os.path.splitext(os.path.basename(src))[0]
        obj = os.path.join(OBJDIR, base + ".o")
        stampf = obj + ".stamp"
        st = _stamp(src)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stampf) and open(stampf).read() == st:
            continue
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print("[aitk build]", " ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT), stampf, st, src))
    failed = False
    for p, stampf, st, src in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(out.decode(errors="replace"))
            sys.stderr.write(f"[aitk build] FAILED: {src}\n")
        else:
            with open(stampf, "w") as fh:
                fh.write(st)
    if failed:
        raise RuntimeError("hipcc failed")
    if procs or not os.path.exists(OUT):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print("[aitk build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
