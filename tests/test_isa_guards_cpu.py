"""Guards on the generated gfx950 code of the persistent GEMM (no GPU: the in-tree object is disassembled).

The K loops of gemm8.hip retire their LDS-DMA pieces with hand-counted `s_waitcnt vmcnt(8)`; the DMAs are inline asm the compiler's waitcnt pass does not
see.  What it DOES see are the epilogue's global loads, and if one of them is still on its scoreboard at the tile loop's back edge it protects the reuse of
that load's destination registers by the fragment reads with `s_waitcnt vmcnt(0)` INSIDE the steady loop — correct, silent, and 20 % slower per K-tile
(round 4, profiles/r04_gemm8_tile_switch.md).  The cure is the builtin vmcnt(0) in front of the loop's exit test; this test keeps it cured: inside any
branch-free stretch that holds eight or more MFMAs the only vmcnt waits are the counted ones."""
import os
import re
import shutil
import subprocess

import pytest

import ai_toolkit_amd  # noqa: F401
from ai_toolkit_amd import _capi

LLVM = "/opt/rocm/lib/llvm/bin"
OBJ = os.path.join(os.path.dirname(_capi.LIB_PATH), "build", "gemm8.o")


def _disassemble(tmp_path):
    if not os.path.exists(OBJ):
        import __graft_entry__

        __graft_entry__.build()
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "gemm8.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", OBJ, str(tmp_path / "host.o")])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--output={co}"])
    return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co], text=True)


@pytest.mark.skipif(not (os.path.exists(os.path.join(LLVM, "llvm-objdump")) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"))),
                    reason="needs the ROCm LLVM tools")
def test_gemm8_k_loops_carry_only_the_counted_vmcnt_waits(tmp_path):
    text = _disassemble(tmp_path)
    kernels = re.split(r"\n(?=[0-9a-f]+ <)", text)
    seen = 0
    for body in kernels:
        m = re.match(r"[0-9a-f]+ <(\w+)>:", body)
        if not m or "gemm_nt_8phase" not in m.group(1):
            continue
        seen += 1
        stretch_mfma, stretch_waits = 0, []
        dense = 0
        for line in body.splitlines()[1:]:
            ins = line.strip().split("//")[0].strip()
            if ins.startswith(("s_cbranch", "s_branch", "s_endpgm")):
                if stretch_mfma >= 8:
                    dense += 1
                    bad = [w for w in stretch_waits if w != 8]
                    assert not bad, f"{m.group(1)}: compiler-inserted vmcnt{bad} inside a {stretch_mfma}-MFMA stretch"
                stretch_mfma, stretch_waits = 0, []
                continue
            if ins.startswith("v_mfma"):
                stretch_mfma += 1
            w = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", ins)
            if w:
                stretch_waits.append(int(w.group(1)))
        assert dense >= 4, (m.group(1), dense)  # the steady loop and the tail phases were found
    assert seen >= 5, seen
