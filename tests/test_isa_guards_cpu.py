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


def _code_object(tmp_path, obj=OBJ):
    if not os.path.exists(obj):
        import __graft_entry__

        __graft_entry__.build()
    base = os.path.basename(obj)
    fat, co = str(tmp_path / (base + ".fat")), str(tmp_path / (base + ".co"))
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj, str(tmp_path / (base + ".host"))])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--output={co}"])
    return co


def _disassemble(tmp_path):
    return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", _code_object(tmp_path)], text=True)


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
            if ins.startswith("v_mfma_f32_32x32") or ins.startswith("v_mfma_scale_f32_32x32"):  # the K loops' tile products (the EMIT_T epilogue's 16x16x32 products wait for their operand loads: not a K loop)
                stretch_mfma += 1
            w = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", ins)
            if w:
                stretch_waits.append(int(w.group(1)))
        assert dense >= 4, (m.group(1), dense)  # the steady loop and the tail phases were found
    assert seen >= 5, seen


# registers the hot kernels may spill (kernel-name fragment -> bound); the spills that exist sit outside the K / KV loops (epilogues, prologues)
SPILL_BOUNDS = {
    "gemm8.o": {"gemm_nt_8phase_kernel": 8, "gemm_nt_8phase_grouped_kernel": 8, "gemm_nt_8phase_conv_kernel": 16, "gemm_nt_8phase_ge_kernel": 0,
                "gemm_nt_8phase_grouped_ge_kernel": 0},
    "attention.o": {"attn_fwd_kernel": 0, "attn_bwd_dq_kernel": 0, "attn_bwd_dkdv_ws_kernel": 2, "attn_bwd_dkdv_pipe_kernel": 0, "attn_bwd_dkdv_kernel": 0},
    "lora_skinny.o": {"lora_down16_kernel": 0, "lora_wgrad_kernel": 0, "lora_wgrad_fused_kernel": 0, "lora_bwd_fused_ct_kernel": 0},
    "norm_elem.o": {"ln_mod_bwd_row2_kernel": 0, "qkv_post_fwd_kernel": 0, "qkv_post_bwd_kernel": 0},
}


@pytest.mark.skipif(not (os.path.exists(os.path.join(LLVM, "llvm-readelf")) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"))),
                    reason="needs the ROCm LLVM tools")
def test_hot_kernels_stay_within_their_spill_bounds(tmp_path):
    """A register spill inside a hand-scheduled loop is a silent 2x: the code-object metadata of every hot kernel is held to the spill count it has today."""
    build_dir = os.path.dirname(OBJ)
    for obj, bounds in SPILL_BOUNDS.items():
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", _code_object(tmp_path, os.path.join(build_dir, obj))], text=True)
        kernels = re.findall(r"\.name:\s+(\S+).*?\.vgpr_spill_count:\s+(\d+)", notes, re.S)
        assert kernels, obj
        for frag, bound in bounds.items():
            hits = [(n, int(c)) for n, c in kernels if frag in n]
            assert hits, (obj, frag)
            over = [(n, c) for n, c in hits if c > bound]
            assert not over, (obj, over, bound)
