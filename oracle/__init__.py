"""ORACLE — test infrastructure only.

CPU restatements (plain PyTorch, fp32) of the reference algorithm on the LoRA train-step hot path.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and only as the checker.

Pinning status:
  * LoRA layer / network naming / state-dict format (oracle/lora_ref.py) ...... PINNED against the reference's own
    toolkit.lora_special classes executed here under import shims (tests/golden/make_golden.py -> tests/golden/*.safetensors)
  * flow-match scheduler (oracle/flowmatch_ref.py) ............................. PINNED the same way (linear / sigmoid /
    add_noise executed from the reference file with a stub base class)
  * FLUX transformer math (oracle/flux_ref.py) ................................. PARITY UNPINNED: the arithmetic lives in the
    un-vendored `diffusers` dependency, no golden vectors exist in the reference and diffusers is not installed.
"""
