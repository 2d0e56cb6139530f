#!/usr/bin/env python
"""Compile the oracle's C restatements (test / baseline infrastructure, not product code) into oracle/_build/.

    oracle/csrc/roi_align_cpu.c -> oracle/_build/libroi_align_cpu.so   (gcc -O3 -fopenmp -ffp-contract=off)

Called by ``__graft_entry__.build()``; ``oracle/_build/`` is git-ignored (``*.so``) and travels to the GPU box with the
snapshot like the product's own libraries.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False, quiet=True):
    src = os.path.join(HERE, "csrc", "roi_align_cpu.c")
    out_dir = os.path.join(HERE, "_build")
    out = os.path.join(out_dir, "libroi_align_cpu.so")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    os.makedirs(out_dir, exist_ok=True)
    cmd = ["gcc", "-O3", "-fopenmp", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", out, "-lm"]
    if not quiet:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, quiet=False))
