"""Build ``csrc/libsmot_emm.so`` (HIP, gfx950) in-tree with hipcc.

    python siam-mot_amd/build.py            # build if stale
    python siam-mot_amd/build.py --force

Two libraries come out of the same sources:
  csrc/libsmot_emm.so        the product: no environment variable is read, no kernel A/B switch, no ablation;
  csrc/libsmot_emm_debug.so  the measurement build (-DSMOT_DEBUG, + measure/csrc/*.hip and *.inc): older kernel generations,
                             A/B switches and timing ablations for tools/ and the A/B tests (csrc/knobs.h).

The library is plain HIP behind a C ABI (include/smot_emm.h): no torch headers, so it is
compiled with hipcc directly rather than through torch.utils.cpp_extension.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libsmot_emm.so")
LIB_DEBUG = os.path.join(CSRC, "libsmot_emm_debug.so")
MEASURE_CSRC = os.path.join(os.path.dirname(HERE), "measure", "csrc")      # measurement-only sources (not product)
DEBUG_ONLY_SOURCES = ["xcorr_variants.hip", "sr_xcorr_plan.hip"]
SOURCES = ["common.hip", "roi_align.hip", "xcorr.hip", "predictor.hip", "decode.hip", "sr_xcorr.hip", "sr_xcorr_small.hip", "nms.hip", "tower_wino.hip", "tower_conv.hip", "preprocess.hip",
           "emm_fused.hip", "track_solver.hip", "box_refine.hip", "linear_rows.hip", "memory_carry.hip"]
ARCH = "gfx950"
# -fno-slp-vectorize: keeps the xcorr FMA stream as v_fma_f32 with an SGPR tap operand instead of
# v_pk_fma_f32 + register shuffles (measured: 450 pk_fma + 204 movs vs 900 fma + 4 movs).
# -ffp-contract=off: the reference's torch kernels round every mul/add separately; coordinate and
# score arithmetic must do the same (a contracted `start + p*bin` moves a bilinear sample point by
# 1 ulp ~ 1e-5 in the output).  FMAs are written explicitly (fmaf) where they are wanted.
# (tower_wino.hip without -fno-slp-vectorize was measured: the packed adds hipcc then forms in the operand transform make
# both forms of the kernel slower — fp32 main loop 40.6 k -> 42.4 k cycles, bf16 x 3 32.7 k -> 37.9 k)
# (-mllvm -amdgpu-kernarg-preload-count=16 — leading scalar / pointer kernel arguments delivered in SGPRs with the wave
# instead of through an s_load — was measured in round 6: the tower kernels get 4-10 dwords preloaded, kernels whose first
# argument is a struct none; interleaved A/B of two library builds: 52.95 vs 52.75 us per frame pair WITH the flag. Not used.)
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-ffp-contract=off", "--offload-arch=" + ARCH]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _src(name):
    return os.path.join(MEASURE_CSRC if name in DEBUG_ONLY_SOURCES else CSRC, name)


def _deps(sources):
    hdrs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    if any(s in DEBUG_ONLY_SOURCES for s in sources):      # the measurement build also includes measure/csrc/*.inc
        hdrs += [os.path.join(MEASURE_CSRC, f) for f in sorted(os.listdir(MEASURE_CSRC)) if f.endswith(".inc")]
    return [_src(s) for s in sources] + hdrs + [
        os.path.join(os.path.dirname(HERE), "include", "smot_emm.h"), os.path.abspath(__file__)]


def _stale(lib, sources):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps(sources))


def _build_one(lib, sources, extra_flags, objdir, verbose):
    from concurrent.futures import ThreadPoolExecutor
    hipcc = _hipcc()
    os.makedirs(objdir, exist_ok=True)
    lib_t = os.path.getmtime(lib) if os.path.exists(lib) else 0.0
    hdr_t = max(os.path.getmtime(d) for d in _deps([]))
    jobs, objs = [], []
    for s in sources:
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        src = _src(s)
        # per-object staleness: recompile only what changed (a header change recompiles everything)
        if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_t) and lib_t:
            continue
        jobs.append([hipcc] + FLAGS + extra_flags + ["-I", CSRC, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH] + objs + ["-o", lib])
    return lib


def build(force=False, verbose=True, debug=True):
    """Compile every HIP source for gfx950 and link the product library (and, with ``debug``, the measurement
    library next to it).  Returns the product library's path."""
    if force:
        for lib in (LIB, LIB_DEBUG):
            if os.path.exists(lib):
                os.remove(lib)
    if force or _stale(LIB, SOURCES):
        _build_one(LIB, SOURCES, [], os.path.join(CSRC, "obj"), verbose)
    if debug and (force or _stale(LIB_DEBUG, SOURCES + DEBUG_ONLY_SOURCES)):
        _build_one(LIB_DEBUG, SOURCES + DEBUG_ONLY_SOURCES, ["-DSMOT_DEBUG"], os.path.join(CSRC, "obj_debug"), verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, debug="--no-debug" not in sys.argv))
