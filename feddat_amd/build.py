"""Build libfeddat_hip.so (gfx950 only) in-tree with hipcc.  No torch extension machinery: the boundary is
a plain C ABI (include/feddat_hip.h) loaded with ctypes."""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libfeddat_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# MFMA results that feed VALU code right away (softmax, ReLU masks): keep them in VGPRs instead of AGPRs, which saves
# one v_accvgpr_read per accumulator element (attention backward: 240 of its 1160 VALU instructions)
_VGPR_MFMA = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
# adapter.hip: the weight-stationary kernels keep 288 dwords of MFMA A operands in AGPRs; their results feed VALU code
PER_FILE_FLAGS = {"attention.hip": _VGPR_MFMA, "adapter.hip": _VGPR_MFMA}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


ABLATE_LIB = os.path.join(PKG, "libfeddat_hip_ablate.so")
F16_LIB = os.path.join(PKG, "libfeddat_hip_f16.so")


def build(force: bool = False, verbose: bool = False, ablate: bool = False, f16: bool = False, variant: str = "",
          extra_flags=()) -> str:
    """ablate=True: the -DFEDDAT_ABLATE build with the timing-only (wrong-result) probes of tools/ compiled in, written to
    libfeddat_hip_ablate.so; the production library has none of them (csrc/common.hip.h: FD_ABL).
    f16=True: the same sources with IEEE-half operands (-DFEDDAT_OPERANDS_F16: v_mfma_f32_16x16x32_f16, csrc/common.hip.h) ->
    libfeddat_hip_f16.so, the same C ABI (feddat_operand_format() tells them apart).
    variant / extra_flags: an experimental build of either (tools/ A/Bs, e.g. variant="nt", extra_flags=["-DFD_EPI_NT"]) ->
    libfeddat_hip[_f16]_<variant>.so in its own object directory; nothing in feddat_amd/ loads such a library."""
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "feddat_hip.h")]
    objdir = os.path.join(PKG, "build_ablate" if ablate else "build_f16" if f16 else "build")
    LIB = ABLATE_LIB if ablate else F16_LIB if f16 else globals()["LIB"]
    FLAGS = globals()["FLAGS"] + (["-DFEDDAT_ABLATE"] if ablate else []) + (["-DFEDDAT_OPERANDS_F16"] if f16 else [])
    if variant:
        objdir, LIB, FLAGS = objdir + "_" + variant, LIB[:-3] + "_" + variant + ".so", FLAGS + list(extra_flags)
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + PER_FILE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {s}\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def build_all(force: bool = False, verbose: bool = False):
    """Both production libraries: bf16 operands and fp16 operands."""
    return [build(force, verbose), build(force, verbose, f16=True)]


if __name__ == "__main__":
    if "--nt" in sys.argv:      # A/B build: streaming hints on the heavy GEMM epilogues (scripts/ab_r05_nt.sh)
        print(build(force="--force" in sys.argv, verbose=True, f16=True, variant="nt", extra_flags=["-DFD_EPI_NT"]))
    elif "--ablate" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, ablate=True))
    else:
        print(*build_all(force="--force" in sys.argv, verbose=True), sep="\n")
