"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports exactly the entry
points that include/feddat_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "feddat_hip.h")).read()
    return sorted(set(re.findall(r"^(?:int|long) (feddat_[a-z0-9_]+)\(", src, flags=re.M)))


@pytest.mark.parametrize("f16", [False, True])
def test_library_builds_and_exports_every_declared_symbol(f16):
    """Both operand-format builds of the one source tree (bf16: libfeddat_hip.so, fp16: libfeddat_hip_f16.so) export the
    whole header and say which format they are."""
    from feddat_amd import build
    path = build.build(f16=f16)
    assert os.path.exists(path) and path.endswith("_f16.so") == f16
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/feddat_hip.h but not exported"
    assert lib.feddat_abi_version() == 8
    assert lib.feddat_operand_format() == (1 if f16 else 0)
    # the MFMA opcode is the build's operand format's, never the other one's (device code is embedded in the .so)
    import subprocess
    txt = subprocess.run(["strings", path], capture_output=True, text=True).stdout
    assert "gemm_nt_v3_kernel" in txt


def test_binding_switches_libraries_per_thread_block():
    """lib.operands(fmt) binds the calling thread to the library of that operand format and restores the previous one."""
    from feddat_amd import lib
    assert lib.current_operands() == "bf16"
    a = lib.load()
    with lib.operands("f16"):
        b = lib.load()
        assert lib.current_operands() == "f16" and b is not a
        assert b.feddat_operand_format() == lib.OPERANDS_FP16
        with lib.operands("bf16"):
            assert lib.load() is a
        assert lib.load() is b
    assert lib.load() is a and a.feddat_operand_format() == lib.OPERANDS_BF16
    with pytest.raises(lib.FeddatHipError):
        with lib.operands("fp32"):
            pass


def test_python_binding_covers_the_header():
    from feddat_amd import lib
    assert sorted(lib.EXPORTED_SYMBOLS) == _declared()
    lib.load()   # loads with the HIP runtime that torch brought in; raises if the .so is missing


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure; a product path that routes through it voids parity claims."""
    pkg = os.path.join(ROOT, "feddat_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.path.basename(dirpath).startswith("build"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h")):      # the Python host side AND the kernel sources
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle-free", ""), f"{f} mentions the oracle"


def test_ops_fail_loudly_without_a_device():
    import torch
    from feddat_amd import lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    x = torch.zeros(16, 768)
    with pytest.raises(lib.FeddatHipError):
        lib.tanh_fwd(x)


def test_library_does_not_read_the_environment_or_link_rccl():
    """The launch path must not call getenv (ablation flags come in through feddat_set_debug_flags), and RCCL is bound at
    run time only (dlopen), so the single-GPU path has no dependency on it."""
    import subprocess
    from feddat_amd import build
    for path in build.build_all():
        und = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True).stdout
        assert "getenv" not in und
        needed = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
        assert "rccl" not in needed.lower()


def test_production_library_has_no_wrong_result_ablations():
    """The timing-only probes of tools/ (skip-epilogue, no-store, no-compute, the deferred-epilogue GEMM instantiation, ...)
    are compiled out of libfeddat_hip.so (-DFEDDAT_ABLATE builds libfeddat_hip_ablate.so for tools/): the production library
    rejects their flags, keeps the kernel-SELECTION bits (bit-identical results), has no deferred-epilogue kernel, and the
    binding does not switch anything on from the environment."""
    import subprocess
    from feddat_amd import build, lib
    path = build.build()
    L = ctypes.CDLL(path)
    for wrong in (4, 8, 16, 512, 1 << 10, 1 << 16, 1 << 20, 2 << 20, 4 << 20, 1 << 24, 2 << 24, 4 << 24):
        assert L.feddat_set_debug_flags(wrong) == 1, wrong           # FEDDAT_EINVAL
    for ok in (0, 1, 2, 3, 3 | 64, 32, 64, 128, 256, 1 << 23, 8 << 28, 1 | 32):
        assert L.feddat_set_debug_flags(ok) == 0, ok
    L.feddat_set_debug_flags(0)
    # gemm_nt_v3_kernel<EPI, RT, FAKE = 1> is the deferred-epilogue probe: no such instantiation in the production object
    syms = subprocess.run(["strings", path], capture_output=True, text=True).stdout
    assert "gemm_nt_v3_kernelILi0ELi6ELi0ELb0EE" in syms           # <EPI, RT, FAKE = 0, DUAL = false>
    assert "gemm_nt_v3_kernelILi0ELi4ELi0ELb1EE" in syms           # the DUAL form (selection flags 1 | 2; bit-identical results)
    assert "gemm_nt_v3_kernelILi0ELi6ELi1E" not in syms
    src = open(os.path.join(ROOT, "feddat_amd", "lib.py")).read()
    assert "FEDDAT_GEMM_DEBUG" not in src and "environ" not in src
