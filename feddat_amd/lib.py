"""ctypes binding of libfeddat_hip.so (include/feddat_hip.h).

The HIP library is THE product path: if it cannot be loaded this module raises -- there is no eager /
PyTorch / CPU fallback anywhere in feddat_amd.  torch is used only as plumbing: device allocations
(tensor.data_ptr()), the current HIP stream, and torch.distributed for the one RCCL all-reduce per round.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading
from typing import Optional, Sequence

import torch  # imported first on purpose: libfeddat_hip.so must bind to the HIP runtime torch already loaded

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libfeddat_hip.so")
# the same sources and C ABI with IEEE-half operands (include/feddat_hip.h, Conventions): bound by operands("f16")
LIB_PATH_F16 = os.path.join(_PKG, "libfeddat_hip_f16.so")
OPERANDS_BF16, OPERANDS_FP16 = 0, 1      # feddat_operand_format()
OPERAND_DTYPE = {"bf16": torch.bfloat16, "f16": torch.float16}

EPI_BF16, EPI_RESID_F32, EPI_GELU, EPI_MUL_DGELU, EPI_F32, EPI_GELU_G8, EPI_MUL_G8, EPI_GELU_G8_F8, EPI_MUL_G8_F8 = range(9)
F8_ACT_SCALE, F8_GRAD_HEADROOM = 0.125, 4.0      # FEDDAT_F8_ACT_SCALE / FEDDAT_F8_GRAD_HEADROOM
G8_LO, G8_STEP = -0.135, 0.005        # FEDDAT_G8_LO / FEDDAT_G8_STEP: gelu' ~ G8_LO + G8_STEP * code

ABI_VERSION = 8
vp, i32, i64, f32, u32 = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint


class AdapterSeg(C.Structure):
    _fields_ = [("row_begin", i32), ("row_end", i32), ("n_adapters", i32), ("train_slot", i32),
                ("x_row_delta", i32), ("reserved", i32), ("scale", f32 * 2), ("wd", vp * 2), ("wdT", vp * 2), ("wu", vp * 2), ("wuT", vp * 2),
                ("bd", vp * 2), ("bu", vp * 2)]


class WgradSeg(C.Structure):
    _fields_ = [("x", vp), ("dy", vp), ("z", vp), ("dz", vp), ("grad", vp), ("rows", i32), ("scale", f32),
                ("grad_unscale", f32), ("reserved", i32), ("grad_unscale_dev", vp)]


class ViltLayerWeights(C.Structure):
    _fields_ = [(n, vp) for n in ("wqkv", "wo", "w1", "w2", "wqkvT", "woT", "w1T", "w2T", "bqkv", "bo", "b1", "b2", "ln1_g",
                                  "ln1_b", "ln2_g", "ln2_b")] + [("ln_eps", f32)]


class ViltLayerActs(C.Structure):
    _fields_ = [(n, vp) for n in ("h_in", "st1", "qkv", "ctx", "lse", "h2", "st2", "u", "h3", "z_save", "h_out", "x16", "f16",
                                  "st1_next")]


class ViltLayerGrads(C.Structure):
    _fields_ = [(n, vp) for n in ("dh_out", "dh_in", "dh3", "dh16", "dU", "dx16", "dctx", "dqkv", "z", "dz")]


class HtJob(C.Structure):            # feddat_ht_job
    _fields_ = [("A", vp), ("sa_i", i64), ("sa_k", i64), ("B", vp), ("sb_k", i64), ("sb_j", i64), ("I", i32), ("J", i32),
                ("K", i32), ("mode", i32), ("alpha", f32), ("bias_j", vp), ("out", vp), ("ldo", i64), ("colsum", vp),
                ("pro", i32), ("pro_a", vp), ("pro_b", vp), ("pro_eps", f32), ("stats_out", vp), ("epi", i32), ("aux", vp),
                ("ld_aux", i64), ("alpha_dev", vp)]


class AdamwGroup(C.Structure):       # feddat_adamw_group
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("n", i64), ("seg_off", vp), ("seg_wd", vp), ("nseg", i32),
                ("state", vp), ("d_sched", i32), ("d_adam", i32), ("skip_if", vp * 2), ("bak", vp), ("bak_mode", i32),
                ("restore_if", vp)]


HT_PRO_NONE, HT_PRO_LN, HT_PRO_TANH_BWD = 0, 1, 2
HT_EPI_NONE, HT_EPI_TANH, HT_EPI_MUL_DGELU = 0, 1, 2


def _fill(struct, **tensors):
    """ctypes struct of device pointers from tensors (None -> NULL); keeps the tensors alive on the struct."""
    s = struct()
    s._keep = tensors
    for k, t in tensors.items():
        setattr(s, k, (t.data_ptr() if isinstance(t, torch.Tensor) else t) if t is not None else None)
    return s


_SIGS = {
    "feddat_abi_version": [],
    "feddat_operand_format": [],
    "feddat_ctx_create": [i32, C.POINTER(vp)],
    "feddat_ctx_destroy": [vp],
    "feddat_ctx_device": [vp, C.POINTER(i32), C.POINTER(i32)],
    "feddat_set_debug_flags": [i32],
    "feddat_comm_unique_id": [vp],
    "feddat_comm_create": [vp, i32, i32, C.POINTER(vp)],
    "feddat_comm_create_timeout": [vp, i32, i32, i32, C.POINTER(vp)],
    "feddat_comm_destroy": [vp],
    "feddat_fedavg_allreduce": [vp, vp, vp, i64, f32, f32, vp],
    "feddat_comm_info": [vp, vp, vp, vp],
    "feddat_gemm_bf16_nt": [vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp],
    "feddat_gemm_skinny_workspace_elems": [i32, i32, i32],
    "feddat_gemm_dual_blocks_per_cu": [vp],
    "feddat_gemm_fp8_nt": [vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp, i32, vp],
    "feddat_layernorm_bwd_dx_fp8": [vp, vp, i64, vp, i64, vp, vp, vp, i64, i32, i32, vp, i64, vp, vp, vp],
    "feddat_gemm_fp8mx_nt": [vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, vp, vp, i32, vp],
    "feddat_gemm_fp8_nt_f32": [vp, i32, vp, vp, i32, vp, i32, i32, i32, vp, vp, i32, vp, i32, vp],
    "feddat_quant_rows_fp8": [vp, i64, i32, i32, vp, vp, vp],
    "feddat_layernorm_fwd_fp8": [vp, i64, vp, vp, f32, i32, i32, vp, vp, vp, vp, vp],
    "feddat_gemm_bf16_nt_skinny": [vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32,
                                   vp, i64, vp],
    "feddat_attn_fwd": [vp, vp, vp, vp, i32, i32, i32, vp],
    "feddat_attn_bwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "feddat_attn_bwd_fp8mx": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "feddat_attn_cls_fwd": [vp, vp, vp, vp, i32, i32, i32, vp],
    "feddat_attn_cls_bwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "feddat_attn2_fwd": [vp, i64, vp, i64, vp, i64, vp, i32, vp, i64, vp, i32, i32, i32, i64, i64, i32, vp],
    "feddat_attn2_bwd": [vp, i64, vp, i64, vp, i64, vp, i32, vp, i64, vp, vp, i64, vp, vp, i64, vp, i64, vp, i64, i32, i32,
                         i32, i64, i64, i32, vp],
    "feddat_attn2_fwd_dropout": [vp, i64, vp, i64, vp, i64, vp, i32, vp, i64, vp, i32, i32, i32, i64, i64, i32, f32, u32, u32,
                                 vp, vp],
    "feddat_attn2_bwd_dropout": [vp, i64, vp, i64, vp, i64, vp, i32, vp, i64, vp, vp, i64, vp, vp, i64, vp, i64, vp, i64,
                                 i32, i32, i32, i64, i64, i32, f32, u32, u32, vp, vp],
    "feddat_dropout": [vp, vp, vp, vp, vp, i64, f32, u32, u32, vp, vp],
    "feddat_layernorm_fwd": [vp, i64, vp, vp, f32, i32, i32, vp, vp, vp, vp],
    "feddat_layernorm_bwd_dx": [vp, vp, i64, vp, i64, vp, vp, vp, i64, i32, i32, vp, i64, vp, vp],
    "feddat_layernorm_bwd_dx_sparse": [vp, vp, i64, vp, i64, vp, vp, vp, i64, i32, i32, i32, vp, i64, vp, vp],
    "feddat_layernorm_bwd_full": [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp],
    "feddat_adapter_fwd": [vp, vp, i32, i32, i32, C.POINTER(AdapterSeg), i32, vp, vp],
    "feddat_adapter_fwd_ln": [vp, vp, i32, i32, i32, C.POINTER(AdapterSeg), i32, vp, vp, f32, vp, vp, vp, vp],
    "feddat_adapter_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, C.POINTER(AdapterSeg), i32, vp],
    "feddat_adapter_bwd_fp8": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, C.POINTER(AdapterSeg), i32, vp],
    "feddat_adapter_pack": [vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "feddat_adapter_pack_strided": [vp, vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, vp],
    "feddat_adapter_wgrad_workspace_elems": [i32],
    "feddat_adapter_wgrad": [C.POINTER(WgradSeg), i32, vp, i64, i32, i32, vp],
    "feddat_adapter_wgrad_partial": [C.POINTER(WgradSeg), i32, vp, i64, i32, i32, vp],
    "feddat_adapter_wgrad_reduce": [vp, i32, i32, vp, i64, vp],
    "feddat_adapter_wgrad_reduce_checked": [vp, i32, i32, vp, i64, vp, vp],
    "feddat_vilt_layer_fwd": [vp, C.POINTER(ViltLayerWeights), C.POINTER(ViltLayerActs), i32, i32, i32, vp, i32,
                              C.POINTER(AdapterSeg), i32, vp, vp, vp],
    "feddat_vilt_layer_bwd": [vp, C.POINTER(ViltLayerWeights), C.POINTER(ViltLayerActs), C.POINTER(ViltLayerGrads), i32, i32,
                              i32, vp, C.POINTER(AdapterSeg), i32, C.POINTER(WgradSeg), i32, vp, i64, i32, vp],
    "feddat_sgemm_f32": [vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, f32, vp, vp, i64, i64, vp, i64, vp],
    "feddat_reduce_partials": [vp, i64, i32, i64, vp, vp],
    "feddat_dat_loss_fwd_bwd": [vp, vp, vp, i32, i32, f32, vp, vp, vp],
    "feddat_dat_loss_fwd_bwd_single": [vp, vp, vp, i32, i32, f32, vp, vp, vp],
    "feddat_dat_loss_fwd_bwd_checked": [vp, vp, vp, i32, i32, f32, vp, vp, vp, vp],
    "feddat_dat_step_finish": [vp, vp, vp, vp, vp, vp, f32, f32, i32, vp],
    "feddat_head_gemm": [C.POINTER(HtJob), i32, vp],
    "feddat_head_ln_gelu": [vp, vp, vp, f32, i32, i32, vp, vp, vp, vp],
    "feddat_head_ln_bwd_full": [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp],
    "feddat_adamw_multi": [C.POINTER(AdamwGroup), i32, f32, i32, i32, f32, f32, f32, vp],
    "feddat_step_tick_multi": [C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), i32, vp],
    "feddat_vqa_score_accumulate": [vp, vp, i32, i32, vp, vp],
    "feddat_lm_loss_fwd_bwd": [vp, vp, i64, vp, vp, vp, i32, i32, f32, f32, f32, vp, i64, vp, vp],
    "feddat_lm_loss_fwd_bwd_dyn": [vp, vp, i64, vp, vp, vp, i32, i32, f32, f32, f32, vp, vp, vp, i64, vp, vp],
    "feddat_axpby3": [vp, f32, vp, f32, vp, f32, vp, vp, i64, vp],
    "feddat_vilt_stage_inputs": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "feddat_softmax_gather_rows": [vp, i64, i32, i32, vp, i64, i32, vp, vp],
    "feddat_topk_rows": [vp, i64, vp, i32, i32, i32, i32, vp, vp, vp],
    "feddat_gather_rows": [vp, vp, vp, vp, i32, i32, vp],
    "feddat_segment_sum_rows": [vp, vp, vp, i32, i32, i32, vp],
    "feddat_adamw_flat": [vp, vp, vp, vp, i64, vp, vp, i32, vp, f32, i32, i32, f32, f32, f32, vp],
    "feddat_step_tick": [vp, i32, i32, vp],
    "feddat_text_embed": [vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, vp],
    "feddat_im2col_patches": [vp, vp, i32, i32, i32, i32, i32, vp],
    "feddat_image_embed_assemble": [vp, vp, vp, vp, i64, vp, vp, i32, i32, i32, i32, i32, vp],
    "feddat_image_embed_assemble_masked": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "feddat_pos_embed_resize_masked": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "feddat_vilt_key_mask": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "feddat_vilt_image_workspace_bytes": [vp, vp, vp, vp, i32],
    "feddat_vilt_image_preprocess": [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, i64, vp],
    "feddat_pos_embed_resize": [vp, vp, i32, i32, i32, i32, vp],
    "feddat_wordpiece_table_entries": [i32],
    "feddat_wordpiece_table_build": [vp, vp, i32, vp, i64],
    "feddat_wordpiece_encode": [vp, vp, i32, vp, i64, i32, i32, i32, i32, i32, vp, vp, vp, vp],
    "feddat_cvt_f32_bf16": [vp, vp, i64, vp],
    "feddat_transpose_f32_bf16": [vp, vp, i32, i32, vp],
    "feddat_tanh_fwd": [vp, i64, vp],
    "feddat_tanh_bwd": [vp, vp, vp, i64, vp],
    "feddat_gelu_fwd": [vp, vp, i64, vp],
    "feddat_gelu_bwd": [vp, vp, vp, i64, vp],
    "feddat_scatter_cls_rows": [vp, vp, vp, i32, i32, i32, vp],
    "feddat_fedavg_accumulate": [vp, vp, i64, f32, f32, i32, vp],
    "feddat_probe_tr16": [vp, vp, vp],
}

EXPORTED_SYMBOLS = tuple(_SIGS.keys())


class FeddatHipError(RuntimeError):
    pass


_lib = None            # the bf16-operand library (the default binding)
_lib_f16 = None        # the fp16-operand library
_tls = threading.local()


def _open(path: str, want_format: int) -> C.CDLL:
    if not os.path.exists(path):
        raise FeddatHipError(
            f"{path} not found: the HIP library is the only implementation of this path "
            "(no CPU/PyTorch fallback). Build it with `python __graft_entry__.py` or `python -m feddat_amd.build`.")
    lib = C.CDLL(path)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i64 if name.endswith(("_workspace_elems", "_workspace_bytes", "_table_entries")) else i32
    if lib.feddat_abi_version() != ABI_VERSION:
        raise FeddatHipError(f"{os.path.basename(path)} ABI version mismatch")
    if lib.feddat_operand_format() != want_format:
        raise FeddatHipError(f"{os.path.basename(path)} was built for another operand format")
    return lib


def load() -> C.CDLL:
    """The library the calling thread is bound to: libfeddat_hip.so (bf16 operands) unless inside `with operands("f16")`.
    Raises loudly if it is missing (build with `python -m feddat_amd.build`)."""
    global _lib, _lib_f16
    if getattr(_tls, "fmt", "bf16") == "f16":
        if _lib_f16 is None:
            _lib_f16 = _open(LIB_PATH_F16, OPERANDS_FP16)
        return _lib_f16
    if _lib is None:
        _lib = _open(LIB_PATH, OPERANDS_BF16)
    return _lib


@contextlib.contextmanager
def operands(fmt: str):
    """Bind this thread's calls to the library built for the 16-bit operand format `fmt` ("bf16" | "f16") for the duration of
    the block.  Both libraries export the same C ABI; buffers the header calls "bf16" hold OPERAND_DTYPE[fmt] values.  An
    engine makes all its calls (and creates its feddat_ctx) inside the block of ITS format, so engines of both formats can
    live in one process."""
    if fmt not in OPERAND_DTYPE:
        raise FeddatHipError(f"unknown operand format {fmt!r} (bf16 | f16)")
    prev = getattr(_tls, "fmt", "bf16")
    _tls.fmt = fmt
    try:
        yield
    finally:
        _tls.fmt = prev


def current_operands() -> str:
    return getattr(_tls, "fmt", "bf16")


def use_ablation_build():
    """tools/ only, and only before the first load(): bind libfeddat_hip_ablate.so (`python -m feddat_amd.build --ablate`), the
    -DFEDDAT_ABLATE build that contains the timing-only probes (wrong results).  Nothing in feddat_amd/, bench.py or tests/
    calls this; the production library rejects those flags (feddat_set_debug_flags -> EINVAL)."""
    global LIB_PATH
    if _lib is not None:
        raise FeddatHipError("use_ablation_build() must be called before the library is loaded")
    LIB_PATH = os.path.join(_PKG, "libfeddat_hip_ablate.so")


def set_debug_flags(flags: int):
    """Kernel-selection switches (bit-identical results): 1 / 2 GEMMs on the two-group / one-wave-per-SIMD kernel, 32 / 64
    force 192- / 256-row tiles, 128 no small-tile kernel, 256 K = 32 fp8 MFMA, bit 23 one attention-backward block per pair,
    bits 28..31 cap the persistent GEMM grid.  The timing-only ablations (8 skip epilogue, 4 / 16, 512, bits 8..22, 24..26)
    exist in the ablation build only (use_ablation_build)."""
    _chk(load().feddat_set_debug_flags(int(flags)), "feddat_set_debug_flags")


class Context:
    """feddat_ctx: explicit per-device handle (sets every kernel's launch attributes on that device at creation)."""

    def __init__(self, device: int):
        self._h = vp()
        self._lib = load()          # a handle belongs to the library (operand format) it was made by
        _chk(self._lib.feddat_ctx_create(int(device), C.byref(self._h)), "feddat_ctx_create")

    def info(self):
        d, cu = i32(), i32()
        _chk(self._lib.feddat_ctx_device(self._h, C.byref(d), C.byref(cu)), "feddat_ctx_device")
        return d.value, cu.value

    def close(self):
        if self._h:
            self._lib.feddat_ctx_destroy(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gemm_dual_blocks_per_cu() -> int:
    """Workgroups per CU the runtime grants the dual form of the persistent GEMM (diagnostics)."""
    n = C.c_int(0)
    _chk(load().feddat_gemm_dual_blocks_per_cu(C.byref(n)), "feddat_gemm_dual_blocks_per_cu")
    return n.value


def rccl_available() -> bool:
    """True when the library could bind RCCL in this process (feddat_comm_info without a communicator: version only)."""
    v = C.c_int(0)
    return load().feddat_comm_info(None, C.byref(v), None, None) == 0      # (ELAUNCH when RCCL could not be bound; the version
    #                                                                          symbol itself is optional: diagnostics only)


class RcclComm:
    """ncclComm_t made through the C ABI (feddat_comm_*): rank 0 draws the unique id, `exchange(id_bytes) -> id_bytes`
    ships it to the other ranks over any host channel (torch.distributed object broadcast, a TCPStore, MPI, a file)."""

    def __init__(self, world: int, rank: int, exchange, timeout_s: float = 120.0):
        """timeout_s: watchdog on the collective ncclCommInitRank (feddat_comm_create_timeout): a peer that died or could not
        load RCCL makes this raise after timeout_s instead of hanging; 0 = wait forever."""
        buf = C.create_string_buffer(128)
        self._lib = load()
        rc0 = self._lib.feddat_comm_unique_id(buf) if rank == 0 else 0
        # rank 0 ships its return code with the id, so that a failure there (RCCL not loadable) raises on EVERY rank instead of
        # leaving the others waiting in the exchange
        msg = exchange(bytes([0 if rc0 == 0 else 1]) + bytes(buf.raw))
        if msg[0] != 0:
            raise FeddatHipError("feddat_comm_unique_id failed on rank 0 (is librccl loadable?)")
        ident = msg[1:]
        self._h = vp()
        _chk(self._lib.feddat_comm_create_timeout(C.c_char_p(ident), world, rank, int(timeout_s * 1000), C.byref(self._h)),
             "feddat_comm_create_timeout")
        self.world, self.rank = world, rank

    def fedavg_allreduce(self, flat, scratch, num: float, total: float):
        _dev(flat, scratch)
        _chk(self._lib.feddat_fedavg_allreduce(self._h, _p(flat), _p(scratch), flat.numel(), float(num), float(total),
                                            _stream()), "feddat_fedavg_allreduce")

    def info(self):
        """{'rccl_version': ncclGetVersion code, 'ranks': ranks of the communicator, 'rank': this rank} from the library."""
        v, n, r = C.c_int(0), C.c_int(0), C.c_int(0)
        _chk(self._lib.feddat_comm_info(self._h, C.byref(v), C.byref(n), C.byref(r)), "feddat_comm_info")
        return {"rccl_version": v.value, "ranks": n.value, "rank": r.value}

    def close(self):
        if self._h:
            self._lib.feddat_comm_destroy(self._h)
            self._h = vp()


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(rc: int, what: str):
    if rc != 0:
        raise FeddatHipError(f"{what} failed with code {rc} ({ {1: 'EINVAL', 2: 'ELAUNCH', 3: 'ETIMEOUT'}.get(rc, '?') })")


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise FeddatHipError("feddat_amd ops need device (HIP) tensors; there is no CPU path")


# ---------------------------------------------------------------------------------------------------
# thin typed wrappers (tensors in, tensors out; no arithmetic happens on the Python side)
# ---------------------------------------------------------------------------------------------------
def gemm_skinny_workspace_elems(M: int, N: int, K: int) -> int:
    return int(load().feddat_gemm_skinny_workspace_elems(M, N, K))


def gemm_bf16_nt(A, B, epi, *, bias=None, resid=None, aux=None, out_f32=None, out_bf16=None, out2_bf16=None,
                 M=None, skinny_workspace=None):
    """C[M,N] = A[M,K] @ B[N,K]^T (+ epilogue).  A, B bf16 2-D (row stride taken from the tensors).
    With M <= 64 and a fp32 `skinny_workspace` the split-K skinny kernel pair is used."""
    _dev(A, B)
    M = A.shape[0] if M is None else M
    K = A.shape[1]
    N = B.shape[0]

    def ld(t):
        return 0 if t is None else t.stride(0)
    if skinny_workspace is not None and M <= 64:
        _dev(skinny_workspace)
        rc = load().feddat_gemm_bf16_nt_skinny(_p(A), A.stride(0), _p(B), B.stride(0), M, N, K, epi, _p(bias),
                                               _p(resid), ld(resid), _p(aux), ld(aux), _p(out_f32), ld(out_f32),
                                               _p(out_bf16), ld(out_bf16), _p(out2_bf16), ld(out2_bf16),
                                               _p(skinny_workspace), skinny_workspace.numel(), _stream())
        _chk(rc, "feddat_gemm_bf16_nt_skinny")
        return
    rc = load().feddat_gemm_bf16_nt(_p(A), A.stride(0), _p(B), B.stride(0), M, N, K, epi, _p(bias), _p(resid),
                                    ld(resid), _p(aux), ld(aux), _p(out_f32), ld(out_f32), _p(out_bf16),
                                    ld(out_bf16), _p(out2_bf16), ld(out2_bf16), _stream())
    _chk(rc, "feddat_gemm_bf16_nt")


def gemm_fp8_nt(A8, a_scale, B8, b_scale, epi, *, bias=None, aux=None, out_bf16=None, out2_bf16=None):
    """C = (A8 @ B8^T) * a_scale[:, None] * b_scale[None, :] (+ bias; epilogue EPI_BF16, EPI_GELU or EPI_MUL_DGELU with aux);
    A8 / B8: e4m3 bytes as uint8 / float8 tensors [M,K] / [N,K]."""
    _dev(A8, B8, a_scale, b_scale, out_bf16, aux)
    M, K = A8.shape
    N = B8.shape[0]
    _chk(load().feddat_gemm_fp8_nt(_p(A8), A8.stride(0), _p(a_scale), _p(B8), B8.stride(0), _p(b_scale), M, N, K, epi,
                                   _p(bias), _p(aux), 0 if aux is None else aux.stride(0), _p(out_bf16), out_bf16.stride(0),
                                   _p(out2_bf16),
                                   0 if out2_bf16 is None else out2_bf16.stride(0), _stream()), "feddat_gemm_fp8_nt")


def gemm_fp8mx_nt(A8, a_mx, B8, b_scale, *, bias=None, out_bf16=None):
    """out = (A . B^T) * b_scale[None, :] + bias with A = A8 (e4m3) * 2^(a_mx - 127) per (row, 32-column block): a_mx uint8
    [M, K / 32] E8M0 block scales applied by the MFMA itself (feddat_gemm_fp8mx_nt)."""
    _dev(A8, a_mx, B8, b_scale, out_bf16)
    M, K = A8.shape
    N = B8.shape[0]
    _chk(load().feddat_gemm_fp8mx_nt(_p(A8), A8.stride(0), _p(a_mx), a_mx.stride(0), _p(B8), B8.stride(0), _p(b_scale), M, N, K,
                                     _p(bias), _p(out_bf16), out_bf16.stride(0), _stream()), "feddat_gemm_fp8mx_nt")


def gemm_fp8_nt_f32(A8, a_scale, B8, b_scale, *, bias=None, resid=None, out_f32=None):
    """out_f32 = (A8 @ B8^T) * a_scale[:, None] * b_scale[None, :] + bias (+ resid)."""
    _dev(A8, B8, a_scale, b_scale, out_f32, resid, bias)
    M, K = A8.shape
    N = B8.shape[0]
    _chk(load().feddat_gemm_fp8_nt_f32(_p(A8), A8.stride(0), _p(a_scale), _p(B8), B8.stride(0), _p(b_scale), M, N, K, _p(bias),
                                       _p(resid), 0 if resid is None else resid.stride(0), _p(out_f32), out_f32.stride(0),
                                       _stream()), "feddat_gemm_fp8_nt_f32")


def quant_rows_fp8(x, y8, scale):
    _dev(x, y8, scale)
    _chk(load().feddat_quant_rows_fp8(_p(x), x.stride(0), x.shape[0], x.shape[1], _p(y8), _p(scale), _stream()),
         "feddat_quant_rows_fp8")


def layernorm_fwd_fp8(x, gamma, beta, eps, rows, H, y_fp8, y_scale, *, y_bf16=None, stats=None, x_stride=None):
    _dev(x, y_fp8, y_scale)
    _chk(load().feddat_layernorm_fwd_fp8(_p(x), H if x_stride is None else x_stride, _p(gamma), _p(beta), eps, rows, H,
                                         _p(y_fp8), _p(y_scale), _p(y_bf16), _p(stats), _stream()), "feddat_layernorm_fwd_fp8")


def attn_fwd(qkv, ctx, lse, B, S, heads, key_mask=None):
    _dev(qkv, ctx)
    _chk(load().feddat_attn_fwd(_p(qkv), _p(key_mask), _p(ctx), _p(lse), B, S, heads, _stream()), "feddat_attn_fwd")


def attn_bwd(qkv, ctx, lse, dctx, dqkv, B, S, heads, key_mask=None):
    _dev(qkv, ctx, dctx, dqkv)
    _chk(load().feddat_attn_bwd(_p(qkv), _p(key_mask), _p(ctx), _p(lse), _p(dctx), _p(dqkv), B, S, heads, _stream()),
         "feddat_attn_bwd")


def attn_bwd_fp8mx(qkv, ctx, lse, dctx, dq8, dq_scale, B, S, heads, key_mask=None):
    """attn_bwd with dq | dk | dv as MX-scaled e4m3: dq8 uint8 [B S, 3 H], dq_scale uint8 [B S, 3 H / 32] (E8M0 per 32 columns)."""
    _dev(qkv, ctx, dctx, dq8, dq_scale)
    _chk(load().feddat_attn_bwd_fp8mx(_p(qkv), _p(key_mask), _p(ctx), _p(lse), _p(dctx), _p(dq8), _p(dq_scale), B, S, heads,
                                      _stream()), "feddat_attn_bwd_fp8mx")


def attn_cls_fwd(qkv, ctx, lse, B, S, heads, key_mask=None):
    """Attention for token 0 of every sample only (the last layer): writes ctx rows b*S and lse[b, h, 0]."""
    _dev(qkv, ctx)
    _chk(load().feddat_attn_cls_fwd(_p(qkv), _p(key_mask), _p(ctx), _p(lse), B, S, heads, _stream()), "feddat_attn_cls_fwd")


def attn_cls_bwd(qkv, ctx, lse, dctx0, dqkv, B, S, heads, key_mask=None):
    """dctx0: fp32 [B, H] gradient of the token-0 context rows -> the complete dqkv."""
    _dev(qkv, ctx, dctx0, dqkv)
    assert dctx0.dtype == torch.float32 and dctx0.is_contiguous()
    _chk(load().feddat_attn_cls_bwd(_p(qkv), _p(key_mask), _p(ctx), _p(lse), _p(dctx0), _p(dqkv), B, S, heads, _stream()),
         "feddat_attn_cls_bwd")


def attn2_fwd(q, k, v, ctx, lse, B, Sq, Skv, heads, *, key_mask=None, causal=False, q_rows=None, kv_rows=None, drop=None):
    """General attention: q [B*q_rows, >=heads*64] / k, v [B*kv_rows, ...] bf16 2-D views (row strides from the tensors).
    drop = (p, key0, key1, step_counter) applies dropout to the attention probabilities (xbert.py:333)."""
    _dev(q, k, v, ctx)
    if drop is not None and drop[0] > 0:
        _chk(load().feddat_attn2_fwd_dropout(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(key_mask),
                                             int(causal), _p(ctx), ctx.stride(0), _p(lse), B, Sq, Skv,
                                             Sq if q_rows is None else q_rows, Skv if kv_rows is None else kv_rows, heads,
                                             drop[0], drop[1], drop[2], _p(drop[3]), _stream()), "feddat_attn2_fwd_dropout")
        return
    _chk(load().feddat_attn2_fwd(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(key_mask), int(causal),
                                 _p(ctx), ctx.stride(0), _p(lse), B, Sq, Skv, Sq if q_rows is None else q_rows,
                                 Skv if kv_rows is None else kv_rows, heads, _stream()), "feddat_attn2_fwd")


def attn2_bwd(q, k, v, ctx, lse, dctx, dsum_ws, dq, dk, dv, B, Sq, Skv, heads, *, key_mask=None, causal=False,
              q_rows=None, kv_rows=None, drop=None):
    _dev(q, k, v, ctx, dctx, dq, dk, dv)
    if drop is not None and drop[0] > 0:
        _chk(load().feddat_attn2_bwd_dropout(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(key_mask),
                                             int(causal), _p(ctx), ctx.stride(0), _p(lse), _p(dctx), dctx.stride(0),
                                             _p(dsum_ws), _p(dq), dq.stride(0), _p(dk), dk.stride(0), _p(dv), dv.stride(0), B, Sq,
                                             Skv, Sq if q_rows is None else q_rows, Skv if kv_rows is None else kv_rows, heads,
                                             drop[0], drop[1], drop[2], _p(drop[3]), _stream()), "feddat_attn2_bwd_dropout")
        return
    _chk(load().feddat_attn2_bwd(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(key_mask), int(causal),
                                 _p(ctx), ctx.stride(0), _p(lse), _p(dctx), dctx.stride(0), _p(dsum_ws), _p(dq),
                                 dq.stride(0), _p(dk), dk.stride(0), _p(dv), dv.stride(0), B, Sq, Skv,
                                 Sq if q_rows is None else q_rows, Skv if kv_rows is None else kv_rows, heads, _stream()),
         "feddat_attn2_bwd")


def layernorm_fwd(x, gamma, beta, eps, rows, H, *, x_stride=None, y_bf16=None, y_f32=None, stats=None):
    _dev(x)
    _chk(load().feddat_layernorm_fwd(_p(x), H if x_stride is None else x_stride, _p(gamma), _p(beta), eps, rows, H,
                                     _p(y_bf16), _p(y_f32), _p(stats), _stream()), "feddat_layernorm_fwd")


def layernorm_bwd_dx(x, stats, gamma, rows, H, *, dy_bf16=None, dy_f32=None, dy_stride=None, x_stride=None,
                     dres=None, dres_stride=None, out_f32=None, out_stride=None, out_bf16=None, dres_every=0):
    """dres_every = E > 0: dres is compact, its row r / E belongs to output row r for r % E == 0 (feddat_layernorm_bwd_dx_sparse)."""
    _dev(x)
    if dres_every:
        _dev(dres)
        _chk(load().feddat_layernorm_bwd_dx_sparse(_p(dy_bf16), _p(dy_f32), H if dy_stride is None else dy_stride, _p(x),
                                                   H if x_stride is None else x_stride, _p(stats), _p(gamma), _p(dres),
                                                   H if dres_stride is None else dres_stride, int(dres_every), rows, H,
                                                   _p(out_f32), H if out_stride is None else out_stride, _p(out_bf16), _stream()),
             "feddat_layernorm_bwd_dx_sparse")
        return
    _chk(load().feddat_layernorm_bwd_dx(_p(dy_bf16), _p(dy_f32), H if dy_stride is None else dy_stride, _p(x),
                                        H if x_stride is None else x_stride, _p(stats), _p(gamma), _p(dres),
                                        H if dres_stride is None else dres_stride, rows, H, _p(out_f32),
                                        H if out_stride is None else out_stride, _p(out_bf16), _stream()),
         "feddat_layernorm_bwd_dx")


def layernorm_bwd_dx_fp8(x, stats, gamma, rows, H, out_fp8, out_scale, *, dy_bf16=None, dy_f32=None, dres=None, out_f32=None,
                         out_bf16=None):
    """layernorm_bwd_dx whose result also leaves as e4m3 rows + per-row scale (the A operand of an fp8 dX product)."""
    _dev(x, out_fp8, out_scale)
    _chk(load().feddat_layernorm_bwd_dx_fp8(_p(dy_bf16), _p(dy_f32), H, _p(x), H, _p(stats), _p(gamma), _p(dres), H, rows, H,
                                            _p(out_f32), H, _p(out_bf16), _p(out_fp8), _p(out_scale), _stream()),
         "feddat_layernorm_bwd_dx_fp8")


def layernorm_bwd_full(dy, x, stats, gamma, rows, H, dx, dgamma, dbeta):
    _dev(dy, x)
    _chk(load().feddat_layernorm_bwd_full(_p(dy), _p(x), _p(stats), _p(gamma), rows, H, _p(dx), _p(dgamma),
                                          _p(dbeta), _stream()), "feddat_layernorm_bwd_full")


def make_segs(segs: Sequence[dict]):
    arr = (AdapterSeg * len(segs))()
    for s, d in zip(arr, segs):
        s.row_begin, s.row_end = d["row_begin"], d["row_end"]
        ads = d["adapters"]  # list of dicts with wd, wdT, wu, wuT (bf16), bd, bu (fp32), scale
        s.n_adapters = len(ads)
        s.train_slot = d.get("train_slot", -1)
        s.x_row_delta = d.get("x_row_delta", 0)
        for a, ad in enumerate(ads):
            s.scale[a] = ad["scale"]
            s.wd[a] = ad["wd"].data_ptr()
            s.wu[a] = ad["wu"].data_ptr()
            s.wdT[a] = ad["wdT"].data_ptr() if ad.get("wdT") is not None else None
            s.wuT[a] = ad["wuT"].data_ptr() if ad.get("wuT") is not None else None
            s.bd[a] = ad["bd"].data_ptr()
            s.bu[a] = ad["bu"].data_ptr()
    return arr


def adapter_fwd(x, out, segs_arr, T, H=768, r=48, z_save=None):
    """z_save: optional fp32 [T, 2, r] receiving relu(W_down x + b_down) per adapter slot (for adapter_bwd's z_saved)."""
    _dev(x, out, z_save)
    _chk(load().feddat_adapter_fwd(_p(x), _p(out), T, H, r, segs_arr, len(segs_arr), _p(z_save), _stream()),
         "feddat_adapter_fwd")


def adapter_fwd_ln(x, out, segs_arr, T, gamma, beta, eps, y_bf16, stats=None, H=768, r=48, z_save=None):
    """adapter_fwd + the next layer's LayerNorm of the output rows (bf16 y, [T,2] stats)."""
    _dev(x, out, y_bf16, z_save)
    _chk(load().feddat_adapter_fwd_ln(_p(x), _p(out), T, H, r, segs_arr, len(segs_arr), _p(gamma), _p(beta), eps,
                                      _p(y_bf16), _p(stats), _p(z_save), _stream()), "feddat_adapter_fwd_ln")


def adapter_bwd(x, dy, dx, segs_arr, T, *, dx_bf16=None, z_out=None, dz_out=None, z_saved=None, H=768, r=48):
    """x may be None when z_saved (the forward's z_save) is given: the backward then does not read x at all."""
    _dev(x, dy, dx, z_saved)
    if x is None and z_saved is None:
        raise FeddatHipError("adapter_bwd needs x or z_saved")
    _chk(load().feddat_adapter_bwd(_p(x), _p(z_saved), _p(dy), _p(dx), _p(dx_bf16), _p(z_out), _p(dz_out), T, H, r,
                                   segs_arr, len(segs_arr), _stream()), "feddat_adapter_bwd")


def adapter_bwd_fp8(dy, dx, dx_fp8, dx_scale, segs_arr, T, *, z_saved, z_out=None, dz_out=None, H=768, r=48):
    """adapter_bwd (saved-z form) whose dx also leaves as e4m3 rows + per-row scale."""
    _dev(dy, dx, dx_fp8, dx_scale, z_saved)
    _chk(load().feddat_adapter_bwd_fp8(_p(z_saved), _p(dy), _p(dx), _p(dx_fp8), _p(dx_scale), _p(z_out), _p(dz_out), T, H, r,
                                       segs_arr, len(segs_arr), _stream()), "feddat_adapter_bwd_fp8")


def make_wgrad_segs(segs: Sequence[dict]):
    arr = (WgradSeg * len(segs))()
    for s, d in zip(arr, segs):
        s.x, s.dy, s.z, s.dz, s.grad = (d[k].data_ptr() for k in ("x", "dy", "z", "dz", "grad"))
        s.rows, s.scale, s.grad_unscale = d["rows"], d["scale"], d.get("grad_unscale", 1.0)
        t = d.get("grad_unscale_dev")          # device float: 1 / the dynamic loss scale (ABI 8)
        s.grad_unscale_dev = t.data_ptr() if t is not None else None
    arr._keep = [d.get("grad_unscale_dev") for d in segs]
    return arr


def adapter_wgrad_workspace_elems(nseg: int) -> int:
    return int(load().feddat_adapter_wgrad_workspace_elems(nseg))


def adapter_wgrad_partial(segs_arr, partials, H=768, r=48):
    _dev(partials)
    _chk(load().feddat_adapter_wgrad_partial(segs_arr, len(segs_arr), _p(partials), partials.numel(), H, r, _stream()),
         "feddat_adapter_wgrad_partial")


def adapter_wgrad_reduce(grads_dev, n, nseg, partials, stride):
    """grads_dev: int64 device tensor [n * nseg] of gradient-buffer addresses; launch l's partials at partials[l * stride:]."""
    _dev(grads_dev, partials)
    _chk(load().feddat_adapter_wgrad_reduce(_p(grads_dev), n, nseg, _p(partials), stride, _stream()),
         "feddat_adapter_wgrad_reduce")


def adapter_wgrad_reduce_checked(grads_dev, n, nseg, partials, stride, nonfinite):
    """feddat_adapter_wgrad_reduce + GradScaler's inf check: nonfinite (int32 device tensor, >= nseg elements) is OR-ed per segment."""
    _dev(grads_dev, partials, nonfinite)
    assert nonfinite.dtype == torch.int32 and nonfinite.numel() >= nseg
    _chk(load().feddat_adapter_wgrad_reduce_checked(_p(grads_dev), n, nseg, _p(partials), stride, _p(nonfinite), _stream()),
         "feddat_adapter_wgrad_reduce_checked")


def adapter_wgrad(segs_arr, partials, H=768, r=48):
    _dev(partials)
    _chk(load().feddat_adapter_wgrad(segs_arr, len(segs_arr), _p(partials), partials.numel(), H, r, _stream()),
         "feddat_adapter_wgrad")


def adapter_pack_strided(wd, wu, stride32, wd16, wdT16, wu16, wuT16, stride16, n, H=768, r=48):
    _dev(wd, wu, wd16)
    _chk(load().feddat_adapter_pack_strided(_p(wd), _p(wu), stride32, _p(wd16), _p(wdT16), _p(wu16), _p(wuT16), stride16,
                                            n, H, r, _stream()), "feddat_adapter_pack_strided")


def adapter_pack(wd, wu, wd16, wdT16, wu16, wuT16, H=768, r=48):
    _dev(wd, wu)
    _chk(load().feddat_adapter_pack(_p(wd), _p(wu), _p(wd16), _p(wdT16), _p(wu16), _p(wuT16), H, r, _stream()),
         "feddat_adapter_pack")


def vilt_layer_fwd(ctx: "Context", W, A, nb, S, heads, segs_arr, *, key_mask=None, ln1_done=False, next_ln_g=None,
                   next_ln_b=None):
    """One ViltLayer + Adaptered_ViltOutput forward as ONE C-ABI call (W / A: ViltLayerWeights / ViltLayerActs structs)."""
    _chk(load().feddat_vilt_layer_fwd(ctx._h, C.byref(W), C.byref(A), nb, S, heads, _p(key_mask), int(ln1_done), segs_arr,
                                      len(segs_arr), _p(next_ln_g), _p(next_ln_b), _stream()), "feddat_vilt_layer_fwd")


def vilt_layer_bwd(ctx: "Context", W, A, G, nb, S, heads, segs_arr, wsegs_arr, partials, *, key_mask=None,
                   wgrad_reduce_now=True):
    _chk(load().feddat_vilt_layer_bwd(ctx._h, C.byref(W), C.byref(A), C.byref(G), nb, S, heads, _p(key_mask), segs_arr,
                                      len(segs_arr), wsegs_arr, 0 if wsegs_arr is None else len(wsegs_arr), _p(partials),
                                      0 if partials is None else partials.numel(), int(wgrad_reduce_now), _stream()),
         "feddat_vilt_layer_bwd")


def sgemm_f32(A, sa_i, sa_k, B, sb_k, sb_j, I, J, K, out, *, ldo=None, ksplit=1, alpha=1.0, bias_j=None,
              out_split_stride=0, colsum=None, colsum_split_stride=None):
    _dev(A, B, out)
    _chk(load().feddat_sgemm_f32(_p(A), sa_i, sa_k, _p(B), sb_k, sb_j, I, J, K, ksplit, alpha, _p(bias_j), _p(out),
                                 J if ldo is None else ldo, out_split_stride, _p(colsum),
                                 I if colsum_split_stride is None else colsum_split_stride, _stream()),
         "feddat_sgemm_f32")


def reduce_partials(inp, stride, nsplit, n, out):
    _dev(inp, out)
    _chk(load().feddat_reduce_partials(_p(inp), stride, nsplit, n, _p(out), _stream()), "feddat_reduce_partials")


def dat_loss_fwd_bwd(logits, teacher, target, dlogits, scalars, temp=3.0):
    _dev(logits, teacher, target)
    B, Cn = logits.shape
    assert scalars.numel() >= 4 + 2 * B
    _chk(load().feddat_dat_loss_fwd_bwd(_p(logits), _p(teacher), _p(target), B, Cn, temp, _p(dlogits), _p(scalars),
                                        _stream()), "feddat_dat_loss_fwd_bwd")


def ht_job(A, sa_i, sa_k, B, sb_k, sb_j, I, J, K, out, ldo=None, mode=0, alpha=1.0, bias_j=None, colsum=None, pro=HT_PRO_NONE,
           pro_a=None, pro_b=None, pro_eps=0.0, stats_out=None, epi=HT_EPI_NONE, aux=None, ld_aux=0, alpha_dev=None) -> HtJob:
    """One product of feddat_head_gemm (include/feddat_hip.h: feddat_ht_job); keeps its tensors alive."""
    _dev(A, B, out)
    j = HtJob()
    j._keep = (A, B, out, bias_j, colsum, pro_a, pro_b, stats_out, aux, alpha_dev)
    ptr = lambda t: t.data_ptr() if t is not None else None
    j.A, j.sa_i, j.sa_k, j.B, j.sb_k, j.sb_j = ptr(A), sa_i, sa_k, ptr(B), sb_k, sb_j
    j.I, j.J, j.K, j.mode, j.alpha = I, J, K, mode, alpha
    j.bias_j, j.out, j.ldo, j.colsum = ptr(bias_j), ptr(out), (J if ldo is None else ldo), ptr(colsum)
    j.pro, j.pro_a, j.pro_b, j.pro_eps, j.stats_out = pro, ptr(pro_a), ptr(pro_b), pro_eps, ptr(stats_out)
    j.epi, j.aux, j.ld_aux, j.alpha_dev = epi, ptr(aux), ld_aux, ptr(alpha_dev)
    return j


def head_gemm(*jobs: HtJob):
    """One launch for one or two independent small fp32 products (feddat_head_gemm)."""
    arr = (HtJob * len(jobs))(*jobs)
    arr._keep = jobs
    _chk(load().feddat_head_gemm(arr, len(jobs), _stream()), "feddat_head_gemm")


def head_ln_gelu(x, gamma, beta, eps, y, stats, gelu_out):
    _dev(x, y, stats, gelu_out)
    _chk(load().feddat_head_ln_gelu(_p(x), _p(gamma), _p(beta), eps, x.shape[0], x.shape[1], _p(y), _p(stats), _p(gelu_out),
                                    _stream()), "feddat_head_ln_gelu")


def head_ln_bwd_full(dy, x, stats, gamma, dx, dgamma, dbeta):
    _dev(dy, x, stats, dx)
    _chk(load().feddat_head_ln_bwd_full(_p(dy), _p(x), _p(stats), _p(gamma), x.shape[0], x.shape[1], _p(dx), _p(dgamma),
                                        _p(dbeta), _stream()), "feddat_head_ln_bwd_full")


def dat_loss_fwd_bwd_single(logits, teacher, target, dlogits, scalars, temp=3.0):
    _dev(logits, teacher, target)
    B, Cn = logits.shape
    assert scalars.numel() >= 4
    _chk(load().feddat_dat_loss_fwd_bwd_single(_p(logits), _p(teacher), _p(target), B, Cn, temp, _p(dlogits), _p(scalars),
                                               _stream()), "feddat_dat_loss_fwd_bwd_single")


def dat_loss_fwd_bwd_checked(logits, teacher, target, dlogits, scalars, nonfinite, temp=3.0):
    """dat_loss_fwd_bwd_single that also ORs 1 into nonfinite[0] (int32 device tensor) when the loss is inf / NaN."""
    _dev(logits, teacher, target, nonfinite)
    B, Cn = logits.shape
    assert scalars.numel() >= 4 and nonfinite.dtype == torch.int32
    _chk(load().feddat_dat_loss_fwd_bwd_checked(_p(logits), _p(teacher), _p(target), B, Cn, temp, _p(dlogits), _p(scalars),
                                                _p(nonfinite), _stream()), "feddat_dat_loss_fwd_bwd_checked")


def dat_step_finish(head_state, ad1_state, ad0_state, flags, scaler_f, scaler_i, growth=2.0, backoff=0.5, growth_interval=2000):
    """End of a dat train_step under the dynamic loss scale (include/feddat_hip.h: feddat_dat_step_finish)."""
    _dev(head_state, ad1_state, ad0_state, flags, scaler_f, scaler_i)
    _chk(load().feddat_dat_step_finish(_p(head_state), _p(ad1_state), _p(ad0_state), _p(flags), _p(scaler_f), _p(scaler_i),
                                       growth, backoff, growth_interval, _stream()), "feddat_dat_step_finish")


def adamw_group(p, g, m, v, seg_off, seg_wd, state, d_sched=0, d_adam=0, skip_if=(), bak=None, bak_mode=0,
                restore_if=None) -> AdamwGroup:
    """skip_if: up to two int32 device tensors (the update is skipped when either is non-zero); bak / bak_mode / restore_if: the
    save (1) / conditional restore (2) of p | m | v around an update that may have to be undone (feddat_adamw_group, ABI 8)."""
    _dev(p, g, m, v, seg_off, seg_wd, state)
    G = AdamwGroup()
    G._keep = (p, g, m, v, seg_off, seg_wd, state, tuple(skip_if), bak, restore_if)
    for k, t in enumerate(skip_if):
        G.skip_if[k] = t.data_ptr()
    if bak is not None:
        assert bak.numel() >= 3 * p.numel() and bak_mode in (1, 2)
        G.bak, G.bak_mode = bak.data_ptr(), bak_mode
    if restore_if is not None:
        G.restore_if = restore_if.data_ptr()
    G.p, G.g, G.m, G.v, G.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
    G.seg_off, G.seg_wd, G.nseg, G.state, G.d_sched, G.d_adam = (seg_off.data_ptr(), seg_wd.data_ptr(), seg_wd.numel(),
                                                                   state.data_ptr(), d_sched, d_adam)
    return G


def adamw_multi(groups, lr, warmup, total, beta1, beta2, eps):
    arr = (AdamwGroup * len(groups))(*groups)
    arr._keep = groups
    _chk(load().feddat_adamw_multi(arr, len(groups), lr, warmup, total, beta1, beta2, eps, _stream()), "feddat_adamw_multi")


def step_tick_multi(states, d_sched, d_adam):
    n = len(states)
    _dev(*states)
    sp = (vp * n)(*[s.data_ptr() for s in states])
    _chk(load().feddat_step_tick_multi(sp, (i32 * n)(*d_sched), (i32 * n)(*d_adam), n, _stream()), "feddat_step_tick_multi")


def vqa_score_accumulate(logits, target, acc):
    """acc[0] += sum_b target[b, argmax logits[b]], acc[1] += B (device; train_vqa_crossvqa.py:241-257)."""
    _dev(logits, target, acc)
    assert logits.shape == target.shape and logits.is_contiguous() and target.is_contiguous() and acc.numel() >= 2
    _chk(load().feddat_vqa_score_accumulate(_p(logits), _p(target), logits.shape[0], logits.shape[1], _p(acc), _stream()),
         "feddat_vqa_score_accumulate")


def lm_loss_fwd_bwd(logits, teacher, labels, row_weight, V, temp, kl_scale, dlogits_bf16, scalars, row_kl=None, grad_scale=1.0,
                    grad_scale_dev=None, nonfinite=None):
    """grad_scale_dev / nonfinite (device float / int32 tensors): the dynamic loss scale and its overflow flag (ABI 8)."""
    _dev(logits, teacher, labels, row_weight, dlogits_bf16, scalars, row_kl, grad_scale_dev, nonfinite)
    R = logits.shape[0]
    assert scalars.numel() >= 4 + 2 * R and labels.dtype == torch.int64
    if grad_scale_dev is not None or nonfinite is not None:
        _chk(load().feddat_lm_loss_fwd_bwd_dyn(_p(logits), _p(teacher), logits.stride(0), _p(labels), _p(row_weight), _p(row_kl), R, V,
                                               temp, kl_scale, float(grad_scale), _p(grad_scale_dev), _p(nonfinite), _p(dlogits_bf16),
                                               0 if dlogits_bf16 is None else dlogits_bf16.stride(0), _p(scalars), _stream()),
             "feddat_lm_loss_fwd_bwd_dyn")
        return
    _chk(load().feddat_lm_loss_fwd_bwd(_p(logits), _p(teacher), logits.stride(0), _p(labels), _p(row_weight), _p(row_kl), R, V, temp,
                                       kl_scale, float(grad_scale), _p(dlogits_bf16), 0 if dlogits_bf16 is None else dlogits_bf16.stride(0),
                                       _p(scalars), _stream()), "feddat_lm_loss_fwd_bwd")


def vilt_stage_inputs(input_ids, token_type_ids, attention_mask, target, pixel_mask, dst: dict, B, Lt, n_labels, Hi, Wi, P):
    """The step's small inputs -> the engine's static buffers (dst: input_ids, token_type_ids, attention_mask, target,
    patch_mask) in one launch.  All sources on the device, int64 / fp32, contiguous."""
    _dev(input_ids, token_type_ids, attention_mask, target, pixel_mask)
    _chk(load().feddat_vilt_stage_inputs(_p(input_ids), _p(token_type_ids), _p(attention_mask), _p(target), _p(pixel_mask),
                                         _p(dst["input_ids"]), _p(dst["token_type_ids"]), _p(dst["attention_mask"]),
                                         _p(dst["target"]), _p(dst["patch_mask"]), B, Lt, n_labels, Hi, Wi, P, _stream()),
         "feddat_vilt_stage_inputs")


def softmax_gather_rows(logits, rows, row_stride, V, ids, out):
    """out[r, j] = softmax(logits.flatten()[r * row_stride : r * row_stride + V])[ids[j]]; ids int64 (any 1-D stride)."""
    _dev(logits, ids, out)
    assert logits.dtype == torch.float32 and ids.dtype == torch.int64 and out.dtype == torch.float32 and out.is_contiguous()
    n = ids.shape[0]
    assert out.shape == (rows, n)
    _chk(load().feddat_softmax_gather_rows(_p(logits), row_stride, rows, V, _p(ids), ids.stride(0), n, _p(out), _stream()),
         "feddat_softmax_gather_rows")


def topk_rows(vals, k, *, minus=None, log_first=False, softmax=False):
    """-> (values [rows, k] fp32, indices [rows, k] int64), descending; ties: lower index first."""
    _dev(vals, minus)
    rows, n = vals.shape
    assert vals.dtype == torch.float32 and vals.stride(1) == 1 and (minus is None or (minus.is_contiguous() and minus.shape == vals.shape))
    ov = torch.empty(rows, k, dtype=torch.float32, device=vals.device)
    oi = torch.empty(rows, k, dtype=torch.int64, device=vals.device)
    _chk(load().feddat_topk_rows(_p(vals), vals.stride(0), _p(minus), rows, n, k, (1 if log_first else 0) | (2 if softmax else 0),
                                 _p(ov), _p(oi), _stream()), "feddat_topk_rows")
    return ov, oi


def dropout(x, drop, *, resid=None, out_f32=None, out_bf16=None):
    """out = mask(idx) x / (1 - p) (+ resid); x fp32 or bf16; drop = (p, key0, key1, step_counter)."""
    _dev(x, resid, out_f32, out_bf16)
    x32, x16 = (x, None) if x.dtype == torch.float32 else (None, x)
    _chk(load().feddat_dropout(_p(x32), _p(x16), _p(resid), _p(out_f32), _p(out_bf16), x.numel(), drop[0], drop[1], drop[2],
                               _p(drop[3]), _stream()), "feddat_dropout")


def dropout_keys(seed: int, pass_id: int, site: int):
    """(key0, key1) of one dropout site: splitmix64 of (seed, pass, site) split in two 32-bit halves (include/feddat_hip.h)."""
    M = (1 << 64) - 1
    z = (seed * 0x9E3779B97F4A7C15 + pass_id * 0xBF58476D1CE4E5B9 + site * 0x94D049BB133111EB + 0x2545F4914F6CDD1D) & M
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & M
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & M
    z ^= z >> 31
    return z & 0xFFFFFFFF, z >> 32


def axpby3(a, alpha, b=None, beta=0.0, c=None, gamma=0.0, *, out_f32=None, out_bf16=None):
    _dev(a, b, c, out_f32, out_bf16)
    _chk(load().feddat_axpby3(_p(a), alpha, _p(b), beta, _p(c), gamma, _p(out_f32), _p(out_bf16), a.numel(), _stream()),
         "feddat_axpby3")


def gather_rows(src, idx, dst_f32=None, dst_bf16=None):
    _dev(src, idx, dst_f32, dst_bf16)
    assert idx.dtype == torch.int32
    _chk(load().feddat_gather_rows(_p(src), _p(idx), _p(dst_f32), _p(dst_bf16), idx.numel(), src.shape[-1], _stream()),
         "feddat_gather_rows")


def segment_sum_rows(src, seg_offsets, dst, accumulate=False):
    _dev(src, seg_offsets, dst)
    assert seg_offsets.dtype == torch.int32
    _chk(load().feddat_segment_sum_rows(_p(src), _p(seg_offsets), _p(dst), seg_offsets.numel() - 1, src.shape[-1],
                                        int(accumulate), _stream()), "feddat_segment_sum_rows")


def adamw_flat(p, g, m, v, seg_off, seg_wd, state, base_lr, warmup, total, beta1=0.9, beta2=0.98, eps=1e-8):
    _dev(p, g, m, v)
    _chk(load().feddat_adamw_flat(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(seg_off), _p(seg_wd), seg_wd.numel(),
                                  _p(state), base_lr, warmup, total, beta1, beta2, eps, _stream()),
         "feddat_adamw_flat")


def step_tick(state, d_sched, d_adam):
    _chk(load().feddat_step_tick(_p(state), d_sched, d_adam, _stream()), "feddat_step_tick")


def text_embed(ids, tts, word, pos, typ, ln_g, ln_b, eps, mod0, h, B, Lt, S, H):
    _dev(ids, h)
    _chk(load().feddat_text_embed(_p(ids), _p(tts), _p(word), _p(pos), _p(typ), _p(ln_g), _p(ln_b), eps, _p(mod0),
                                  _p(h), B, Lt, S, H, _stream()), "feddat_text_embed")


def image_embed_assemble_masked(proj, cls, pos0, pos_grid, patch_mask, attention_mask, mod1, h, key_mask, B, Lt, gh, gw, g, H,
                                nrep=1):
    """image_embed_assemble + pos_embed_resize_masked + vilt_key_mask in one launch (patch_mask: int64 [B, gh, gw])."""
    _dev(proj, h, patch_mask)
    _chk(load().feddat_image_embed_assemble_masked(_p(proj), _p(cls), _p(pos0), _p(pos_grid), _p(patch_mask),
                                                   _p(attention_mask), _p(mod1), _p(h), _p(key_mask), B, Lt, gh, gw, g, H, nrep,
                                                   _stream()), "feddat_image_embed_assemble_masked")


def im2col_patches(pixels, patches, B, Cc, Hi, Wi, P):
    _dev(pixels, patches)
    _chk(load().feddat_im2col_patches(_p(pixels), _p(patches), B, Cc, Hi, Wi, P, _stream()), "feddat_im2col_patches")


def image_embed_assemble(proj, cls, pos0, pos_img, mod1, h, B, Lt, npatch, S, H, pos_batch_stride=0):
    _dev(proj, h)
    _chk(load().feddat_image_embed_assemble(_p(proj), _p(cls), _p(pos0), _p(pos_img), pos_batch_stride, _p(mod1), _p(h),
                                            B, Lt, npatch, S, H, _stream()), "feddat_image_embed_assemble")


def pos_embed_resize_masked(grid, pixel_mask, out, g, B, Hi, Wi, P, H):
    _dev(grid, pixel_mask, out)
    if pixel_mask.dtype != torch.int64:
        raise FeddatHipError("pixel_mask must be int64 (HF processor output)")
    _chk(load().feddat_pos_embed_resize_masked(_p(grid), _p(pixel_mask), _p(out), g, B, Hi, Wi, P, H, _stream()),
         "feddat_pos_embed_resize_masked")


def vilt_key_mask(attention_mask, pixel_mask, key_mask, B, Lt, Hi, Wi, P, nrep=1):
    _dev(key_mask)
    for m in (attention_mask, pixel_mask):
        if m is not None and (m.dtype != torch.int64 or not m.is_cuda):
            raise FeddatHipError("masks must be int64 device tensors (HF processor output)")
    _chk(load().feddat_vilt_key_mask(_p(attention_mask), _p(pixel_mask), _p(key_mask), B, Lt, Hi, Wi, P, nrep,
                                     _stream()), "feddat_vilt_key_mask")


def pos_embed_resize(grid, out, g, gh, gw, H):
    _dev(grid, out)
    _chk(load().feddat_pos_embed_resize(_p(grid), _p(out), g, gh, gw, H, _stream()), "feddat_pos_embed_resize")


def cvt_f32_bf16(x, out):
    _dev(x, out)
    _chk(load().feddat_cvt_f32_bf16(_p(x), _p(out), x.numel(), _stream()), "feddat_cvt_f32_bf16")


def transpose_f32_bf16(x, out, R, Cc):
    _dev(x, out)
    _chk(load().feddat_transpose_f32_bf16(_p(x), _p(out), R, Cc, _stream()), "feddat_transpose_f32_bf16")


def tanh_fwd(x):
    _dev(x)
    _chk(load().feddat_tanh_fwd(_p(x), x.numel(), _stream()), "feddat_tanh_fwd")


def tanh_bwd(y, dy, dx):
    _dev(y, dy, dx)
    _chk(load().feddat_tanh_bwd(_p(y), _p(dy), _p(dx), y.numel(), _stream()), "feddat_tanh_bwd")


def gelu_fwd(x, y):
    _dev(x, y)
    _chk(load().feddat_gelu_fwd(_p(x), _p(y), x.numel(), _stream()), "feddat_gelu_fwd")


def gelu_bwd(x, dy, dx):
    _dev(x, dy, dx)
    _chk(load().feddat_gelu_bwd(_p(x), _p(dy), _p(dx), x.numel(), _stream()), "feddat_gelu_bwd")


def scatter_cls_rows(rows, out_f32, out_bf16, B, S, H):
    _dev(rows)
    _chk(load().feddat_scatter_cls_rows(_p(rows), _p(out_f32), _p(out_bf16), B, S, H, _stream()),
         "feddat_scatter_cls_rows")


def fedavg_accumulate(acc, x, num, total, first):
    _dev(acc, x)
    _chk(load().feddat_fedavg_accumulate(_p(acc), _p(x), acc.numel(), float(num), float(total), int(first),
                                         _stream()), "feddat_fedavg_accumulate")


def probe_tr16(inp, out):
    _dev(inp, out)
    _chk(load().feddat_probe_tr16(_p(inp), _p(out), _stream()), "feddat_probe_tr16")
