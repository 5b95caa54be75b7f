"""ctypes binding of libdiffgraph_hip.so (the C ABI declared in include/diffgraph_hip.h).

There is no CPU fallback: if the library is missing or a call fails, a DiffGraphHipError is
raised.  Tensors cross the boundary as raw device pointers (tensor.data_ptr()) plus sizes;
the current torch HIP stream is passed as the stream argument.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdiffgraph_hip.so")

DG_D = 64
DG_MAX_TASKS = 8
DG_IN_PRESCALED, DG_OUT_NORM, DG_OUT_PRESCALE = 1, 2, 4
DG_BWD_SAVED_PRESCALED = 8
DG_ACC_REBUILD = 16
DG_BWD_NORM_FEED = 32
DG_ROWS_PARTIAL_ONLY, DG_ROWS_FULL_ONLY = 64, 128
DG_OUT_PLANES, DG_IN_PLANES = 0x200, 0x400
DG_ACC_SHARED = 0x800


class DiffGraphHipError(RuntimeError):
    pass


class LayerTask(C.Structure):
    _fields_ = [("graph", C.c_void_p), ("x", C.c_void_p), ("e_out", C.c_void_p), ("acc_in", C.c_void_p),
                ("acc_out", C.c_void_p), ("flags", C.c_int32), ("slope", C.c_float), ("partial_out", C.c_void_p),
                ("partial_from", C.c_int32), ("row_norm_out", C.c_void_p)]


class BwdTask(C.Structure):
    _fields_ = [("graph", C.c_void_p), ("w_in", C.c_void_p), ("e_saved", C.c_void_p), ("d_sum", C.c_void_p),
                ("w_out", C.c_void_p), ("flags", C.c_int32), ("slope", C.c_float), ("row_norm", C.c_void_p)]


class FinalizeTask(C.Structure):
    _fields_ = [("graph", C.c_void_p), ("partials", C.c_void_p), ("n_parts", C.c_int32), ("part_stride_rows", C.c_int64),
                ("row_begin", C.c_int32), ("n_rows", C.c_int32), ("x_prev", C.c_void_p), ("e_out", C.c_void_p),
                ("acc_in", C.c_void_p), ("acc_out", C.c_void_p), ("flags", C.c_int32), ("slope", C.c_float)]


class Denoiser(C.Structure):
    _fields_ = [("w_emb", C.c_void_p), ("b_emb", C.c_void_p), ("w_in", C.c_void_p), ("b_in", C.c_void_p),
                ("w_out", C.c_void_p), ("b_out", C.c_void_p), ("freqs_host", C.POINTER(C.c_float)),
                ("norm", C.c_int32)]


class PSampleTask(C.Structure):
    _fields_ = [("dn", C.POINTER(Denoiser)), ("time_bias", C.c_void_p), ("x_T", C.c_void_p), ("x_0", C.c_void_p),
                ("n", C.c_int64)]


class LossTask(C.Structure):
    _fields_ = [("dn", C.POINTER(Denoiser)), ("time_bias", C.c_void_p), ("target", C.c_void_p), ("x_start", C.c_void_p),
                ("noise", C.c_void_p), ("ts", C.c_void_p), ("n", C.c_int64), ("out", C.c_void_p), ("row_loss", C.c_void_p),
                ("g_row_loss", C.c_void_p), ("g_out", C.c_void_p), ("d_target_add", C.c_void_p), ("d_x_start", C.c_void_p),
                ("d_target", C.c_void_p), ("dout_tot", C.c_void_p), ("hid", C.c_void_p), ("dpre", C.c_void_p),
                ("xn", C.c_void_p), ("keep_mask", C.c_void_p), ("drop_seed", C.c_void_p), ("drop_p", C.c_float),
                ("wg_tab", C.c_void_p), ("wg_partials", C.c_void_p), ("rows", C.c_void_p), ("n_rows", C.c_void_p),
                ("rows_cap", C.c_int32)]


class WgradTask(C.Structure):
    _fields_ = [("dn", C.POINTER(Denoiser)), ("ts", C.c_void_p), ("dout_tot", C.c_void_p), ("hid", C.c_void_p),
                ("dpre", C.c_void_p), ("xn", C.c_void_p), ("n", C.c_int64), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("d_w_emb", C.c_void_p), ("d_b_emb", C.c_void_p), ("d_w_in", C.c_void_p),
                ("d_b_in", C.c_void_p), ("d_w_out", C.c_void_p), ("d_b_out", C.c_void_p)]


class WgradReduceTask(C.Structure):
    _fields_ = [("dn", C.POINTER(Denoiser)), ("partials", C.c_void_p), ("n_partials", C.c_int32), ("mk", C.c_void_p),
                ("d_w_emb", C.c_void_p), ("d_b_emb", C.c_void_p), ("d_w_in", C.c_void_p), ("d_b_in", C.c_void_p),
                ("d_w_out", C.c_void_p), ("d_b_out", C.c_void_p)]


# name -> (restype, argtypes); one entry per symbol declared in include/diffgraph_hip.h
SIGNATURES = {
    "dg_version": (C.c_int, []),
    "dg_last_error": (C.c_char_p, []),
    "dg_source_digest": (C.c_char_p, []),
    "dg_graph_create": (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                  C.POINTER(C.c_void_p)]),
    "dg_graph_create_ex": (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_int32, C.POINTER(C.c_void_p)]),
    "dg_graph_create_weighted": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_int32, C.POINTER(C.c_void_p)]),
    "dg_graph_num_rows": (C.c_int32, [C.c_void_p]),
    "dg_graph_destroy": (None, [C.c_void_p]),
    "dg_graph_num_nodes": (C.c_int32, [C.c_void_p]),
    "dg_graph_num_edges": (C.c_int64, [C.c_void_p]),
    "dg_graph_is_symmetric": (C.c_int, [C.c_void_p]),
    "dg_graph_export": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "dg_spmm_norm_act_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_int,
                                       C.c_void_p]),
    "dg_gcn_layer_fwd": (C.c_int, [C.POINTER(LayerTask), C.c_int32, C.c_int32, C.c_void_p]),
    "dg_gcn_partial_finalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "dg_gcn_partial_finalize_batched": (C.c_int, [C.POINTER(FinalizeTask), C.c_int32, C.c_int32, C.c_void_p]),
    "dg_allgather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "dg_gcn_layer_bwd": (C.c_int, [C.POINTER(BwdTask), C.c_int32, C.c_int32, C.c_void_p]),
    "dg_sum_tables": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "dg_graphset_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "dg_graphset_destroy": (None, [C.c_void_p]),
    "dg_graphset_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "dg_set_layer_first": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p]),
    "dg_set_layer_last": (C.c_int, [C.c_void_p] * 6 + [C.c_float, C.c_int32, C.c_void_p]),
    "dg_set_layer_bwd_top": (C.c_int, [C.c_void_p] * 6 + [C.c_float, C.c_int32, C.c_void_p]),
    "dg_set_layer_bwd_mid": (C.c_int, [C.c_void_p] * 6 + [C.c_float, C.c_int32, C.c_void_p]),
    "dg_set_layer_bwd_last": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "dg_set_layer_bwd_last_adam": (C.c_int, [C.c_void_p] * 8 + [C.c_float] * 5 + [C.c_int32, C.c_void_p]),
    "dg_q_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                              C.c_int32, C.c_void_p, C.c_void_p]),
    "dg_time_bias_table": (C.c_int, [C.POINTER(Denoiser), C.c_int32, C.c_void_p, C.c_void_p]),
    "dg_time_bias_tables": (C.c_int, [C.POINTER(C.POINTER(Denoiser)), C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p]),
    "dg_denoise_fwd": (C.c_int, [C.POINTER(Denoiser), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                 C.c_int32, C.c_void_p, C.c_void_p]),
    "dg_denoise_fwd_train": (C.c_int, [C.POINTER(Denoiser), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "dg_p_sample_loop": (C.c_int, [C.POINTER(Denoiser), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                   C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "dg_p_sample_loop_batched": (C.c_int, [C.POINTER(PSampleTask), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_int32, C.c_void_p]),
    "dg_diffusion_loss_fwd": (C.c_int, [C.POINTER(Denoiser), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dg_diffusion_loss_bwd": (C.c_int, [C.POINTER(Denoiser), C.c_void_p, C.c_int32] + [C.c_void_p] * 9 +
                              [C.c_int64, C.c_int32] + [C.c_void_p] * 7),
    "dg_diffusion_loss_fwd_batched": (C.c_int, [C.POINTER(LossTask), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int32, C.c_void_p]),
    "dg_diffusion_loss_bwd_batched": (C.c_int, [C.POINTER(LossTask), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int32, C.c_void_p]),
    "dg_diffusion_draw": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dg_dropout_keep_mask": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "dg_denoiser_wgrad_workspace": (C.c_size_t, [C.c_int32]),
    "dg_denoiser_wgrad": (C.c_int, [C.POINTER(Denoiser), C.c_int32] + [C.c_void_p] * 5 + [C.c_int64, C.c_int32, C.c_void_p,
                                    C.c_size_t] + [C.c_void_p] * 7),
    "dg_wgrad_partial_floats": (C.c_int64, []),
    "dg_denoiser_wgrad_tab": (C.c_int, [C.POINTER(C.POINTER(Denoiser)), C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p]),
    "dg_denoiser_wgrad_reduce": (C.c_int, [C.POINTER(WgradReduceTask), C.c_int32, C.c_void_p]),
    "dg_denoiser_wgrad_static_max_steps": (C.c_int32, []),
    "dg_denoiser_wgrad_static": (C.c_int, [C.POINTER(Denoiser), C.c_int32] + [C.c_void_p] * 5 + [C.c_int64, C.c_int32] +
                                 [C.c_void_p] * 7),
    "dg_denoiser_wgrad_batched": (C.c_int, [C.POINTER(WgradTask), C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dg_eval_topk_workspace": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "dg_eval_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                               C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dg_eval_metrics": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 8),
    "dg_neg_sampling": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p,
                                  C.c_void_p]),
    "dg_spgemm_bitmap_words": (C.c_int32, [C.c_int32]),
    "dg_spgemm_pattern_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "dg_spgemm_pattern_fill": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dg_adam_step": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "dg_adam_step_prepared": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "dg_active_rows": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 3 + [C.c_void_p] * 5),
    "dg_rec_head_fwd": (C.c_int, [C.c_void_p] * 8 + [C.c_int32] * 3 + [C.c_double, C.c_double, C.c_int32] + [C.c_void_p] * 4),
    "dg_rec_head_reduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "dg_linear_fwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 3 + [C.c_void_p, C.c_void_p]),
    "dg_linear_wgrad_workspace": (C.c_size_t, [C.c_int32, C.c_int32]),
    "dg_linear_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "dg_nc_logits": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "dg_nc_nll_fwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dg_nc_nll_bwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int32, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 4),
    "dg_nc_head_fwd": (C.c_int, [C.c_void_p] * 8 + [C.c_int32, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 5),
    "dg_nc_head_bwd": (C.c_int, [C.c_void_p] * 8 + [C.c_int32, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 5),
    "dg_adam_prepare": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_void_p]),
    "dg_adam_multi": (C.c_int, [C.POINTER(C.c_void_p)] * 4 + [C.POINTER(C.c_int64), C.c_int32, C.c_void_p] + [C.c_float] * 5
                      + [C.c_void_p]),
    "dg_clear": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "dg_rec_head_bwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 3 + [C.c_double, C.c_double, C.c_int32] + [C.c_void_p] * 10),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Loads the shared library (once) and types every exported entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DiffGraphHipError(
            f"{LIB_PATH} not found: build it with `python -m diffgraph_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # a library built from other kernel sources than the ones beside it is stale: refuse it instead of measuring or
    # testing something else than the tree holds (sources absent, e.g. an installed copy: nothing to compare with)
    from . import build as _build
    if os.path.isdir(_build.CSRC) and not os.environ.get("DG_ALLOW_STALE_LIB"):
        built, here = lib.dg_source_digest().decode(), _build.source_digest()
        if built != here:
            raise DiffGraphHipError(f"{LIB_PATH} was built from other kernel sources (digest {built}, the sources here have "
                                    f"{here}): rebuild it with `python -m diffgraph_amd.build`")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().dg_last_error()
        raise DiffGraphHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def device_guard(device=None):
    """Context that makes `device` the current GPU (no-op for None)."""
    import contextlib
    import torch
    if device is None:
        return contextlib.nullcontext()
    device = torch.device(device)
    if device.type != "cuda":
        raise DiffGraphHipError(f"graphs live on a GPU, got device {device}")
    return torch.cuda.device(device)


def stream_ptr(device=None) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream
