"""ctypes binding of libgeossl_hip.so (C ABI in include/geossl_hip.h).

There is no CPU fallback: if the library is missing this module raises, and every op refuses
non-CUDA tensors.  PyTorch is used for device memory and streams only.
"""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libgeossl_hip.so")

MAX_L = 12
TN_MAX = 32
EPI_BIAS, EPI_SSP, EPI_RESIDUAL, EPI_MUL_DSSP, CHAIN_SAME_INPUT, CHAIN_NEW_INPUT, CHAIN_ADD_PREV = 1, 2, 4, 8, 16, 32, 64
EPI_SILU, EPI_MUL_DSILU = 128, 256

vp = C.c_void_p
i64 = C.c_int64
i32 = C.c_int
f32 = C.c_float
f64 = C.c_double
u64 = C.c_uint64


class FilterWeights(C.Structure):
    _fields_ = [("w1", vp * MAX_L), ("b1", vp * MAX_L), ("w2", vp * MAX_L), ("b2", vp * MAX_L)]


PREPARE_MAX = 64


class PrepareBatch(C.Structure):
    _fields_ = [("W", vp * PREPARE_MAX), ("image", vp * PREPARE_MAX), ("ldw", i32 * PREPARE_MAX), ("tb", i32 * PREPARE_MAX)]


CHAIN_MAX = 5


class ChainStage(C.Structure):
    _fields_ = [("image", vp), ("bias", vp), ("res", vp), ("tprev", vp), ("out", vp), ("ld", i32), ("flags", i32),
                ("xin", vp), ("ldxin", i32), ("pad_", i32), ("out_act", vp)]


class Chain(C.Structure):
    _fields_ = [("nstage", i32), ("st", ChainStage * CHAIN_MAX)]


LOOP_MAX_OPS = 14


class LoopOp(C.Structure):
    _fields_ = [("kind", i32), ("swap", i32), ("X", vp), ("Wf", vp), ("out", vp), ("chain", Chain)]


class FilterGradIn(C.Structure):
    _fields_ = [("x", vp * MAX_L), ("dagg", vp * MAX_L)]


class FilterGradOut(C.Structure):
    _fields_ = [("dw1", vp * MAX_L), ("db1", vp * MAX_L), ("dw2", vp * MAX_L), ("db2", vp * MAX_L)]


class TnBatch(C.Structure):
    _fields_ = [("A", vp * TN_MAX), ("B", vp * TN_MAX), ("dW", vp * TN_MAX), ("db", vp * TN_MAX)]


class NcsnWeights(C.Structure):
    _fields_ = [(k, vp) for k in ("in_w1", "in_b1", "in_w2", "in_b2", "o1_w", "o1_b", "o2_w", "o2_b", "o3_w", "o3_b",
                                  "sigmas")]


class NcsnGrads(C.Structure):
    _fields_ = [(k, vp) for k in ("in_w1", "in_b1", "in_w2", "in_b2", "o1_w", "o1_b", "o2_w", "o2_b", "o3_w", "o3_b")]


class NcsnSaved(C.Structure):
    _fields_ = [(k, vp) for k in ("a1", "a2", "pd", "emb", "gscale")]


class NcsnHeadFwd(C.Structure):
    _fields_ = [("h", vp), ("distance", vp), ("noise_level", vp), ("distance_noise", vp), ("w", NcsnWeights),
                ("saved", NcsnSaved), ("anneal_power", f32), ("pad_", f32), ("loss_e", vp), ("workspace", vp)]


class NcsnHeadBwd(C.Structure):
    _fields_ = [("h", vp), ("w", NcsnWeights), ("saved", NcsnSaved), ("out_scale", f32), ("pad_", f32), ("dfeat", vp),
                ("demb", vp), ("grow", vp), ("grads", NcsnGrads), ("workspace", vp), ("dh", vp)]


COPY_MAX = 8


class CopyBatch(C.Structure):
    _fields_ = [("dst", vp * COPY_MAX), ("src", vp * COPY_MAX), ("bytes", i64 * COPY_MAX)]


class Gather(C.Structure):
    _fields_ = [(k, vp) for k in ("x_src", "pos_src", "src_off", "mol_ptr", "x_dst", "pos_dst", "batch_dst", "se_ptr", "sei0",
                                  "sei1", "pair_ptr2", "pair_i", "pair_j", "inc_ptr", "inc_idx", "e0_src", "e1_src",
                                  "e_src_off", "e_ptr", "e0_dst", "e1_dst", "zero")] + \
               [("zero_count", i64), ("x_cols", i32), ("option", i32)]


P = C.POINTER
# name -> (restype, argtypes); mirrors include/geossl_hip.h one to one
PROTOTYPES = {
    "geossl_abi_version": (i32, []),
    "geossl_layout_build": (i32, [vp, i64, i64, vp, vp, vp, vp]),
    "geossl_pair_index_fill": (i32, [vp, vp, i64, vp, vp, vp]),
    "geossl_radius_graph_count": (i32, [vp, vp, i64, i32, f32, i32, vp, vp]),
    "geossl_radius_graph_fill": (i32, [vp, vp, i64, i32, f32, i32, vp, vp, vp, vp, vp]),
    "geossl_pair_geometry": (i32, [vp, vp, vp, i64, i32, f32, i32, f32, vp, vp, vp, vp]),
    "geossl_rbf_fwd": (i32, [vp, i64, vp, i32, f32, vp, vp]),
    "geossl_ssp_fwd": (i32, [vp, i64, vp, vp]),
    "geossl_ssp_bwd": (i32, [vp, vp, i64, vp, vp]),
    "geossl_cfconv_filter_fwd": (i32, [vp, vp, i64, P(FilterWeights), i32, i32, i32, vp, f32, vp, vp, vp]),
    "geossl_cfconv_filter_bwd_workspace_floats": (i64, [i64, i32, i32, i32]),
    "geossl_cfconv_filter_bwd": (i32, [vp, vp, vp, vp, vp, i64, i64, P(FilterWeights), P(FilterGradIn), i32, i32, i32, vp,
                                       f32, vp, P(FilterGradOut), vp, i32, vp]),
    "geossl_cfconv_filter_dpos": (i32, [vp, vp, vp, vp, vp, i64, P(FilterWeights), P(FilterGradIn), i32, i32, i32, vp, f32,
                                        f32, vp, vp, vp, vp]),
    "geossl_pair_position_grad": (i32, [vp, vp, vp, vp, vp, i64, i64, i32, vp, vp]),
    "geossl_cfconv_aggregate": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp, vp]),
    "geossl_aggregate_parts": (i32, [i32]),
    "geossl_cfconv_aggregate_work": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp, vp]),
    "geossl_pair_product": (i32, [vp, vp, vp, vp, vp, i64, i32, i32, vp, vp]),
    "geossl_linear": (i32, [vp, i32, vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, vp]),
    "geossl_tn_plan": (None, [i64, i32, P(i32), P(i32)]),
    "geossl_tn_workspace_floats": (i64, [i64, i32, i32, i32]),
    "geossl_row_normalize_fwd": (i32, [vp, i64, i32, f32, vp, vp, vp]),
    "geossl_row_normalize_bwd": (i32, [vp, vp, vp, i64, i32, f32, vp, vp]),
    "geossl_atom_tuples": (i32, [vp, vp, i64, i32, vp, vp, vp]),
    "geossl_linear_image_words": (i64, [i32, i32]),
    "geossl_linear_prepare": (i32, [P(PrepareBatch), i32, i32, i32, i32, vp]),
    "geossl_linear_prepared": (i32, [vp, i32, vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, vp]),
    "geossl_chain_image_words": (i64, [i32]),
    "geossl_chain_prepare": (i32, [P(PrepareBatch), i32, i32, i32, vp]),
    "geossl_linear_chain": (i32, [vp, i32, P(Chain), i64, i32, vp]),
    "geossl_linear_wgrad": (i32, [P(TnBatch), i32, i64, i32, i32, i32, i32, i32, vp, i32, vp]),
    "geossl_embedding_fwd": (i32, [vp, i64, vp, i32, i64, i32, vp, vp, vp]),
    "geossl_embedding_bwd_workspace_floats": (i64, [i32, i32]),
    "geossl_embedding_bwd": (i32, [vp, i64, vp, i32, i64, i32, vp, vp, i32, vp]),
    "geossl_segment_reduce_fwd": (i32, [vp, vp, i64, i32, i32, vp, vp]),
    "geossl_segment_reduce_bwd": (i32, [vp, vp, i64, i32, i32, vp, i32, vp]),
    "geossl_axpy": (i32, [vp, vp, f32, i64, vp, vp]),
    "geossl_pair_distance": (i32, [vp, vp, vp, i64, vp, vp]),
    "geossl_super_edge_ptr": (i32, [vp, vp, vp, i64, i64, vp, vp, vp]),
    "geossl_incidence_count": (i32, [vp, vp, vp, vp, i64, i32, vp, vp]),
    "geossl_incidence_fill": (i32, [vp, vp, vp, vp, i64, i32, vp, vp, vp]),
    "geossl_ddm_loss_fwd_workspace_floats": (i64, [i32]),
    "geossl_ddm_loss_fwd": (i32, [vp, vp, vp, vp, i64, vp, vp, vp, P(NcsnWeights), i32, f32, vp, P(NcsnSaved), vp, vp]),
    "geossl_schnet_layer_loop": (i32, [vp, i32, vp, i32, vp, vp, vp, i32, i32, i64, i32, i32, vp]),
    "geossl_schnet_layer_loop_ragged": (i32, [vp, i32, vp, vp, vp, i64, i32, i64, i32, vp]),
    "geossl_copy2": (i32, [vp, vp, i64, vp, vp, i64, vp]),
    "geossl_ddm_noise": (i32, [vp, f32, f32, i64, i64, i64, i32, i32, vp, vp, vp, vp, vp, vp]),
    "geossl_ddm_noise_seeded": (i32, [u64, f32, f32, i64, i64, i64, i32, i32, vp, vp, vp, vp, vp, vp]),
    "geossl_ddm_views": (i32, [vp, vp, vp, vp, i64, i64, vp, vp, vp, vp, i64, vp, vp]),
    "geossl_loss_reduce_partials": (i32, [vp, vp, f32, vp, i32, vp]),
    "geossl_ddm_loss_fwd2": (i32, [vp, vp, vp, vp, i64, i32, vp]),
    "geossl_loss_reduce_partials2": (i32, [vp, vp, vp, f32, f32, vp, vp]),
    "geossl_ddm_loss_bwd_fused2": (i32, [vp, vp, vp, i64, i64, i32, vp, vp, vp, vp, i32, vp]),
    "geossl_loss_reduce_workspace_floats": (i64, [i64]),
    "geossl_loss_reduce": (i32, [vp, i64, vp, f32, vp, vp, i32, vp]),
    "geossl_ddm_loss_bwd_rows": (i32, [P(NcsnWeights), P(NcsnSaved), i64, i32, vp, f32, vp, vp, vp, vp, vp, vp]),
    "geossl_ddm_loss_bwd_workspace_floats": (i64, [i64, i32]),
    "geossl_ddm_loss_bwd_weights": (i32, [vp, vp, vp, i64, i32, P(NcsnWeights), P(NcsnSaved), vp, vp, vp, P(NcsnGrads),
                                          vp, i32, vp]),
    "geossl_ddm_loss_bwd_fused_workspace_floats": (i64, [i64, i32]),
    "geossl_ddm_loss_bwd_fused": (i32, [vp, vp, vp, i64, i64, i32, P(NcsnWeights), P(NcsnSaved), vp, f32, vp, vp, vp, vp,
                                        P(NcsnGrads), vp, i32, vp]),
    "geossl_incidence_gather": (i32, [vp, vp, vp, i64, i32, vp, i32, vp]),
    "geossl_painn_edge_geom": (i32, [vp, vp, vp, i64, f32, vp, vp, i32, vp, vp, vp, vp]),
    "geossl_silu_fwd": (i32, [vp, i64, vp, vp]),
    "geossl_silu_bwd": (i32, [vp, vp, i64, vp, vp]),
    "geossl_painn_interaction_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp, vp]),
    "geossl_painn_interaction_bwd_workspace_floats": (i64, [i64, i32, i32]),
    "geossl_painn_interaction_fwd_mol": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i64, i32, i32, vp,
                                               vp, vp]),
    "geossl_painn_interaction_fwd_mma": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i64, i32, i32,
                                               vp, vp, vp]),
    "geossl_painn_interaction_bwd_mol_workspace_floats": (i64, [i64, i64, i32, i32]),
    "geossl_painn_interaction_bwd_mol": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i64, i32, i32,
                                               vp, vp, vp, vp, vp, i32, vp]),
    "geossl_painn_interaction_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp, vp, vp,
                                           vp, i32, vp]),
    "geossl_painn_mix_pre_fwd": (i32, [vp, vp, i64, i32, f32, vp, vp, vp]),
    "geossl_painn_mix_post_fwd": (i32, [vp, vp, vp, vp, vp, i64, i32, vp, vp, vp]),
    "geossl_painn_mix_post_bwd": (i32, [vp, vp, vp, vp, vp, i64, i32, vp, vp, vp]),
    "geossl_painn_mix_pre_bwd": (i32, [vp, vp, vp, vp, i64, i32, vp, vp, vp]),
    "geossl_add": (i32, [vp, vp, i64, vp, vp]),
    "geossl_painn_edge_grads": (i32, [vp] * 11 + [i64, i32, i32, vp, vp, vp, i32, vp]),
    "geossl_painn_edge_geom_bwd": (i32, [vp, vp, vp, i64, f32, vp, vp, i32, vp, vp, vp, vp, vp]),
    "geossl_painn_position_grad": (i32, [vp, vp, vp, vp, vp, i64, vp, vp]),
    "geossl_adam_step": (i32, [vp, vp, vp, vp, i64, f64, f64, f64, f64, f64, i64, f32, vp]),
    # capacity launches: the namesake's arguments + device-side row count(s) before the stream
    "geossl_copy_n": (i32, [P(CopyBatch), i32, vp]),
    "geossl_ddm_views_dyn": (i32, [vp, vp, vp, vp, i64, i64, vp, vp, vp, vp, i64, vp, vp, vp, vp]),
    "geossl_embedding_fwd_dyn": (i32, [vp, i64, vp, i32, i64, i32, vp, vp, vp, vp]),
    "geossl_embedding_bwd_dyn": (i32, [vp, i64, vp, i32, i64, i32, vp, vp, i32, vp, vp]),
    "geossl_cfconv_filter_fwd_dyn": (i32, [vp, vp, i64, P(FilterWeights), i32, i32, i32, vp, f32, vp, vp, vp, vp]),
    "geossl_cfconv_filter_bwd_dyn": (i32, [vp, vp, vp, vp, vp, i64, i64, P(FilterWeights), P(FilterGradIn), i32, i32, i32,
                                           vp, f32, vp, P(FilterGradOut), vp, i32, vp, vp, vp]),
    "geossl_cfconv_aggregate_work_dyn": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp]),
    "geossl_cfconv_aggregate_targets_dyn": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp]),
    "geossl_linear_chain_dyn": (i32, [vp, i32, P(Chain), i64, i32, vp, vp]),
    "geossl_linear_wgrad_dyn": (i32, [P(TnBatch), i32, i64, i32, i32, i32, i32, i32, vp, i32, vp, vp]),
    "geossl_ddm_loss_fwd2_dyn": (i32, [vp, vp, vp, vp, i64, i32, vp, vp, vp]),
    "geossl_ddm_loss_bwd_fused2_dyn": (i32, [vp, vp, vp, i64, i64, i32, vp, vp, vp, vp, i32, vp, vp, vp]),
    # PaiNN on a capacity bucket
    "geossl_painn_group_capacity": (i64, [i64, i64]),
    "geossl_painn_edge_layout": (i32, [vp, vp, i64, vp, i64, i64, i64] + [vp] * 12),
    "geossl_painn_edge_geom_dyn": (i32, [vp, vp, vp, i64, f32, vp, vp, i32, vp, vp, vp, vp, vp]),
    "geossl_painn_interaction_fwd_mma_dyn": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i64, i32,
                                                   i32, vp, vp, vp, vp]),
    "geossl_tape_unary": (i32, [i32, vp, i64, f32, f32, vp, vp]),
    "geossl_tape_binary": (i32, [i32, vp, i32, vp, i32, i64, i32, f32, vp, vp]),
    "geossl_tape_colsum_workspace_floats": (i64, [i64, i32]),
    "geossl_tape_reduce": (i32, [i32, vp, i64, i32, vp, vp, vp]),
    "geossl_tape_gather_rows": (i32, [vp, vp, i32, i64, i32, vp, vp]),
    "geossl_tape_scatter_rows": (i32, [vp, vp, i32, vp, i64, i32, vp, vp]),
    "geossl_tape_copy2d": (i32, [vp, i64, vp, i64, i64, i32, vp]),
    "geossl_tape_fill": (i32, [vp, i64, f32, vp]),
    "geossl_painn_stage_cap": (i32, [i32, i32, i32]),
    "geossl_painn_interaction_fwd_atoms": (i32, [vp] * 12 + [i64, vp, i32, i32, vp, vp, vp]),
    "geossl_painn_interaction_bwd_atoms": (i32, [vp] * 13 + [i64, vp, i32, i32, vp, vp, vp, vp, vp, i32, vp]),
    "geossl_painn_interaction_bwd_mol_skip": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i64, i32,
                                                    i32, vp, vp, vp, vp, vp, i32, vp]),
    "geossl_painn_mix_pre_fwd_dyn": (i32, [vp, vp, i64, i32, f32, vp, vp, vp, vp]),
    "geossl_painn_mix_post_fwd_dyn": (i32, [vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp]),
    "geossl_painn_mix_post_bwd_dyn": (i32, [vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp]),
    "geossl_painn_mix_pre_bwd_dyn": (i32, [vp, vp, vp, vp, i64, i32, vp, vp, vp, vp]),
    "geossl_gather_molecules": (i32, [P(Gather), i64, vp]),
}

_lib = None


class GeosslHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and bind every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("GEOSSL_HIP_LIB", LIB_PATH)  # override: A/B timing of two builds on one machine
    if not os.path.exists(path):
        raise GeosslHipError(
            "libgeossl_hip.so is not built (%s). Run `python -m geossl_amd.build` (needs hipcc); "
            "there is no CPU fallback." % path)
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.geossl_abi_version() != 1:
        raise GeosslHipError("ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise GeosslHipError("%s failed with hipError %d" % (what, rc))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream():
    """The current HIP stream of the current device as a raw handle.  (torch.cuda.current_stream().cuda_stream builds a
    Stream object per call - 8 us, a third of the host time of a launch-bound path like the second-order tape.)"""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise GeosslHipError(
                "geossl_amd runs on MI355X only: got a %s tensor. The HIP path has no CPU fallback." % t.device.type)



# ---- direct gradient accumulation (opt-in) -------------------------------------------------------------------------
# The custom autograd nodes can accumulate parameter gradients straight into dense ``p.grad`` buffers and return None
# for the parameter inputs (no temporaries, no AccumulateGrad adds).  That bypasses autograd's contract -
# ``torch.autograd.grad(loss, params)``, tensor hooks and bucketed reducers never see those gradients - so it only
# happens inside this context, which DDMTrainer (owner of the flat gradient buffer) opens around ``loss.backward()``.
# A process-wide flag, not a thread-local: autograd runs the backward of CUDA nodes on its own device thread.
_DIRECT = {"depth": 0}


class direct_grads:
    def __enter__(self):
        _DIRECT["depth"] += 1
        return self

    def __exit__(self, *exc):
        _DIRECT["depth"] -= 1
        return False


def direct_grads_enabled(params):
    """True when the caller opted in AND every parameter owns a dense fp32 CUDA ``.grad`` to accumulate into."""
    return _DIRECT["depth"] > 0 and all(
        p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32 and p.grad.is_cuda
        for p in params)


class StatusWord:
    """Device-side error word of a module (e.g. "an atom type was outside the embedding table", which the reference
    reports as an IndexError from ``Embedding``): kernels set it, nothing resets it.  Checked WITHOUT draining the
    stream: ``arm()`` queues an async copy into pinned host memory plus an event, ``poll()`` reads the copy once that
    event has completed - the error surfaces one or two calls late instead of costing a sync per step.
    ``check()`` synchronises (tests, GEOSSL_DEBUG, end of an epoch)."""

    def __init__(self, device, message):
        self.word = torch.zeros(1, dtype=torch.int32, device=device)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.event = None
        self.message = message

    def __deepcopy__(self, memo):  # (lives in module.__dict__: a copied / pickled module makes its own on first use)
        return None

    def __reduce__(self):
        return (type(None), ())

    def poll(self):
        if self.event is None or torch.cuda.is_current_stream_capturing():  # no event queries inside a capture
            return
        if self.event.query():
            self.event = None
            if int(self.host[0]):
                raise IndexError(self.message)

    def arm(self, every=1):
        """Queue the copy + event (if none is pending).  `every` > 1: only on every that-many-th call - the copy and
        its event cost a launch and a queue marker (~10 us), which a step of a few hundred us notices."""
        self.calls = getattr(self, "calls", 0) + 1
        if every > 1 and self.calls % every != 1:
            return
        if self.event is None and not torch.cuda.is_current_stream_capturing():
            self.host.copy_(self.word, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()

    def check(self):
        if int(self.word.item()):
            raise IndexError(self.message)


def module_status(module, device, message):
    """The StatusWord of an nn.Module, created on first use (a plain attribute: not a buffer, not in state_dict)."""
    st = module.__dict__.get("_geossl_status")
    if st is None or st.word.device != device:
        st = StatusWord(device, message)
        module.__dict__["_geossl_status"] = st
    return st


# Optional per-entry-point HIP-event timing (bench.py): {entry point name: [(start_event, end_event), ...]}.
# Events are recorded on the stream the kernels are launched on (torch's current stream).
TIMERS = None
CALLS = None  # bench.py: set to 0 to count the C-ABI calls (an int, else None)


def call(name, *args):
    global CALLS
    lib = load()
    if CALLS is not None:
        CALLS += 1
    timers = TIMERS
    key = name[:-4] if name.endswith("_dyn") else name  # (a `_dyn` entry point is timed under its namesake)
    if timers is not None and key in timers:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        timers[key].append((e0, e1))
    else:
        rc = getattr(lib, name)(*args)
    check(rc, name)


def fill_ptrs(arr, tensors):
    for i, t in enumerate(tensors):
        arr[i] = ptr(t)
