"""ctypes binding of libsehip.so -- the C-ABI boundary (include/sehip.h).

There is no fallback: if the shared library is missing or a call fails, a SehipError is raised.
All device pointers are borrowed from torch tensors that the caller keeps alive; every call is
asynchronous on the caller's current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEHIP_LIB") or os.path.join(_HERE, "libsehip.so")   # SEHIP_LIB: A/B of two builds in tools/


class SehipError(RuntimeError):
    pass


_lib = None

P = C.c_void_p
I = C.c_int
L = C.c_long
F = C.c_float
U = C.c_uint

# name -> argtypes (all return int status unless listed in _RESTYPE)
_PROTOS = {
    "sehip_version": [],
    "sehip_check_device": [I],
    "sehip_set_deterministic": [I],
    "sehip_get_deterministic": [],
    "sehip_event_create": [],
    "sehip_stream_create": [C.c_int],
    "sehip_stream_destroy": [P],
    "sehip_event_destroy": [P],
    "sehip_stream_depend": [P, P, P],
    "sehip_event_record": [P, P],
    "sehip_stream_wait_event": [P, P],
    "sehip_stft_frames": [I, I, I],
    "sehip_stft_custom_frames": [I, I, I, I],
    "sehip_stft_custom_fwd": [P, I, I, I, I, I, I, P, P],
    "sehip_istft_custom_fwd": [P, I, I, I, I, I, I, I, P, P, P],
    "sehip_comm_unique_id": [P],
    "sehip_comm_init": [P, I, I, P],
    "sehip_allreduce_f32": [P, P, L, P],
    "sehip_allreduce_i32_max": [P, P, L, P],
    "sehip_comm_info": [P, P, P, P],
    "sehip_comm_destroy": [P],
    "sehip_wav_row_stats": [P, P, I, P, P],
    "sehip_wav_collate": [P, P, P, P, P, P, I, F, I, I, P, P],
    "sehip_sisnr_fwd": [P, P, I, I, P, P, P],
    "sehip_sisdr_metric": [P, P, I, I, P, P, P],
    "sehip_sisnr_bwd": [P, P, P, P, I, I, P, P],
    "sehip_sisnr_pit_fwd": [P, P, I, I, I, I, P, P, P, P, P],
    "sehip_sisnr_pit_bwd": [P, P, P, P, P, I, I, I, I, P, P],
    "sehip_psa_loss_fwd": [P, P, P, L, P, P, P],
    "sehip_psa_loss_bwd": [P, P, P, L, P, P, P],
    "sehip_pointwise_loss_fwd": [P, P, L, I, P, P, P],
    "sehip_pointwise_loss_bwd": [P, P, L, I, P, P, P],
    "sehip_grad_sumsq": [P, L, P, P],
    "sehip_opt_step": [P, P, P, P, L, P, F, F, F, F, F, I, P, F, I, F, P],
    "sehip_opt_step_g": [P, P, P, P, L, P, F, F, F, F, F, I, P, F, I, F, P, P],
    "sehip_opt_step_m": [P, P, P, P, L, P, F, F, F, F, F, I, P, F, I, F, P, P, I, P, P, P, P],
    "sehip_opt_begin_g": [P, I, P, P, I, P, P],
    "sehip_counter_add": [P, I, P],
    "sehip_zero_regions": [P, L, P, L, P, L, P, L, P],
    "sehip_gemm_takes_gln_stats": [P],
    "sehip_init": [],
    "sehip_grad_metric": [P, P, I, L, P, P, P, P],
    "sehip_opt_begin": [P, I, P, P, I, P],
    "sehip_grad_sumsq_acc": [P, L, P, P],
    "sehip_grad_metric_acc": [P, P, I, L, P, P, P, P],
    "sehip_cbn_scratch_floats": [L, I],
    "sehip_stft_fwd": [P, P, I, I, I, I, I, P, P, P],
    "sehip_istft_fwd": [P, P, P, P, I, I, I, I, I, I, I, P, P, P],
    "sehip_istft_bwd": [P, P, P, P, P, P, I, I, I, I, I, I, I, P, P],
    "sehip_gemm": [P, P],
    "sehip_gemm_pair": [P, P, P],
    "sehip_wgrad": [P, P],
    "sehip_wgrad_pair": [P, P, P],
    "sehip_bnr_rows": [P, P],
    "sehip_cbn_bwd_finalize_n": [P, I, P, P, P, P, L, I, P, P, P, P, P, P, P, P],
    "sehip_wgrad_group_bytes": [I],
    "sehip_wgrad_group_prepare": [P, I, P, P],
    "sehip_wgrad_group": [P, I, I, P],
    "sehip_wgrad_dense_group_bytes": [I],
    "sehip_wgrad_dense_group_prepare": [P, I, P, L, P],
    "sehip_wgrad_dense_group": [P, I, P, P, P],
    "sehip_gemm_desc_size": [],
    "sehip_conv_small_takes": [P, P],
    "sehip_pack_bf16": [P, P, L, P, P],
    "sehip_pack_f32": [P, P, L, P, P],
    "sehip_pack_head": [P, P, L, P, P, L, P, P, L, P],
    "sehip_unpack_grad": [P, P, L, P, P],
    "sehip_unpack_grad_sums": [P, P, L, P, P, I, P, P, P, P, P],
    "sehip_unpack_grad_sums_perm": [P, P, P, L, P, P, I, P, P, P, P, P],
    "sehip_unpack_grad1": [P, P, L, P, P],
    "sehip_unpack_grad1_sums": [P, P, L, P, P, I, P, P, P, P, P],
    "sehip_unpack_grad_list_sums": [P, P, P, L, P, P, I, P, P, P],
    "sehip_unpack_grad_list": [P, P, P, L, P, P],
    "sehip_pack_bf16_runs": [P, P, P, L, P, P],
    "sehip_pack_bf16_runs_to": [P, P, P, P, L, P, P],
    "sehip_cbn_stats": [P, L, I, P, P],
    "sehip_cbn_finalize": [P, P, P, P, P, P, P, P, P, P, P, P, L, I, F, F, I, P, P],
    "sehip_cbn_finalize_n": [P, I, P, P, P, P, P, P, P, P, P, P, P, L, I, F, F, I, P, P],
    "sehip_cbn_apply": [P, P, P, L, I, P, P],
    "sehip_cbn_finalize_apply_n": [P, P, I, P, P, P, P, P, P, P, P, P, P, P, L, I, F, F, I, P, P, P, P],
    "sehip_cbn_bwd_reduce": [P, P, P, P, P, L, I, I, I, I, P, P],
    "sehip_cbn_bwd_finalize": [P, P, P, P, P, L, I, P, P, P, P, P, P, P, P],
    "sehip_cbn_bwd_apply": [P, P, P, P, P, P, L, I, I, I, I, P, P],
    "sehip_cbn_bwd_fused": [P, P, P, P, P, P, P, P, L, I, I, I, I, P, P, I, P, P, P, P, P, P, P, P],
    "sehip_cbn_bwd_reduce_fin": [P, P, P, P, P, P, P, P, L, I, I, I, I, P, I, P, P, P, P, P, P, P, P, P],
    "sehip_rbn_scratch_floats": [L, I],
    "sehip_rbn_stats": [P, L, I, I, P, P],
    "sehip_rbn_finalize": [P, P, P, P, P, P, P, P, P, P, P, L, I, I, F, F, I, P, P],
    "sehip_rbn_finalize_s": [P, P, P, P, P, P, P, P, P, P, P, L, I, I, F, F, I, P, P, P],
    "sehip_rbn_apply": [P, P, L, I, I, P, P],
    "sehip_rbn_bwd_reduce": [P, P, P, L, I, I, P, P],
    "sehip_rbn_bwd_reduce_fin": [P, P, P, L, I, I, P, P, P, P, P, P, P, P],
    "sehip_rbn_bwd_finalize": [P, P, L, I, I, P, P, P, P, P, P],
    "sehip_rbn_bwd_apply": [P, P, P, P, L, I, I, P, P],
    "sehip_dcunet_pack_input": [P, I, I, I, P, P],
    "sehip_dcunet_mask_fwd": [P, P, P, P, P, P, I, I, I, I, I, I, P, P, P],
    "sehip_dcunet_mask_bwd": [P, P, P, P, P, P, I, I, I, I, I, I, P, P, P],
    "sehip_dcunet_mask_fwd_bn": [P, P, P, P, P, P, P, I, I, I, I, I, I, P, P, P],
    "sehip_dcunet_tail_scratch_floats": [I, I, I, I],
    "sehip_dcunet_tail_bwd": [P, P, P, P, P, P, P, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P],
    "sehip_ctn_encoder_fwd": [P, P, P, P, I, I, I, I, I, P, P, P],
    "sehip_ctn_encoder_bwd": [P, P, P, P, P, I, I, I, I, I, P, P, P],
    "sehip_ctn_codec_bwd_scratch_floats": [I, I, I, I, I],
    "sehip_ctn_gln_stats": [P, P, I, I, I, P, P],
    "sehip_ctn_dwconv_fwd": [P, P, P, P, P, P, I, I, P, I, I, I, P, P, P],
    "sehip_ctn_gln_apply": [P, P, P, P, P, I, I, I, P, P],
    "sehip_ctn_gln_bwd": [P, P, P, P, P, P, P, I, I, I, I, I, I, P, P, P, P, P, P],
    "sehip_ctn_gln_bwd_scratch_floats": [I, I, I],
    "sehip_ctn_mask_softmax_fwd": [P, L, I, I, P, P],
    "sehip_ctn_mask_softmax_bwd": [P, P, L, I, I, P],
    "sehip_ctn_decoder_fwd": [P, P, P, I, I, I, I, I, I, I, P, P],
    "sehip_ctn_decoder_bwd": [P, P, P, P, I, I, I, I, I, I, I, P, P, P, P, P],
    "sehip_dmx_prep": [P, I, I, I, I, I, I, I, I, P, I, I, P, P, P, P],
    "sehip_dmx_post": [P, P, I, I, I, L, I, I, I, P, I, I, P, P],
    "sehip_dmx_post_bwd": [P, P, I, I, I, L, I, I, I, P, I, I, P, P],
    "sehip_dmx_gn_stats": [P, I, I, I, I, P, P],
    "sehip_dmx_act_fwd": [P, P, P, P, I, F, I, P, P, P, I, I, I, P, P],
    "sehip_dmx_act_bwd": [P, P, P, P, P, I, F, I, P, I, I, I, P, P, P, P],
    "sehip_dmx_add": [P, P, L, P, P],
    "sehip_dmx_f32_to_bf16": [P, L, P, P],
    "sehip_dmx_frames": [I, P, P, I, I, I, I, I, I, P, P],
    "sehip_dmx_lstm_sync_bytes": [],
    "sehip_dmx_lstm_fwd": [P, P, I, I, I, P, P, P, P],
    "sehip_dmx_lstm_bwd": [P, P, P, P, I, I, I, P, P, P, P],
    "sehip_dmx_attn_fwd": [P, I, I, I, I, I, I, P, P],
    "sehip_dmx_attn_bwd_scratch_floats": [I, I, I],
    "sehip_dmx_attn_bwd": [P, P, I, I, I, I, I, I, P, P, P],
    "sehip_lstm_fwd": [P, P, P, I, I, I, P, P, P, P],
    "sehip_lstm_bwd": [P, P, P, P, P, I, I, I, P, P, P],
    "sehip_rlstm_fwd": [P, P, I, I, I, P, P, P, P],
    "sehip_rlstm_bwd": [P, P, P, P, I, I, I, P, P],
    "sehip_lstm2_gran_bytes": [I, I, I],
    "sehip_lstm2_sync_bytes": [],
    "sehip_lstm2_fwd": [P, P, P, P, P, P, I, I, I, P, P, P, P, P, P, P, P, U, P],
    "sehip_lstm2_bwd": [P, P, P, P, P, P, P, P, P, I, I, I, P, P, P, P, P, P, U, P],
    "sehip_lstm_fwd_chunk": [P, P, P, I, I, I, I, I, P, P, P, P],
    "sehip_lstm_bwd_chunk": [P, P, P, P, P, I, I, I, I, I, P, P, P, P],
}
_RESTYPE = {"sehip_lstm2_gran_bytes": C.c_long, "sehip_dmx_attn_bwd_scratch_floats": C.c_long, "sehip_ctn_codec_bwd_scratch_floats": C.c_long, "sehip_ctn_gln_bwd_scratch_floats": C.c_long, "sehip_wgrad_group_bytes": C.c_long, "sehip_wgrad_dense_group_bytes": C.c_long, "sehip_cbn_scratch_floats": C.c_long, "sehip_rbn_scratch_floats": C.c_long, "sehip_dcunet_tail_scratch_floats": C.c_long, "sehip_event_create": C.c_void_p, "sehip_stream_create": C.c_void_p}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SehipError(
                f"{LIB_PATH} is missing: build it with `python {os.path.join(_HERE, 'build.py')}` "
                "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the HIP path.")
        _lib = C.CDLL(LIB_PATH)
        _lib.sehip_last_error.restype = C.c_char_p
        _lib.sehip_last_kernel.restype = C.c_char_p
        for name, args in _PROTOS.items():
            fn = getattr(_lib, name, None)
            if fn is None:
                continue
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, C.c_int)
    return _lib


def declared_symbols():
    return ["sehip_last_error", "sehip_last_kernel"] + list(_PROTOS)


def check(status, what=""):
    if status != 0:
        msg = lib().sehip_last_error().decode()
        raise SehipError(f"{what}: status {status}: {msg}")


def call(name, *args):
    fn = getattr(lib(), name)
    check(fn(*args), name)


def ptr(t):
    """Device (or host) pointer of a contiguous tensor, None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "libsehip needs contiguous tensors"
    return t.data_ptr()


import threading

_tls = threading.local()      # per thread: autograd's backward thread (and any user thread) never sees another thread's cached stream


def stream():
    """hipStream_t of torch's current stream (cached inside a stream_scope of the calling thread)."""
    h = getattr(_tls, "handle", None)
    if h is not None:
        return h
    return torch.cuda.current_stream().cuda_stream


class stream_scope:
    """Looks torch's current stream up ONCE for all library calls the calling thread makes inside the scope: the lookup costs
    ~8 us of host time and a training step makes ~110 of them (a quarter of the step's launch time).  Code inside must not
    switch torch's current stream and then expect stream() to follow (the side streams of sehip/plan.py are passed as explicit
    handles): with SEHIP_DEBUG_STREAMS set the exit checks that torch's current stream is still the cached one."""

    def __enter__(self):
        self._prev = getattr(_tls, "handle", None)
        _tls.handle = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, *exc):
        cached = _tls.handle
        _tls.handle = self._prev
        if exc[0] is None and os.environ.get("SEHIP_DEBUG_STREAMS") and torch.cuda.current_stream().cuda_stream != cached:
            raise SehipError("stream_scope: torch's current stream changed inside the scope; library calls went to the stream "
                             "that was current at entry")
        return False


def require_gpu(t, what):
    if not t.is_cuda:
        raise SehipError(f"{what}: tensor is on {t.device}; the HIP path needs a gfx950 GPU (no CPU fallback)")
