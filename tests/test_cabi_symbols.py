"""CPU: libsehip.so loads and exports every symbol include/sehip.h declares (no compute calls without a GPU), and the
ctypes mirror of the descriptor matches the C layout."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "sehip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sehip_[a-z0-9_]+)\s*\(", text)))


def test_header_lists_the_whole_abi():
    names = declared_functions()
    assert len(names) >= 25
    for must in ("sehip_gemm", "sehip_wgrad", "sehip_stft_fwd", "sehip_istft_bwd", "sehip_sisnr_fwd", "sehip_opt_step",
                 "sehip_cbn_bwd_apply", "sehip_lstm_bwd", "sehip_unpack_grad"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from sehip import _lib
    lib = _lib.lib()
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python binding does not call anything the header does not declare
    undeclared = [n for n in _lib.declared_symbols() if n not in declared_functions()]
    assert not undeclared, undeclared
    assert lib.sehip_version() >= 100


def test_descriptor_layout_matches_c():
    from sehip import _lib
    from sehip.plan import CGemmDesc
    assert ctypes.sizeof(CGemmDesc) == _lib.lib().sehip_gemm_desc_size()


def test_errors_are_reported_not_thrown():
    from sehip import _lib
    lib = _lib.lib()
    # argument validation happens before any HIP call: safe without a GPU
    assert lib.sehip_sisnr_fwd(None, None, 0, 0, None, None, None) != 0
    assert b"empty" in lib.sehip_last_error()
    assert lib.sehip_stft_fwd(None, None, 1, 1000, 400, 100, 256, None, None, None) != 0
    assert b"fft_len 512" in lib.sehip_last_error()
    with pytest.raises(_lib.SehipError):
        _lib.call("sehip_lstm_fwd", None, None, None, 1, 1, 32, None, None, None, None)
    # chunked recurrences: the step range is validated, a partial backward range needs the carried state
    assert lib.sehip_lstm_fwd_chunk(None, None, None, 2, 10, 64, 5, 5, None, None, None, None) != 0
    assert b"step range" in lib.sehip_last_error()
    assert lib.sehip_lstm_bwd_chunk(None, None, None, None, None, 2, 10, 64, 0, 5, None, None, None, None) != 0
    assert b"state buffer" in lib.sehip_last_error()
    # stft_custom: n_fft in [2, 4096] (512 on the FFT path, anything else a direct DFT: round 6); frame count helper follows torch.stft
    assert lib.sehip_stft_custom_fwd(None, 1, 100000, 8192, 2048, 8192, 1, None, None) != 0
    assert b"n_fft 8192 outside" in lib.sehip_last_error()
    assert lib.sehip_stft_custom_frames(32768, 512, 128, 1) == 257
    assert lib.sehip_stft_custom_frames(2048, 512, 256, 0) == 7
    assert lib.sehip_stft_custom_frames(1500, 255, 60, 1) == 25        # odd n_fft: the centre padding is n_fft - 1 samples in all
    assert lib.sehip_stream_depend(None, None, None) != 0
