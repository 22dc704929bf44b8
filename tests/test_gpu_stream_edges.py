"""GPU: the streaming kernels of the outer layers (csrc/convt.hip) at the edges of their launch geometry, at the HEADLINE widths (where
every one of them engages: 4 KB frames): a single utterance of 16 frames, fewer frames than workgroups per utterance (the chunk count
collapses to the frame count, one frame per workgroup, prologue longer than the run), an odd batch above 32 (more than 512 rows of
BatchNorm sums from the fused reduce pass: sehip_cbn_bwd_finalize_n's 1024-row instantiation), a frame count that leaves a short last
chunk.  Every operator is re-computed by the oracle from the HIP path's own inputs (the op-local tests of tests/test_gpu_ops_local.py,
same tolerances).  Reference math: src/model/dccrn.py:139-212, 316-450, 457-634."""
import pytest

import test_gpu_ops_local as L
from test_gpu_ops_local import (test_encoder_conv_forward_dgrad_wgrad, test_decoder_deconv_forward_dgrad_wgrad,  # noqa: F401
                                test_complex_batchnorm_prelu_forward_backward)

pytestmark = pytest.mark.gpu
WIDTHS = dict(kernel_num=[16, 32, 64, 128, 256, 256], rnn_units=128)


@pytest.fixture(scope="module", params=[(1, 1500), (5, 2500), (33, 1700), (3, 6700)], ids=lambda p: f"B{p[0]}-N{p[1]}")
def run(request):
    B, N = request.param
    r = L.build_run(dict(WIDTHS, length=N), B, N, seed=21 + B)
    ws = r["ws"]
    # the streaming launches are what ran: the library names the kernel of the last product it launched
    from sehip import _lib
    assert ws.bnr_rows, "the fused BatchNorm reduce pass did not engage at the headline widths"
    ws.gemm("dec4.dg")                 # (an input gradient: it rewrites the same values, no sums are accumulated twice)
    assert b"convs_stream_kernel" in _lib.lib().sehip_last_kernel()
    return r
