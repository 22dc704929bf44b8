"""Host logic of the gather-ordered un-pack table (plan.gather_ordered_unpack_table, round 6): a permutation inside every tensor, the
table rows moved with it, neighbouring rows reading neighbouring packed entries."""
import numpy as np


def test_gather_order_is_a_permutation_inside_every_tensor():
    from sehip.plan import DCCRNConfig, DCCRNStatic, gather_ordered_unpack_table
    st = DCCRNStatic(DCCRNConfig(kernel_num=[16, 16, 32, 32, 64, 64], length=4000))
    offs = st.layout.tensor_offsets
    tg, pm = gather_ordered_unpack_table(st.utab, offs)
    n = st.utab.shape[0]
    assert pm.dtype == np.int32 and np.array_equal(np.sort(pm), np.arange(n))
    assert np.array_equal(tg, st.utab[pm])
    t_of = np.searchsorted(offs, np.arange(n), side="right") - 1
    assert np.array_equal(t_of[pm], t_of)                      # the per-tensor sums are taken by position
    # an emulated un-pack gives the same gradients either way
    rng = np.random.default_rng(0)
    packed = rng.standard_normal(int((st.utab >> 1).max()) + 1).astype(np.float32)

    def unpack(tab):
        e = tab.astype(np.int64)
        v = np.where(e >= 0, packed[np.maximum(e >> 1, 0)] * np.where(e & 1, -1.0, 1.0), 0.0)
        return v.sum(axis=1, dtype=np.float64)
    g0 = unpack(st.utab)
    g1 = np.empty_like(g0); g1[pm] = unpack(tg)
    assert np.array_equal(g0, g1)
    f0, f1 = st.utab[:, 0].astype(np.int64) >> 1, tg[:, 0].astype(np.int64) >> 1
    assert (np.diff(f1) == 1).mean() > 0.9 > (np.diff(f0) == 1).mean()
