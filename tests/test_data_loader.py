"""the reference's on-disk dataset format and chronological / inductive split, against the split the
reference loader itself produced for the same files (tests/golden/loader.npz, made by tests/golden/make_golden.py)."""
import os

import numpy as np

from lstep_amd import data as ld
from lstep_amd import synth


def test_loader_reproduces_reference_split(golden, tmp_path):
    z = golden("loader")
    g = synth.make_temporal_graph(num_nodes=120, num_edges=1500, seed=40, tie_quantum=7.0)
    rng = np.random.RandomState(41)
    ld.write_dataset(os.path.join(tmp_path, "processed_data"), "tiny", g["src"], g["dst"], g["ts"], rng.randint(0, 2, size=1500),
                     rng.standard_normal((1501, 12)), np.zeros((121, 172)))
    res = ld.get_link_prediction_data("tiny", 0.15, 0.15, root=os.path.join(tmp_path, "processed_data"))
    assert tuple(res[0].shape) == tuple(z["node_shape"]) == (121, 172) and tuple(res[1].shape) == tuple(z["edge_shape"]) == (1501, 172)
    np.testing.assert_allclose([res[1].sum(), np.abs(res[1]).sum()], z["edge_digest"], rtol=1e-12)
    assert np.all(res[1][:, 12:] == 0)  # 12 feature columns padded to 172 with zeros
    for name, d in zip(("full", "train", "val", "test", "new_val", "new_test"), res[2:]):
        np.testing.assert_array_equal(d.edge_ids, z[f"{name}/edge_ids"], err_msg=name)
        np.testing.assert_array_equal(d.src_node_ids, z[f"{name}/src"], err_msg=name)
        assert d.num_unique_nodes == int(z[f"{name}/num_unique_nodes"][0])
        assert d.src_node_ids.dtype == np.longlong and d.node_interact_times.dtype == np.float64
    # chronological, disjoint val/test, train strictly before the validation time
    assert res[3].node_interact_times.max() <= res[4].node_interact_times.min() <= res[5].node_interact_times.min()
