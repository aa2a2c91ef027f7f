"""CPU-side checks of the training path's host helpers (the networks themselves need the HIP library: tests/test_gpu_train.py)."""
import os

import numpy as np

from yoho_amd import synth, weights as W
from yoho_amd.utils import quaternion_from_matrix, matrix_from_quaternion

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train.npz")


def test_quaternion_from_matrix_known_answers():
    # known-answer vectors of the reference's docstring (utils/r_eval.py quaternion_from_matrix)
    assert np.allclose(quaternion_from_matrix(np.identity(3)), [1, 0, 0, 0])
    q = quaternion_from_matrix(np.diag([1.0, -1.0, -1.0]))
    assert np.allclose(q, [0, 1, 0, 0]) or np.allclose(q, [0, -1, 0, 0])
    R = [[-0.545, 0.797, 0.260], [0.733, 0.603, -0.313], [-0.407, 0.021, -0.913]]
    assert np.allclose(quaternion_from_matrix(R), [0.19069, 0.43736, 0.87485, -0.083611], atol=1e-5)
    R = [[0.395, 0.362, 0.843], [-0.626, 0.796, -0.056], [-0.677, -0.498, 0.529]]
    assert np.allclose(quaternion_from_matrix(R), [0.82336615, -0.13610694, 0.46344705, -0.29792603], atol=1e-6)
    rs = np.random.RandomState(0)
    for _ in range(20):                                              # round trip through the reference's quaternion -> matrix
        q = rs.randn(4)
        q /= np.linalg.norm(q)
        if q[0] < 0:
            q = -q
        assert np.allclose(quaternion_from_matrix(matrix_from_quaternion(q)), q, atol=1e-9)


def test_training_fixture_is_complete_and_reproducible(tables):
    g = np.load(GOLD)
    names = [n for n, _ in W.PARTI_SPEC if not n.endswith("num_batches_tracked")]
    for n in names:
        key = ("p1_buf_" if ("running_" in n) else "p1_grad_") + n
        assert key in g.files, key
    assert sum(k.startswith("p2_grad_") for k in g.files) == 22      # PartII's own parameters (PartI is frozen there)
    b1 = synth.train_batch(int(g["bn"]), tables.P, seed=int(g["seed"]))
    b2 = synth.train_batch(int(g["bn"]), tables.P, seed=int(g["seed"]))
    assert all(np.array_equal(b1[k], b2[k]) for k in b1) and b1["feats0"].shape == (1, 6, 32, 60)
    assert np.allclose(np.linalg.norm(b1["deltaR"][0], axis=1), 1, atol=1e-6)
    d = synth.tensor_digest(np.arange(40, dtype=np.float32))
    assert d.shape == (19,) and d[1] == 780.0 and np.array_equal(d[3:], np.arange(16))
