"""The only known answers the reference itself holds on this path: docstring examples.
splice: st/tools/tftools/tfAudioTools.py:69-80; batch_splice :111-123 (off-path, same
padding rule); sequence_mask: st/tools/tftools/tfTools.py:18-25."""
import numpy as np

from oracle import ops


def test_splice_docstring_example():
    x = np.array([[i, i, i] for i in range(1, 8)], np.float32)
    want = np.array([[1, 1, 1, 2, 2, 2, 3, 3, 3], [2, 2, 2, 3, 3, 3, 4, 4, 4], [3, 3, 3, 4, 4, 4, 5, 5, 5],
                     [4, 4, 4, 5, 5, 5, 6, 6, 6], [5, 5, 5, 6, 6, 6, 7, 7, 7], [6, 6, 6, 7, 7, 7, 0, 0, 0],
                     [7, 7, 7, 0, 0, 0, 0, 0, 0]], np.float32)
    assert np.array_equal(ops.splice(x, 0, 2), want)


def test_process_raw_feature_shape_and_rows():
    x = np.arange(10 * 2, dtype=np.float32).reshape(10, 2)
    y = ops.process_raw_feature(x, num_context=5, downsample=3)
    assert y.shape == (4, 12)
    assert np.array_equal(y[0], x[0:6].reshape(-1))
    assert np.array_equal(y[3, :2], x[9]) and np.all(y[3, 2:] == 0)


def test_timing_signal_first_rows():
    s = ops.timing_signal(3, 6)
    assert np.allclose(s[0], [0, 0, 0, 1, 1, 1])
    assert np.isclose(s[1, 0], np.sin(1.0)) and np.isclose(s[1, 2], np.sin(1e-4), rtol=1e-5)


def test_handle_gradients_known_answer():
    """gradientTools.py:40-67 by hand: L2 skips names containing 'bias'; clip_by_global_norm scales by clip / max(norm, clip)."""
    import numpy as np
    from oracle.model import handle_gradients
    g = {'a/kernel': np.array([3.0, 0.0]), 'a/bias': np.array([4.0])}
    P = {'a/kernel': np.array([10.0, -10.0]), 'a/bias': np.array([100.0])}
    out = handle_gradients(g, P, lambda_l2=0.1)
    assert np.allclose(out['a/kernel'], [4.0, -1.0]) and np.allclose(out['a/bias'], [4.0])
    out = handle_gradients(g, P, grad_clip_global_norm=2.5)            # norm 5 -> factor 0.5
    assert np.allclose(out['a/kernel'], [1.5, 0.0]) and np.allclose(out['a/bias'], [2.0])
    out = handle_gradients(g, P, grad_clip_global_norm=50.0)           # norm below the clip: unchanged
    assert np.allclose(out['a/kernel'], [3.0, 0.0])
