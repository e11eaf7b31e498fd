"""st/tools/avg_checkpoints.py:28-117 - "Script to average values of variables in a list of checkpoint files", same flags:

    python -m neurst_amd.st.tools.avg_checkpoints --checkpoints model-100,model-200 --prefix /path/to/dir/ --output_path averaged.ckpt
    python -m neurst_amd.st.tools.avg_checkpoints --num_last_checkpoints 5 --prefix /path/to/dir/ --output_path averaged.ckpt

Reads TensorFlow (TensorBundle) or .npz checkpoints; writes the container named by --format (default: that of the inputs)."""
import argparse
import logging
import os

from .tftools import checkpointTools as ck


def all_model_checkpoint_paths(directory):
    """tf.train.get_checkpoint_state(dir).all_model_checkpoint_paths, oldest first (both state-file formats)"""
    state = os.path.join(directory, 'checkpoint')
    lines = [l.strip() for l in open(state) if l.strip()]
    listed = [l.split(':', 1)[1].strip().strip('"') for l in lines if l.startswith('all_model_checkpoint_paths:')]
    if not listed:
        listed = [l for l in lines if ':' not in l]
    return [n if os.path.isabs(n) else os.path.join(directory, n) for n in listed]


def checkpoint_exists(path):
    return os.path.exists(path) or os.path.exists(path + '.npz') or os.path.exists(path + '.index')


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--checkpoints', default='', help='Comma-separated list of checkpoints to average.')
    p.add_argument('--num_last_checkpoints', type=int, default=0, help='Averages the last N saved checkpoints. If the checkpoints flag is set, this is ignored.')
    p.add_argument('--prefix', default='', help='Prefix (e.g., directory) to append to each checkpoint.')
    p.add_argument('--output_path', default='averaged.ckpt', help='Path to output the averaged checkpoint to.')
    p.add_argument('--format', default=None, choices=[None, 'npz', 'tf'], help="container of the output (default: 'tf' if the inputs are TensorBundles)")
    a = p.parse_args(argv)
    if a.checkpoints:
        checkpoints = [c.strip() for c in a.checkpoints.split(',') if c.strip()]
        if not checkpoints:
            raise ValueError('No checkpoints provided for averaging.')
        if a.prefix:
            checkpoints = [a.prefix + c for c in checkpoints]
    else:
        assert a.num_last_checkpoints >= 1, 'Must average at least one model'
        assert a.prefix, 'Prefix must be provided when averaging last N checkpoints'
        checkpoints = all_model_checkpoint_paths(os.path.dirname(a.prefix))[-a.num_last_checkpoints:]
    checkpoints = [c for c in checkpoints if checkpoint_exists(c)]
    if not checkpoints:
        raise ValueError('None of the provided checkpoints exist. %s' % a.checkpoints if a.checkpoints
                         else 'Could not find checkpoints at %s' % os.path.dirname(a.prefix))
    logging.info('Reading variables and averaging checkpoints: %s', checkpoints)
    fmt = a.format or ('tf' if os.path.exists(ck.resolve(checkpoints[0]) + '.index') else 'npz')
    out = ck.average_checkpoints(checkpoints, a.output_path, fmt=fmt)
    logging.info('Averaged checkpoints saved in %s', out)
    return out


if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO)
    main()
