"""st/tools/summaryTools.py:5-71 - "Logging in tensorboard without tensorflow ops": scalar and histogram summaries in
TensorBoard's event-file format, written without TensorFlow.  An event file is a TFRecord container (the same framing
and CRC-32C as the data files, tftools/tfRecord.py) of `Event` messages (tensorflow/core/util/event.proto:
wall_time = 1 double, step = 2 int64, file_version = 3 string, summary = 5; summary.proto: Summary.value = 1 with
tag = 1, simple_value = 2 float, histo = 5; HistogramProto min/max/num/sum/sum_squares = 1..5 double,
bucket_limit = 6 / bucket = 7 packed doubles)."""
import logging
import os
import socket
import struct
import time

import numpy as np

from .tftools.tfRecord import TFRecordWriter, _ld, _varint


def _double(field, v):
    return _varint((field << 3) | 1) + struct.pack('<d', float(v))


def _float(field, v):
    return _varint((field << 3) | 5) + struct.pack('<f', float(v))


def _int(field, v):
    return _varint(field << 3) + _varint(int(v) & 0xFFFFFFFFFFFFFFFF)


def _event(step=None, summary=None, file_version=None, wall_time=None):
    out = _double(1, time.time() if wall_time is None else wall_time)
    if step is not None:
        out += _int(2, step)
    if file_version is not None:
        out += _ld(3, file_version.encode())
    if summary is not None:
        out += _ld(5, summary)
    return out


class Summary(object):
    def __init__(self, log_dir):
        """Creates a summary writer logging to log_dir (tf.summary.FileWriter's file naming)."""
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, 'events.out.tfevents.{:010d}.{}'.format(int(time.time()), socket.gethostname()))
        self.writer = TFRecordWriter(self.path)
        self.writer.write(_event(file_version='brain.Event:2'))
        self.flush()

    def flush(self):
        self.writer.f.flush()

    def _add(self, values, step):
        self.writer.write(_event(step=step, summary=b''.join(_ld(1, v) for v in values)))

    def summary_scalar(self, tag, value, step):
        """:12-25"""
        self._add([_ld(1, tag.encode()) + _float(2, value)], step)
        self.flush()

    def summary_distribution(self, distribute, name):
        """:27-38 - distribute: [time x num_output], one scalar per output and time step"""
        len_time, num_output = distribute.shape
        for t in range(len_time):
            for i in range(num_output):
                self._add([_ld(1, (name + '_' + str(i)).encode()) + _float(2, distribute[t][i])], t)
        self.flush()
        logging.info('summaried the distribution!')

    def histogram(self, tag, values, step, bins=1000):
        """:40-71 - numpy histogram; the first bin edge is dropped as in the reference"""
        values = np.array(values)
        counts, bin_edges = np.histogram(values, bins=bins)
        hist = (_double(1, np.min(values)) + _double(2, np.max(values)) + _double(3, int(np.prod(values.shape))) +
                _double(4, np.sum(values)) + _double(5, np.sum(values ** 2)) +
                _ld(6, struct.pack('<%dd' % (len(bin_edges) - 1), *bin_edges[1:])) +
                _ld(7, struct.pack('<%dd' % len(counts), *counts.astype(np.float64))))
        self._add([_ld(1, tag.encode()) + _ld(5, hist)], step)
        self.flush()

    def close(self):
        self.writer.close()
