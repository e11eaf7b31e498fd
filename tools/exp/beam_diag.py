import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import test_gpu_model as T
import __graft_entry__ as ge
for c in [('lut', 4, 0.2, 30, 9, 0.16), ('costt', 4, 0.0, 30, 10, 0.17), ('lut', 3, 1.0, 30, 11, 0.15), ('lut', 4, 0.2, 30, 12, 0.0)]:
    T.test_fp32_beam_search_ids_bit_exact(*c)
