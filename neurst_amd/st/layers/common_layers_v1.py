"""layer_norm of st/layers/common_layers_v1.py:342-380 (mean / biased variance over the last
dimension, eps=1e-6 inside the rsqrt, learned scale and bias).  The hot path always calls it
through `residual` (st/models/tools/utils.py:324-337), where it is fused with the dropout and
the residual add in one HBM-bound kernel."""
from ..models import graph as G
from ... import ops


def layer_norm_vars(filters, name=None):
    g = G.current()
    with (g.variable_scope(name) if name else g.default_scope('layer_norm')):
        scale = g.get_variable('layer_norm_scale', [filters])
        bias = g.get_variable('layer_norm_bias', [filters])
    return scale, bias
