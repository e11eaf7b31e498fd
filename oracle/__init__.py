"""CPU oracle for the neurst-ST training hot path.  TEST INFRASTRUCTURE ONLY.

This package is a numpy restatement of the arithmetic of the reference
(`dqqcasia/neurst`, TensorFlow 1.15) on the `TransformerTPE` train/infer path.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it; the product path (`neurst_amd/`) never does and fails loudly without
its HIP library.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for any
op on this path, and TensorFlow cannot be imported in the build container
(ordinary ModuleNotFoundError).  The only known answers
the reference holds are docstring examples (`splice`, `sequence_mask`), which
`tests/test_oracle_pins.py` checks.  Everything else is pinned by (1) following
the reference files line by line (each function cites file:line), (2) an
independent cross-check against torch-CPU autograd / torch.nn.functional.ctc_loss
and finite differences in `tests/`.
"""
